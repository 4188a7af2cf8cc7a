cd $GRAFT_REPO_ROOT
timeout 600 python tools/s3_bucket_stats.py 667e6 21 2>&1 | tail -32
