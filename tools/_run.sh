cd $GRAFT_REPO_ROOT
python tools/labels_bench.py
SP_CTAB=0 python tools/labels_bench.py
