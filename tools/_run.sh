cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q -k "count or golden or smoke or pipeline" 2>&1 | tail -3 )
timeout 300 python tools/k1_bench.py 667e6 2 15 2>&1 | tail -1 | cut -c1-500
timeout 300 python tools/k1_bench.py 667e6 2 14 2>&1 | tail -1 | cut -c1-500
tools/ab_round.sh "part1t|||" "part1t_b|||" | grep -A1 Gbases
