cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
t0=$SECONDS; ( timeout 600 python tools/fuzz_parity.py 2000 60606 2>&1 | tail -1 ); echo "  ($((SECONDS - t0)) s for 2000 default iterations)"
