cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r04_fuzz_long.txt
for seed in 70001 70002 70003 70004 70005 70006 70007 70008; do
  t0=$SECONDS
  ( timeout 900 python tools/fuzz_parity.py 15000 $seed 2>&1 | tail -2 ) >> gpurun_out/r04_fuzz_long.txt
  echo "  (seed $seed: $((SECONDS - t0)) s)" >> gpurun_out/r04_fuzz_long.txt
done
cat gpurun_out/r04_fuzz_long.txt
