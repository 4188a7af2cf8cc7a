cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r04_fuzz_final.txt
for seed in 40404 40405 40406 40407; do
  t0=$SECONDS
  ( timeout 600 python tools/fuzz_parity.py 4000 $seed 2>&1 | tail -2 ) >> gpurun_out/r04_fuzz_final.txt
  echo "  (seed $seed, default lanes: $((SECONDS - t0)) s)" >> gpurun_out/r04_fuzz_final.txt
done
t0=$SECONDS
( SP_LANES_SPARSE=0 SP_LANES_DENSE=0 SP_LANES=0 timeout 600 python tools/fuzz_parity.py 4000 50505 2>&1 | tail -1 ) >> gpurun_out/r04_fuzz_final.txt
echo "  (seed 50505, one stream everywhere: $((SECONDS - t0)) s)" >> gpurun_out/r04_fuzz_final.txt
cat gpurun_out/r04_fuzz_final.txt
