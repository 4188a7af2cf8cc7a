#!/bin/bash
# Collect the judged evidence of one round on the GPU box (run through gpurun from the repo root):
#   1. bench.py default run (JSON line with roofline + cpu_baseline)
#   2. rocprofv3 --kernel-trace of the same command (rocpd database -> per-kernel table)
#   3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes
# Outputs under gpurun_out/; copy the summaries into profiles/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_final.json 2> $O/bench_final.err
rm -rf $O/prof_final $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
rocprofv3 --kernel-trace -d $O/prof_final -o wheat -- python $R/bench.py --no-cpu-baseline > $O/prof_final.log 2>&1
python $R/tools/rocpd_summary.py $O/prof_final/wheat_results.db > $O/kernel_stats.md 2>> $O/prof_final.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o wheat -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_summary.py $O > $O/pmc_traffic.json 2>> $O/prof_final.log
tail -1 $O/bench_final.json | cut -c1-400
head -12 $O/kernel_stats.md
