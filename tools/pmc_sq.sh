#!/bin/bash
# One SQ counter pass (wave / busy / wait cycles, VALU and LDS instruction counts) of bench.py per library variant, seconds each:
#   tools/pmc_sq.sh name[:lib.so] ...   ->  gpurun_out/pmc1_<name>.json   (SP_PMC_ARGS="-k 17" etc. go to bench.py)
# VALU utilisation of a kernel = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x its time x 2.4 GHz): how round 6 found that k3_eval
# and k5_map2 were bound by instruction issue (0.88, 0.70) and what the two-phase walk removed (15.9 G -> 11.3 G instructions).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  v=${spec%%:*}; lib=${spec#*:}
  if [ "$lib" != "$spec" ]; then export SUBPHASER_HIP_LIB=$R/$lib; else unset SUBPHASER_HIP_LIB; fi
  rm -rf /tmp/pm_$v
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pm_$v -o run -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${SP_PMC_ARGS:-} > /tmp/pm_$v.log 2>&1
  python $R/tools/pmc_table.py /tmp/pm_$v/run_counter_collection.csv > $R/gpurun_out/pmc1_$v.json
done
