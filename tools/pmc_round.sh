#!/bin/bash
# PMC passes of one round on the GPU box (run through gpurun from the repo root).  Each group is its own
# rocprofv3 run (--pmc with --kernel-trace only: MI355X_MICROARCH.md; gpurun refuses other combinations).
#   usage: tools/pmc_round.sh <tag> [bench args...]
# Output: gpurun_out/<tag>_pmc.json (per kernel, per launch averages) + gpurun_out/<tag>_kernel_stats.md
set -u
TAG=${1:-r02}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"
         "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
rm -rf $O/${TAG}_prof
CMD=${SP_PMC_CMD:-"python $R/bench.py --no-cpu-baseline"}
CMD1=${SP_PMC_CMD:-"python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"}
timeout 900 rocprofv3 --kernel-trace -d $O/${TAG}_prof -o run -- $CMD "$@" > $O/${TAG}_prof.log 2>&1
python $R/tools/rocpd_summary.py $O/${TAG}_prof/run_results.db > $O/${TAG}_kernel_stats.md 2>> $O/${TAG}_prof.log
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf $O/${TAG}_pmc$i
  timeout 600 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/${TAG}_pmc$i -o run -- $CMD1 "$@" > $O/${TAG}_pmc$i.log 2>&1
  i=$((i+1))
done
python $R/tools/pmc_table.py $O/${TAG}_pmc*/run_counter_collection.csv > $O/${TAG}_pmc.json 2>> $O/${TAG}_prof.log
rm -rf $O/${TAG}_pmc[0-9]* $O/${TAG}_prof
head -30 $O/${TAG}_kernel_stats.md
