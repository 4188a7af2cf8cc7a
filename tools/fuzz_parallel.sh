#!/bin/bash
# N fuzz processes side by side on one GPU (they are bound by the host -- the oracle and Python -- and keep each other's kernels
# company): tools/fuzz_parallel.sh <procs> <iterations each> <first seed> <seconds each> [streams]  ->  one summary line per process
cd ${GRAFT_REPO_ROOT:-.}
n=$1; iters=$2; seed=$3; secs=$4; mode=$5
for i in $(seq 0 $((n - 1))); do
    SP_FUZZ_SECONDS=$secs python tools/fuzz_parity.py $iters $((seed + 7919 * i)) $mode 2>&1 | grep -E "^fuzz|MISMATCH|Error|error" &
done
wait
