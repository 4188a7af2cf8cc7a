#!/bin/bash
# N fuzz processes side by side on one GPU (they are bound by the host -- the oracle and Python -- and keep each other's kernels
# company; in stream mode they share ONE busy neighbour, tools/gpu_busy.py, instead of one each: sixteen contexts ran out of
# device memory):  tools/fuzz_parallel.sh <procs> <iterations each> <first seed> <seconds each> [streams]  ->  one line per process
cd ${GRAFT_REPO_ROOT:-.}
n=$1; iters=$2; seed=$3; secs=$4; mode=$5
if [ "$mode" = "streams" ]; then
    export SP_FUZZ_NO_BUSY=1
    sleep $((secs + 30)) | python tools/gpu_busy.py ${SP_BUSY_SCALE:-0.01} > /dev/null 2>&1 &
    busy=$!
fi
pids=""
for i in $(seq 0 $((n - 1))); do
    (SP_FUZZ_SECONDS=$secs python tools/fuzz_parity.py $iters $((seed + 7919 * i)) $mode > gpurun_out/fuzz_proc_$i.log 2>&1; grep -E "^fuzz|MISMATCH|Error|error" gpurun_out/fuzz_proc_$i.log) &
    pids="$pids $!"
done
wait $pids
[ -n "$busy" ] && kill $busy 2>/dev/null
wait 2>/dev/null
