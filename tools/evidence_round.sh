set -u
# Round evidence on the GPU box (through gpurun, from the repo root): GPU suite, rocprofv3 kernel traces + stamped PMC
# passes of the five bench workloads, their bench lines (in-run verification against the oracle), the distributed path with
# one rank, end-to-end CLI runs with their phase tables, feature mode at scale, fuzz (single-stream and stream mode).
export SP_COMMIT=${SP_COMMIT:-unknown}
R=${SP_ROUND:-r06}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/${R}_final_pytest_gpu.txt
bash tools/pmc_round.sh ${R}_wheat > /dev/null 2>&1
bash tools/pmc_round.sh ${R}_wheat_k17 -k 17 > /dev/null 2>&1
bash tools/pmc_round.sh ${R}_wheat_k21 -k 21 > /dev/null 2>&1
bash tools/pmc_round.sh ${R}_peanut --config peanut > /dev/null 2>&1
bash tools/pmc_round.sh ${R}_ara --config ara > /dev/null 2>&1
cp gpurun_out/${R}_*_pmc.json gpurun_out/${R}_*_kernel_stats.md profiles/
timeout 600 python bench.py 2> gpurun_out/${R}_bench_wheat_k15.err | tail -1 > gpurun_out/${R}_bench_wheat_k15.json
timeout 600 python bench.py -k 17 2>/dev/null | tail -1 > gpurun_out/${R}_bench_wheat_k17.json
timeout 600 python bench.py -k 21 2>/dev/null | tail -1 > gpurun_out/${R}_bench_wheat_k21.json
timeout 600 python bench.py --config peanut 2>/dev/null | tail -1 > gpurun_out/${R}_bench_peanut_k15.json
timeout 600 python bench.py --config ara 2>/dev/null | tail -1 > gpurun_out/${R}_bench_ara_k15.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 600 python bench.py --no-cpu-baseline --force-dist --dist-selfcheck 2>/dev/null | tail -1 > gpurun_out/${R}_bench_wheat_k15_dist1.json
python tools/stage_ms.py gpurun_out/${R}_bench_*.json
for c in wheat peanut ara; do timeout 600 python tools/e2e_cli.py $c /tmp/sp_e2e_$c > gpurun_out/${R}_e2e_cli_$c.log 2>&1; rm -rf /tmp/sp_e2e_$c; done
timeout 900 python tools/feat_bench.py wheat 2e6 > gpurun_out/${R}_feature_mode_2M.log 2>&1
: > gpurun_out/${R}_fuzz_evidence.txt
(echo "# single-stream fuzz, 4 processes x 300 s"; bash tools/fuzz_parallel.sh 4 1000000 ${SP_FUZZ_SEED:-601} 300) >> gpurun_out/${R}_fuzz_evidence.txt 2>&1
(echo "# stream-mode fuzz (3..7 chains forced in flight, a busy neighbour), 2 processes x ${SP_FUZZ_STREAM_SECS:-900} s"; SP_BUSY_SCALE=0.002 bash tools/fuzz_parallel.sh 2 1000000 $((${SP_FUZZ_SEED:-601} + 8500)) ${SP_FUZZ_STREAM_SECS:-900} streams) >> gpurun_out/${R}_fuzz_evidence.txt 2>&1
cat gpurun_out/${R}_final_pytest_gpu.txt gpurun_out/${R}_fuzz_evidence.txt
