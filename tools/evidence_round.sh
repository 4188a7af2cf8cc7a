set -u
# Round evidence on the GPU box (through gpurun, from the repo root): GPU suite, rocprofv3 kernel traces + stamped PMC
# passes of the five bench workloads, their bench lines (in-run verification against the oracle), fuzz.
export SP_COMMIT=${SP_COMMIT:-unknown}
R=${SP_ROUND:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${R}_final_pytest_gpu.txt
bash tools/pmc_round.sh ${R}_wheat > /dev/null 2>&1
bash tools/pmc_round.sh ${R}_wheat_k17 -k 17 > /dev/null 2>&1
bash tools/pmc_round.sh ${R}_wheat_k21 -k 21 > /dev/null 2>&1
bash tools/pmc_round.sh ${R}_peanut --config peanut > /dev/null 2>&1
bash tools/pmc_round.sh ${R}_ara --config ara > /dev/null 2>&1
cp gpurun_out/${R}_*_pmc.json gpurun_out/${R}_*_kernel_stats.md profiles/
python bench.py 2> gpurun_out/${R}_bench_wheat_k15.err | tail -1 > gpurun_out/${R}_bench_wheat_k15.json
python bench.py -k 17 2>/dev/null | tail -1 > gpurun_out/${R}_bench_wheat_k17.json
python bench.py -k 21 2>/dev/null | tail -1 > gpurun_out/${R}_bench_wheat_k21.json
python bench.py --config peanut 2>/dev/null | tail -1 > gpurun_out/${R}_bench_peanut_k15.json
python bench.py --config ara 2>/dev/null | tail -1 > gpurun_out/${R}_bench_ara_k15.json
python tools/stage_ms.py gpurun_out/${R}_bench_*.json
cat gpurun_out/${R}_final_pytest_gpu.txt
