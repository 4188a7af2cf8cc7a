set -u
export SP_COMMIT=61e9496
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r03_final_pytest_gpu.txt
bash tools/pmc_round.sh r03_wheat > /dev/null 2>&1
bash tools/pmc_round.sh r03_wheat_k17 -k 17 > /dev/null 2>&1
bash tools/pmc_round.sh r03_wheat_k21 -k 21 > /dev/null 2>&1
cp gpurun_out/r03_wheat*_pmc.json gpurun_out/r03_wheat*_kernel_stats.md profiles/
python bench.py 2> gpurun_out/r03_bench_wheat_k15.err | tail -1 > gpurun_out/r03_bench_wheat_k15.json
python bench.py -k 17 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_wheat_k17.json
python bench.py -k 21 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_wheat_k21.json
python bench.py --config peanut --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_peanut_k15.json
python bench.py --config ara --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_ara_k15.json
python tools/stage_ms.py gpurun_out/r03_bench_*.json
cat gpurun_out/r03_final_pytest_gpu.txt
