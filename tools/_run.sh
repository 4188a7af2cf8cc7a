cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py -k 17 2>/dev/null | tail -1 > gpurun_out/r04_bench_wheat_k17.json
python bench.py -k 21 2>/dev/null | tail -1 > gpurun_out/r04_bench_wheat_k21.json
python tools/stage_ms.py gpurun_out/r04_bench_wheat_k17.json gpurun_out/r04_bench_wheat_k21.json
