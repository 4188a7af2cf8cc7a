cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/s12_pytest.txt
tools/ab_round.sh "k21|||-k 21" "k21_sort||SP_S3_BIG=sort|-k 21" "k25|||-k 25" > gpurun_out/s12_k.txt 2>&1
( timeout 900 python bench.py -k 21 --steps 3 --warmup 1 2> gpurun_out/s12_verify.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k21 verified', d['verified'], d['value'], d['cpu_baseline'])" ) > gpurun_out/s12_verify.txt 2>&1
cat gpurun_out/s12_pytest.txt gpurun_out/s12_k.txt gpurun_out/s12_verify.txt
