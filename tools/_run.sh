cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/s10_pytest.txt
tools/ab_round.sh "k21_l3|||-k 21" "k21_l0||SP_LANES_SPARSE=0|-k 21" "k21_l2||SP_LANES_SPARSE=2|-k 21" "k21_l4||SP_LANES_SPARSE=4|-k 21" "k17_l3|||-k 17" > gpurun_out/s10_k.txt 2>&1
cat gpurun_out/s10_pytest.txt gpurun_out/s10_k.txt
