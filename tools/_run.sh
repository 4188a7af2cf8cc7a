cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/ab_round.sh "l2||SP_LANES_DENSE=2|" "l3||SP_LANES_DENSE=3|" "l4||SP_LANES_DENSE=4|" "l5||SP_LANES_DENSE=5|" "l6||SP_LANES_DENSE=6|" 2>&1 | grep Gbases
