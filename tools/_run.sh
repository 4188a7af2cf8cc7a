cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/ab_round.sh "l3|||" "l2||SP_LANES_DENSE=2|" "l4||SP_LANES_DENSE=4|"
python - <<P
import json
for n in ("l3","l2","l4"):
    d=json.loads(open("gpurun_out/ab_%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["host_wall_ms_per_step"])
P
