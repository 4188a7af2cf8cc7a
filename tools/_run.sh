cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -k "map or pipeline or smoke or golden or bench or synth or dist" 2>&1 | tail -3 ) > gpurun_out/s17_pytest.txt
tools/ab_round.sh "split|||" > gpurun_out/s17.txt 2>&1
cat gpurun_out/s17_pytest.txt
python - <<'PY'
import json
for n in ("split",):
    d=json.load(open("gpurun_out/ab_%s.json"%n)); print(n, d["ms_per_step"], d["config"]["mapped_positions"], d["config"]["windows"], {k:v["ms_per_step"] for k,v in d["stages"].items() if k.startswith("k5")}, d["host_wall_ms_per_step"])
PY
