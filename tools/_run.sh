cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
tools/ab_round.sh "ovfidx|||" "ovfidx_b|||" | grep -A1 Gbases
python - <<'PY'
import json
d=json.load(open("gpurun_out/ab_ovfidx.json")); print({k:v["ms_per_step"] for k,v in d["stages"].items() if k.startswith("k3")}, d["host_wall_ms_per_step"])
PY
( timeout 300 python tools/fuzz_parity.py 2000 80808 2>&1 | tail -1 )
