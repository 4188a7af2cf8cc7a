#!/usr/bin/env python3
"""print value, ms_per_step and the per-kernel ms of one or more bench.py JSON lines (experiment helper)"""
import json
import sys
for p in sys.argv[1:]:
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(p, "unreadable:", e)
        continue
    st = d.get("stages", {})
    top = sorted(st.items(), key=lambda kv: -kv[1]["ms_per_step"])[:12]
    print("%s: %.2f %s  %.2f ms/step  verified=%s" % (p, d["value"], d["unit"], d["ms_per_step"], d.get("verified")))
    print("   " + "  ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in top))
