#!/usr/bin/env python3
"""Per-kernel, per-launch averages of every counter found in rocprofv3 counter-collection CSVs
(one CSV per --pmc pass).  FETCH_SIZE / WRITE_SIZE are KiB in the CSV and converted to bytes here;
`FETCH_SIZE_x2` is the gfx950 streaming-read correction MI355X_MICROARCH.md prescribes (valid for wide
coalesced reads only: gather kernels are flagged and should be read uncorrected).

usage: pmc_table.py <csv>... > table.json"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

GATHER = ("k5_map", "k5_map2", "k5_map_lab", "k5_map_sparse", "k5_map_sparse2", "k5_map_feat", "k5_map_feat_sparse")


def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            a = acc[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    out = {}
    for k, cs in sorted(acc.items()):
        ent = {}
        for name, (n, tot) in sorted(cs.items()):
            v = tot / max(n, 1)
            if name in ("FETCH_SIZE", "WRITE_SIZE"):
                v *= 1024.0
            ent[name] = v
            ent.setdefault("calls", n)
        if "FETCH_SIZE" in ent:
            ent["read_bytes"] = ent["FETCH_SIZE"] * (1.0 if k in GATHER else 2.0)
            ent["fetch_correction"] = 1.0 if k in GATHER else 2.0
        if "WRITE_SIZE" in ent:
            ent["write_bytes"] = ent["WRITE_SIZE"]
        if "TCC_HIT_sum" in ent and ent["TCC_HIT_sum"] + ent.get("TCC_MISS_sum", 0) > 0:
            ent["l2_hit_rate"] = ent["TCC_HIT_sum"] / (ent["TCC_HIT_sum"] + ent["TCC_MISS_sum"])
        out[k] = ent
    from subphaser_amd._native import csrc_fingerprint
    json.dump({"unit": "per launch (averages over the launches of one bench step)", "kernels": out,
               # what was profiled: bench.py drops `traffic` when the kernel sources no longer hash to this
               "csrc_sha16": csrc_fingerprint(), "commit": os.environ.get("SP_COMMIT", "unknown")}, sys.stdout,
              indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
