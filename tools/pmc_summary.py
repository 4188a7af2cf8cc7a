#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
as MI355X_MICROARCH.md prescribes).  Units: the counters are KiB.  gfx950 correction: FETCH_SIZE
reports half of the bytes of a wide coalesced read stream -> x2 for the streaming kernels
(calibrated here on k0_pack, which reads exactly 1 B/base and writes 0.375 B/base: FETCH x 2 and
WRITE x 1 reproduce both).  Gather-dominated kernels (k5_map*) issue 64-B requests for single
bytes; their FETCH_SIZE is left uncorrected (x1) and flagged.

usage: pmc_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> > traffic.json
Everything is reported per kernel launch (per call)."""
import collections
import csv
import json
import sys

GATHER = ("k5_map", "k5_map_sparse", "k5_map_feat", "k5_map_feat_sparse")


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main(root):
    f = load(root + "/pmc_FETCH_SIZE/wheat_counter_collection.csv", "FETCH_SIZE")
    w = load(root + "/pmc_WRITE_SIZE/wheat_counter_collection.csv", "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        calls = max(f.get(k, [0, 0])[0], w.get(k, [0, 0])[0])
        corr = 1.0 if k in GATHER else 2.0
        rd = f.get(k, [0, 0.0])[1] * 1024 * corr
        wr = w.get(k, [0, 0.0])[1] * 1024
        out[k] = {"calls": calls, "read_bytes_per_call": rd / max(calls, 1), "write_bytes_per_call": wr / max(calls, 1),
                  "fetch_correction": corr}
    json.dump({"unit": "bytes per launch", "kernels": out}, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
