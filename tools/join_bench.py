#!/usr/bin/env python3
"""experiment helper: time the filter stage alone on a synthetic genome (kernel times from the library's profiler)
    python tools/join_bench.py --config peanut [-k 15] [--engine 3] [--reps 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from subphaser_amd import _native                     # noqa: E402
from subphaser_amd.hotpath import HotPath             # noqa: E402
from subphaser_amd.synth import SynthGenome           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="peanut")
ap.add_argument("-k", type=int, default=15)
ap.add_argument("--engine", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
gen = SynthGenome(a.config, a.scale)
ctx = _native.Context(0)
ptrs = []
for c in gen.chroms:
    p = ctx.dev_alloc(c["length"])
    ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
    ptrs.append(p)
hp = HotPath(ctx, gen.labels, [c["length"] for c in gen.chroms], gen.sgs, k=a.k, engine=a.engine)
r = hp.count_and_filter(ptrs)
print("union %d rows %d hist %d" % (r.n_union, r.n_rows, r.n_hist))
ctx.prof_reset()
ctx.prof_enable(True)
for _ in range(a.reps):
    ctx.filter(*hp.csr, hp.min_fold, hp.baseline, hp.min_freq, hp.max_freq, hp.ratio)
ctx.prof_enable(False)
for name, st in sorted(ctx.prof_report().items(), key=lambda kv: -kv[1]["ms"]):
    print("%-20s %8.3f ms/call  x%d" % (name, st["ms"] / st["calls"], st["calls"] // a.reps))
