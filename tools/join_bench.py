#!/usr/bin/env python3
"""Timing of the list-join filter (sps_bounds / sps_join / tallies / placement) on a synthetic genome (dev tool).
usage: join_bench.py [config] [k] [engine];  SUBPHASER_HIP_LIB selects a library variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from subphaser_amd import _native
from subphaser_amd.config import sets_to_csr
from subphaser_amd.synth import SynthGenome
gen = SynthGenome(sys.argv[1] if len(sys.argv) > 1 else "peanut")
K = int(sys.argv[2]) if len(sys.argv) > 2 else 15
engine = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = _native.Context(0)
ctx.genome_reset(len(gen.chroms))
for i, c in enumerate(gen.chroms):
    p = ctx.dev_alloc(c["length"])
    ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
    ctx.genome_add_device(i, p, c["length"])
    ctx.dev_free(p)
ctx.count(K, 3, engine)
csr = sets_to_csr(gen.sgs, gen.labels)
ctx.filter(*csr, 2.0, 1, 200, 1e9, 1.0)
ctx.prof_enable(True)
for _ in range(3):
    res = ctx.filter(*csr, 2.0, 1, 200, 1e9, 1.0)
ctx.prof_enable(False)
print(res)
print(os.environ.get("SUBPHASER_HIP_LIB", "default"), {k: round(v["ms"] / v["calls"], 3) for k, v in ctx.prof_report().items()})
