#!/usr/bin/env python3
"""Timing of the filter kernels (k3_eval / k3_emit) on the wheat-like genome (dev tool).
SUBPHASER_HIP_LIB selects a library variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from subphaser_amd import _native
from subphaser_amd.config import sets_to_csr
from subphaser_amd.synth import SynthGenome
gen = SynthGenome(sys.argv[1] if len(sys.argv) > 1 else "wheat")
ctx = _native.Context(0)
ctx.genome_reset(len(gen.chroms))
for i, c in enumerate(gen.chroms):
    p = ctx.dev_alloc(c["length"])
    ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
    ctx.genome_add_device(i, p, c["length"])
    ctx.dev_free(p)
ctx.count(15, 3, 0)
print('overflow pairs per chromosome:', [ctx.table_overflow(i) for i in range(len(gen.chroms))])
csr = sets_to_csr(gen.sgs, gen.labels)
ctx.prof_enable(True)
for _ in range(3):
    try:
        print(ctx.filter(*csr, 2.0, 1, 200, 1e9, 1.0))
    except Exception as e:
        print("filter:", e)
ctx.prof_enable(False)
print(os.environ.get("SUBPHASER_HIP_LIB", "default"), {k: round(v["ms"] / v["calls"], 3) for k, v in ctx.prof_report().items()})
