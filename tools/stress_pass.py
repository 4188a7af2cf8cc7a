#!/usr/bin/env python3
"""Repeat count -> filter -> labels -> map of a synthetic genome and compare every pass with the first (dev tool: the whole
path must be deterministic in its results).  usage: stress_pass.py [config] [k] [iterations] [scale]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from subphaser_amd import _native
from subphaser_amd.config import sets_to_csr
from subphaser_amd.synth import SynthGenome
name = sys.argv[1] if len(sys.argv) > 1 else "wheat"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 15
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
gen = SynthGenome(name, scale)
ctx = _native.Context(0)
ctx.genome_reset(len(gen.chroms))
for i, c in enumerate(gen.chroms):
    p = ctx.dev_alloc(c["length"])
    ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
    ctx.genome_add_device(i, p, c["length"])
    ctx.dev_free(p)
csr = sets_to_csr(gen.sgs, gen.labels)
def h(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
def one():
    ctx.count(K, 3, 0)
    nu, nr, nh = ctx.filter(*csr, 2.0, 1, 200 * scale if scale < 1 else 200, 1e9, 1.0)
    keys, counts, _, tot = ctx.filter_fetch(nr, want_freqs=False)
    hist = np.sort(ctx.filter_hist(nh))
    sg = (np.arange(keys.size) % gen.S).astype(np.uint8)
    ctx.labels_set(keys, sg, gen.S)
    bins, nm = ctx.map_bins_all(10000, 1_000_000)
    return (ctx.lengths().tolist(), nu, nr, nh, h(keys), h(counts), h(tot), h(hist), [h(b) for b in bins], np.asarray(nm).tolist(), ctx.labels_hit())
ref = one()
bad = 0
for it in range(iters):
    got = one()
    if got != ref:
        bad += 1
        print("it=%d differs in fields %s" % (it, [i for i, (a, b) in enumerate(zip(got, ref)) if a != b]), flush=True)
print("stress_pass %s k=%d x%g: %d passes, %d bad (rows %d)" % (name, K, scale, iters, bad, ref[2]))
