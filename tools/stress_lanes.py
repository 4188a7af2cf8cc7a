#!/usr/bin/env python3
"""Repeat the count of a synthetic genome with the default streams and compare every chromosome's length sum, dump size and -- at
reduced scales, where the dumps can travel -- a hash of its FULL dump with a single-stream reference count (dev tool: hunts
timing-dependent miscounts).  usage: stress_lanes.py [config] [k] [iterations] [scale]"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from subphaser_amd import _native
from subphaser_amd.synth import SynthGenome
name = sys.argv[1] if len(sys.argv) > 1 else "wheat"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 19
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
gen = SynthGenome(name, scale)
ctx = _native.Context(0)
ctx.genome_reset(len(gen.chroms))
for i, c in enumerate(gen.chroms):
    p = ctx.dev_alloc(c["length"])
    ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
    ctx.genome_add_device(i, p, c["length"])
    ctx.dev_free(p)
# the reference count runs on ONE stream: SP_LANES_SPARSE (k > 15), SP_LANES_DENSE (byte tables), SP_LANES + SP_C2_BATCH (lists)
names = ["SP_LANES_SPARSE"] if K > 15 else ["SP_LANES_DENSE", "SP_LANES", "SP_C2_BATCH"]
saved = {n: os.environ.pop(n, None) for n in names}
for n in names:
    os.environ[n] = "0"
full = scale <= 0.05
def dump_hashes():
    if not full:
        return []
    return [hashlib.sha1(np.concatenate([a.view(np.uint8) for a in ctx.dump(i)]).tobytes()).hexdigest()[:12] for i in range(len(gen.chroms))]
ctx.count(K, 3, 0)
ref = (ctx.lengths().tolist(), [ctx.dump_size(i) for i in range(len(gen.chroms))], dump_hashes())
ref_dumps = [ctx.dump(i) for i in range(len(gen.chroms))] if full and K > 15 else None
for n in names:
    if saved[n] is None:
        del os.environ[n]
    else:
        os.environ[n] = saved[n]
bad = 0
for it in range(iters):
    ctx.count(K, 3, 0)
    got = (ctx.lengths().tolist(), [ctx.dump_size(i) for i in range(len(gen.chroms))], dump_hashes())
    if got != ref:
        bad += 1
        d = [i for i in range(len(gen.chroms)) if got[0][i] != ref[0][i] or got[1][i] != ref[1][i] or (full and got[2][i] != ref[2][i])]
        print("it=%d: chromosomes %s differ: %s vs %s" % (it, d, [(got[0][i], got[1][i]) for i in d], [(ref[0][i], ref[1][i]) for i in d]), flush=True)
        if ref_dumps is not None:
            for i in d:
                gk, gc = ctx.dump(i)
                rk, rc = ref_dumps[i]
                gs = dict(zip(gk.tolist(), gc.tolist())); rs = dict(zip(rk.tolist(), rc.tolist()))
                keys = sorted(set(gs) | set(rs))
                diff = [(k_, gs.get(k_, 0), rs.get(k_, 0)) for k_ in keys if gs.get(k_, 0) != rs.get(k_, 0)]
                sh2 = 2 * K - 19; sh1 = 2 * K - 10
                fines = sorted(set(k_ >> sh2 for k_, _, _ in diff)); l1 = sorted(set(k_ >> sh1 for k_, _, _ in diff))
                print("   chrom %d: %d keys differ, %d fine buckets, level-1 buckets %s; net %d" % (i, len(diff), len(fines), l1[:20], sum(a - b for _, a, b in diff)))
                print("   first: " + " ".join("%x:%d/%d" % t for t in diff[:10]))
                print("   fine buckets: %s" % fines[:40])
print("stress_lanes %s k=%d x%g: %d iterations, %d bad" % (name, K, scale, iters, bad))
