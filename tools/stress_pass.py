#!/usr/bin/env python3
"""Repeat count -> filter -> labels -> map of a synthetic genome and compare every pass -- hashes of every chromosome's full
dump, of the matrix rows, totals, histogram and bin counts -- with a first pass that ran on ONE stream (dev / test tool: the
whole path must be deterministic in its results whatever runs next to it).  `lanes` forces that many chains in flight, `busy`
runs a second context on the same GPU (tools/gpu_busy.py).  Also the body of tests/test_gpu_parity.py::test_stress_pass_with_streams.
usage: stress_pass.py [config=wheat] [k=15] [iterations=30] [scale=1.0] [lanes=default] [busy=0]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from subphaser_amd import _native
from subphaser_amd.config import sets_to_csr
from subphaser_amd.synth import SynthGenome

LANE_VARS = ("SP_LANES_DENSE", "SP_LANES_SPARSE", "SP_LANES")


def _h(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]


def _set_lanes(n, saved):
    for v in LANE_VARS:
        if n is None:
            if saved[v] is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = saved[v]
        else:
            os.environ[v] = str(n)


def run(name, K, iters, scale, ctx, lanes=None, busy=False, dumps=None, verbose=True):
    from contextlib import nullcontext
    from gpu_busy import busy_neighbour
    gen = SynthGenome(name, scale)
    ctx.genome_reset(len(gen.chroms))
    for i, c in enumerate(gen.chroms):
        p = ctx.dev_alloc(c["length"])
        ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
        ctx.genome_add_device(i, p, c["length"])
        ctx.dev_free(p)
    csr = sets_to_csr(gen.sgs, gen.labels)
    if dumps is None:
        dumps = gen.total_bases <= 400_000_000      # full dumps travel to the host: reduced scales only

    def one():
        ctx.count(K, 3, 0)
        dh = [_h(np.concatenate([a.view(np.uint8) for a in ctx.dump(i)])) for i in range(len(gen.chroms))] if dumps else \
             [ctx.dump_size(i) for i in range(len(gen.chroms))]
        nu, nr, nh = ctx.filter(*csr, 2.0, 1, 200 * scale if scale < 1 else 200, 1e9, 1.0)
        keys, counts, _, tot = ctx.filter_fetch(nr, want_freqs=False)
        hist = np.sort(ctx.filter_hist(nh))
        sg = (np.arange(keys.size) % gen.S).astype(np.uint8)
        ctx.labels_set(keys, sg, gen.S)
        bins, nm = ctx.map_bins_all(10000, 1_000_000)
        return (ctx.lengths().tolist(), dh, nu, nr, nh, _h(keys), _h(counts), _h(tot), _h(hist), [_h(b) for b in bins],
                np.asarray(nm).tolist(), ctx.labels_hit())

    saved = {v: os.environ.get(v) for v in LANE_VARS}
    bad = 0
    try:
        _set_lanes(0, saved)
        ref = one()
        with (busy_neighbour() if busy else nullcontext()):
            for it in range(iters):
                _set_lanes(lanes if not isinstance(lanes, (list, tuple)) else lanes[it % len(lanes)], saved)
                got = one()
                if got != ref:
                    bad += 1
                    print("it=%d differs in fields %s" % (it, [i for i, (a, b) in enumerate(zip(got, ref)) if a != b]), flush=True)
    finally:
        _set_lanes(None, saved)
    if verbose:
        print("stress_pass %s k=%d x%g: %d passes, %d bad (rows %d)" % (name, K, scale, iters, bad, ref[3]))
    return bad, ref[3]


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "wheat"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    lanes = None if len(sys.argv) <= 5 or sys.argv[5] == "default" else [int(x) for x in sys.argv[5].split(",")]
    busy = len(sys.argv) > 6 and sys.argv[6] not in ("0", "")
    sys.exit(1 if run(name, K, iters, scale, _native.Context(0), lanes, busy)[0] else 0)
