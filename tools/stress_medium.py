#!/usr/bin/env python3
"""Repeat the medium-scale parity scenario of tests/test_gpu_parity.py on ONE context, cycling k: full dumps, matrix rows,
histogram totals and bin counts of every pass against the oracle (hunts state-dependent / timing-dependent mismatches).
`lanes` forces that many chains in flight (SP_LANES_*), `busy` runs a second context next to it (tools/gpu_busy.py).
Also the body of tests/test_gpu_parity.py::test_stress_medium_with_streams.
usage: stress_medium.py [iterations=30] [ks=22,17,22,15] [lanes=default] [busy=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_ctx import OracleContext
from subphaser_amd import _native
from subphaser_amd.config import sets_to_csr

def rand_seq(rng, n, p_n=0.0, p_lower=0.0):
    s = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, n)].copy()
    if p_n:
        s[rng.rand(n) < p_n] = ord("N")
    if p_lower:
        m = rng.rand(n) < p_lower
        s[m] = s[m] | 0x20
    return s

MAPS = ((10000, 10_000_000), (10000, 1_000_000), (333, 50_000))


def build_case():
    rng = np.random.RandomState(2024)
    fams = [[rand_seq(rng, 600) for _ in range(8)] for _ in range(2)]
    seqs = []
    for c in range(3):
        s = rand_seq(rng, 6_000_000 + 4099 * c, 0.0005, 0.1)
        lib = fams[c % 2]
        for _ in range(1500):
            r = lib[rng.randint(0, len(lib))].copy()
            mut = rng.rand(r.size) < 0.02
            r[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, int(mut.sum()))]
            p = rng.randint(0, s.size - 700)
            s[p:p + r.size] = r
        seqs.append(s)
    return seqs, sets_to_csr([[[0], [1], [2]]], [0, 1, 2])


def oracle_ref(seqs, csr, k):
    o = OracleContext()
    o.genome_reset(3)
    for i, s in enumerate(seqs):
        o.genome_add(i, s)
    o.count(k, 3, 0)
    nu, nr, nh = o.filter(*csr, 2.0, 1, 50, 1e9, 1.0)
    R = dict(lengths=o.lengths().tolist(), dumps=[o.dump(i) for i in range(3)], f=(nu, nr, nh), rows=o.filter_fetch(nr),
             hist=np.sort(o.filter_hist(nh)))
    keys = R["rows"][0]
    sg = (np.arange(keys.size) % 2).astype(np.uint8)
    o.labels_set(keys, sg, 2)
    R["maps"] = {(i, bs, ch): o.map_bins(i, bs, ch) for i in range(3) for bs, ch in MAPS}
    R["hit"] = o.labels_hit()
    return R


def check(gpu, seqs, csr, k, R, reload=True):
    """one GPU pass against the oracle's; returns the list of what differs (empty: identical)"""
    if reload:
        gpu.genome_reset(3)
        for i, s in enumerate(seqs):
            gpu.genome_add(i, s)
    gpu.count(k, 3, 0)
    what = []
    if gpu.lengths().tolist() != R["lengths"]:
        what.append("lengths %s vs %s" % (gpu.lengths().tolist(), R["lengths"]))
    for i in range(3):
        gk, gc = gpu.dump(i)
        ok, oc = R["dumps"][i]
        if gk.shape != ok.shape or not (gk == ok).all() or not (gc == oc).all():
            what.append("dump %d: %d vs %d entries" % (i, gk.size, ok.size))
            gs = set(zip(gk.tolist(), gc.tolist())); os_ = set(zip(ok.tolist(), oc.tolist()))
            extra = sorted(gs - os_)[:12]; missing = sorted(os_ - gs)[:12]
            sh = max(0, 2 * k - 19)
            what.append("extra " + " ".join("%x:%d(b%d)" % (a, b, a >> sh) for a, b in extra))
            what.append("missing " + " ".join("%x:%d(b%d)" % (a, b, a >> sh) for a, b in missing))
    nu, nr, nh = gpu.filter(*csr, 2.0, 1, 50, 1e9, 1.0)
    if (nu, nr, nh) != R["f"]:
        what.append("filter %s vs %s" % ((nu, nr, nh), R["f"]))
    else:
        rows = gpu.filter_fetch(nr)
        for a, b in zip(rows, R["rows"]):
            if not (a == b).all():
                what.append("rows differ")
                break
        if not (np.sort(gpu.filter_hist(nh)) == R["hist"]).all():
            what.append("hist differs")
    if not what:
        keys = R["rows"][0]
        sg = (np.arange(keys.size) % 2).astype(np.uint8)
        gpu.labels_set(keys, sg, 2)
        for (i, bs, ch), (o_, on) in R["maps"].items():
            g, gn = gpu.map_bins(i, bs, ch)
            if g.shape != o_.shape or not (g == o_).all() or gn != on:
                what.append("map_bins %d %d %d" % (i, bs, ch))
        allb, nm = gpu.map_bins_all(10000, 10_000_000)
        for i in range(3):
            o_, on = R["maps"][(i, 10000, 10_000_000)]
            if not (allb[i] == o_).all() or int(nm[i]) != on:
                what.append("map_bins_all %d" % i)
        if gpu.labels_hit() != R["hit"]:
            what.append("labels_hit %d vs %d" % (gpu.labels_hit(), R["hit"]))
    return what


LANE_VARS = ("SP_LANES_DENSE", "SP_LANES_SPARSE", "SP_LANES")


class forced_lanes:
    """`with forced_lanes(n):` -- n chains in flight whatever the genome's size (None: the library's defaults)"""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.saved = {v: os.environ.get(v) for v in LANE_VARS}
        if self.n is not None:
            for v in LANE_VARS:
                os.environ[v] = str(self.n)

    def __exit__(self, *exc):
        for v, x in self.saved.items():
            if x is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = x
        return False


def run(iters, ks, gpu, lanes=None, busy=False, refs=None, verbose=True):
    from contextlib import nullcontext
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gpu_busy import busy_neighbour
    seqs, csr = build_case()
    refs = {} if refs is None else refs
    for k in sorted(set(ks)):
        if k not in refs:
            refs[k] = oracle_ref(seqs, csr, k)
            if verbose:
                print("oracle k=%d done" % k, flush=True)
    bad = 0
    with (busy_neighbour() if busy else nullcontext()):
        for it in range(iters):
            k = ks[it % len(ks)]
            with forced_lanes(lanes if not isinstance(lanes, (list, tuple)) else lanes[it % len(lanes)]):
                what = check(gpu, seqs, csr, k, refs[k])
            if what:
                bad += 1
                print("it=%d k=%d MISMATCH: %s" % (it, k, "; ".join(what)), flush=True)
    if verbose:
        print("stress: %d iterations, %d bad" % (iters, bad))
    return bad


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "22,17,22,15").split(",")]
    lanes = None if len(sys.argv) <= 3 or sys.argv[3] == "default" else [int(x) for x in sys.argv[3].split(",")]
    busy = len(sys.argv) > 4 and sys.argv[4] not in ("0", "")
    sys.exit(1 if run(iters, ks, _native.Context(0), lanes, busy) else 0)
