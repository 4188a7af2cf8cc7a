#!/usr/bin/env python3
"""Repeat the medium-scale parity scenario of tests/test_gpu_parity.py on ONE context, cycling k (dev tool: hunts
state-dependent / timing-dependent mismatches).  usage: stress_medium.py [iterations=30] [ks=22,17,22,15]"""
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

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "22,17,22,15").split(",")]
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
csr = sets_to_csr([[[0], [1], [2]]], [0, 1, 2])
MAPS = ((10000, 10_000_000), (10000, 1_000_000), (333, 50_000))
gpu = _native.Context(0)
ref = {}
for k in sorted(set(ks)):
    o = OracleContext()
    o.genome_reset(3)
    for i, s in enumerate(seqs):
        o.genome_add(i, s)
    o.count(k, 3, 0)
    nu, nr, nh = o.filter(*csr, 2.0, 1, 50, 1e9, 1.0)
    ref[k] = dict(lengths=o.lengths().tolist(), dumps=[o.dump(i) for i in range(3)], f=(nu, nr, nh), rows=o.filter_fetch(nr),
                  hist=np.sort(o.filter_hist(nh)))
    keys = ref[k]["rows"][0]
    sg = (np.arange(keys.size) % 2).astype(np.uint8)
    o.labels_set(keys, sg, 2)
    ref[k]["maps"] = {(i, bs, ch): o.map_bins(i, bs, ch) for i in range(3) for bs, ch in MAPS}
    ref[k]["hit"] = o.labels_hit()
    print("oracle k=%d done" % k, flush=True)
bad = 0
for it in range(iters):
    k = ks[it % len(ks)]
    R = ref[k]
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
            sh = 2 * k - 19
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
    if what:
        bad += 1
        print("it=%d k=%d MISMATCH: %s" % (it, k, "; ".join(what)), flush=True)
print("stress: %d iterations, %d bad" % (iters, bad))
