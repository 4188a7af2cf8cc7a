#!/usr/bin/env python3
"""Host wall time and kernel time of sp_labels_set / sp_labels_set_device on the wheat-like label set (dev tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from subphaser_amd import _native
from subphaser_amd.seqs import KmerLabels
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_225_481
k = 15
rng = np.random.default_rng(1)
keys = np.unique(rng.integers(0, 1 << (2 * k), size=2 * n, dtype=np.uint64))[:n]
from subphaser_amd import kmer
keys = np.unique(kmer.canonical(keys, k))
sg = (np.arange(keys.size) % 3).astype(np.uint8)
ctx = _native.Context(0)
s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=100000)]
ctx.genome_reset(1); ctx.genome_add(0, s); ctx.count(k, 1, 1)
lab = KmerLabels(keys, sg, ["A", "B", "C"], k)
for name, f in (("host", lambda: ctx.labels_set(keys, sg, 3)), ("device", lambda: ctx.labels_set_from(lab, 3))):
    f(); ctx.sync()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    ctx.prof_enable(False)
    rep = ctx.prof_report()
    print(name, "wall %.3f ms" % (dt * 1e3), {k_: round(v["ms"] / 10, 3) for k_, v in rep.items()})
