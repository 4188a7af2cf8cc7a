#!/usr/bin/env python3
"""Dev tool: size / multiplicity statistics of the fine buckets of one synthetic chromosome at k > 15
(SP_S3_DUMP hook of s3_count_chrom).  usage: s3_bucket_stats.py [bases] [k]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 667_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 21
os.environ["SP_S3_DUMP"] = "/tmp/s3dump"
from subphaser_amd import _native
ctx = _native.Context(0)
d = ctx.dev_alloc(n)
ctx.synth_chrom(d, n, 3, 0, 0, 3, 0)
ctx.genome_reset(1)
ctx.genome_add_device(0, d, n)
ctx.count(K, 3, 0)
sp = np.fromfile("/tmp/s3dump.spans", np.uint64).reshape(-1, 2)
keys = np.fromfile("/tmp/s3dump.keys", np.uint32 if K <= 25 else np.uint64)
sz = sp[:, 1].astype(np.int64)
print("buckets", len(sz), "keys", sz.sum(), "mean", sz.mean())
edges = [0, 1, 512, 1024, 1536, 2048, 3072, 4096, 8192, 16384, 32768, 65536, 1 << 20, 1 << 40]
for a, b in zip(edges[:-1], edges[1:]):
    m = (sz >= a) & (sz < b)
    print("size [%d, %d): %d buckets, %d keys (%.2f %%)" % (a, b, m.sum(), sz[m].sum(), 100.0 * sz[m].sum() / sz.sum()))
big = np.flatnonzero(sz > 2048)
rng = np.random.default_rng(1)
pick = rng.choice(big, size=min(200, big.size), replace=False)
rows = []
for b in pick:
    k = keys[int(sp[b, 0]):int(sp[b, 0] + sp[b, 1])]
    u, c = np.unique(k, return_counts=True)
    rows.append((len(k), len(u), int((c >= 3).sum()), int(c.max()), int(c[c >= 3].sum()), int((c == 1).sum())))
rows = np.array(rows)
print("sampled big buckets: n, distinct, kept(>=3), max count, keys in kept, singletons")
for q in (10, 50, 90, 99):
    print("  pct %d:" % q, np.percentile(rows, q, axis=0).astype(int).tolist())
print("  distinct > 1536:", int((rows[:, 1] > 1536).sum()), "of", len(rows), "; distinct/n median %.2f" % np.median(rows[:, 1] / rows[:, 0]))
small = np.flatnonzero((sz > 0) & (sz <= 2048))
pick = rng.choice(small, size=300, replace=False)
rows = []
for b in pick:
    k = keys[int(sp[b, 0]):int(sp[b, 0] + sp[b, 1])]
    u, c = np.unique(k, return_counts=True)
    rows.append((len(k), len(u), int((c >= 3).sum()), int(c.max()), int(c[c >= 3].sum())))
rows = np.array(rows)
print("sampled small buckets: n, distinct, kept(>=3), max count, keys in kept")
for q in (10, 50, 90, 99):
    print("  pct %d:" % q, np.percentile(rows, q, axis=0).astype(int).tolist())
