#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the oracle (dev tool, needs a GPU):
random genomes (N runs, lower case, homopolymers, tandem repeats, empty / shorter-than-k chromosomes),
random k in 1..32, thresholds, engines, label sets, bin / chunk sizes and set structures.
Stream mode (round 6): every count runs with 3..7 chains in flight (SP_LANES_DENSE / SP_LANES_SPARSE / SP_LANES forced -- the
library keeps toy genomes on one stream by default, which is why 250 K iterations never met the round-5 `s3_part1` race), some
chromosomes are several tiles long, and a second context (tools/gpu_busy.py) competes for the CUs.
usage: fuzz_parity.py [iterations=200] [seed=0] [streams]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
from oracle_ctx import OracleContext
from subphaser_amd import _native
from subphaser_amd.config import sets_to_csr

ALPHA = np.frombuffer(b"ACGTacgtNRY", np.uint8)


def rand_chrom(rng, streams=False):
    kind = rng.randint(0, 10)
    if kind == 0:
        return np.empty(0, np.uint8)
    if kind == 1:
        return ALPHA[rng.randint(0, 4, size=rng.randint(1, 40))]
    n = int(rng.choice([200, 3000, 20000, 70000, 70000, 250000] if streams else [200, 3000, 20000, 70000]))
    p = np.array([.23, .23, .23, .23, .015, .015, .015, .015, .006, .002, .002])
    s = ALPHA[rng.choice(len(ALPHA), size=n, p=p / p.sum())].copy()
    for _ in range(rng.randint(0, 6)):                       # repeats, homopolymers, tandem arrays, N blocks
        a = rng.randint(0, max(1, n - 10)); m = rng.randint(1, max(2, n // 8))
        what = rng.randint(0, 4)
        if what == 0:
            s[a:a + m] = ord("N")
        elif what == 1:
            s[a:a + m] = ALPHA[rng.randint(0, 4)]
        elif what == 2:
            u = ALPHA[rng.randint(0, 4, size=rng.randint(1, 30))]
            s[a:a + m] = np.resize(u, min(m, n - a))
        else:
            b = rng.randint(0, max(1, n - m)); s[a:a + min(m, n - a)] = s[b:b + min(m, n - a)]
    return s

def _tr(what):
    if os.environ.get("SP_FUZZ_TRACE"):
        print("   " + what, file=sys.stderr, flush=True)


LANE_VARS = ("SP_LANES_DENSE", "SP_LANES_SPARSE", "SP_LANES", "SP_C2_BATCH")


def run(iters, seed, gpu, ora, verbose=True, streams=False):
    if not streams:
        return _run(iters, seed, gpu, ora, verbose, False)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gpu_busy import busy_neighbour
    saved = {n: os.environ.get(n) for n in LANE_VARS}
    from contextlib import nullcontext
    try:
        # (SP_FUZZ_NO_BUSY=1: several fuzz processes side by side share ONE neighbour -- tools/fuzz_parallel.sh)
        with (nullcontext() if os.environ.get("SP_FUZZ_NO_BUSY") else busy_neighbour()):
            return _run(iters, seed, gpu, ora, verbose, True)
    finally:
        for n, v in saved.items():
            if v is None:
                os.environ.pop(n, None)
            else:
                os.environ[n] = v


def _run(iters, seed, gpu, ora, verbose, streams):
    import time
    rng = np.random.RandomState(seed)
    bad = 0
    t0, limit = time.time(), float(os.environ.get("SP_FUZZ_SECONDS", "0"))      # (a time budget: stop early, report what ran)
    it = -1
    for it in range(iters):
        if limit and time.time() - t0 > limit:
            it -= 1
            break
        C = int(rng.randint(2, 7) if streams else rng.randint(1, 6))
        seqs = [rand_chrom(rng, streams) for _ in range(C)]
        if streams:      # the library reads these at every count
            for n in LANE_VARS[:3]:
                os.environ[n] = str(int(rng.randint(3, 8)))
            b = int(rng.randint(0, 3))
            if b == 2:
                os.environ.pop("SP_C2_BATCH", None)
            else:
                os.environ["SP_C2_BATCH"] = str(b)
        k = int(rng.choice([1, 2, 3, 5, 8, 11, 12, 13, 14, 15, 16, 17, 19, 21, 24, 27, 31, 32]))
        lower = int(rng.randint(1, 4))
        engine = int(rng.choice([0, 1, 2, 3])) if k <= 15 else int(rng.choice([0, 1]))      # 3: lists (k >= 9), else falls back
        if streams and k <= 15 and engine in (0, 1) and rng.randint(0, 2):
            engine = int(rng.choice([2, 3]))      # (engines 0 / 1 on toy genomes run no chains side by side)
        if streams and k > 15 and int(os.environ.get("SP_LANES_SPARSE", "3")) > 3:
            # (a k > 15 lane owns ~6.5 GB of partition buffers whatever the genome's size -- 15 GB with 64-bit residuals -- and they
            # are never given back: several fuzz processes side by side run out of a 288-GB device at seven lanes each)
            os.environ["SP_LANES_SPARSE"] = "3"
        tag = "it=%d C=%d k=%d L=%d eng=%d lens=%s%s" % (it, C, k, lower, engine, [len(s) for s in seqs],
                                                       " lanes=%s" % [os.environ.get(n) for n in LANE_VARS] if streams else "")
        if os.environ.get("SP_FUZZ_TRACE"):      # a GPU fault kills the process: leave the case on stderr first
            print(tag, file=sys.stderr, flush=True)
        try:
            for ctx in (gpu, ora):
                ctx.genome_reset(C)
                for i, s in enumerate(seqs):
                    ctx.genome_add(i, s)
            _tr("count")
            try:
                gpu.count(k, lower, engine)
            except ValueError as e:
                if engine in (2, 3) and "engine" in str(e).lower():
                    gpu.count(k, lower, 0)
                else:
                    raise
            ora.count(k, lower)
            assert gpu.lengths().tolist() == ora.lengths().tolist(), "lengths"
            _tr("dump")
            dumps = []
            for i in range(C):
                gk, gc = gpu.dump(i); ok, oc = ora.dump(i)
                assert gk.shape == ok.shape and (gk == ok).all() and (gc == oc).all(), "dump %d" % i
                dumps.append((gk, gc))
            allk = np.unique(np.concatenate([d[0] for d in dumps])) if C else np.empty(0, np.uint64)
            if allk.size:
                sel = allk[rng.rand(allk.size) < rng.choice([0.02, 0.3, 1.0])]
                S = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9]))      # > 7: the per-k-mer label engines
                sg = rng.randint(0, S, size=sel.size).astype(np.uint8)
                _tr("labels S=%d n=%d" % (S, sel.size))
                for ctx in (gpu, ora):
                    ctx.labels_set(sel, sg, S)
                _tr("map_bins")
                for i in range(C):
                    bs = int(rng.choice([1, 7, 100, 10000])); ch = int(rng.choice([0, 50, 1000, 10_000_000]))
                    g, gn = gpu.map_bins(i, bs, ch); o, on = ora.map_bins(i, bs, ch)
                    assert g.shape == o.shape and (g == o).all() and gn == on, "map %d bin=%d chunk=%d" % (i, bs, ch)
                assert gpu.labels_hit() == ora.labels_hit(), "labels_hit"
                _tr("features")
                feats = [seqs[rng.randint(0, C)][a:a + int(rng.randint(0, 400))] for a in rng.randint(0, 3000, size=5)]
                assert (gpu.map_features(feats) == ora.map_features(feats)).all(), "features"
                iv_c = rng.randint(0, C, size=12)
                iv_a = np.array([rng.randint(0, len(seqs[c]) + 1) for c in iv_c], np.int64)
                iv_b = np.array([rng.randint(a, len(seqs[c]) + 1) for c, a in zip(iv_c, iv_a)], np.int64)
                for ctx in (gpu, ora):
                    ctx.labels_set(sel, sg, S)
                _tr("intervals")
                gi, oi = gpu.map_intervals(iv_c, iv_a, iv_b), ora.map_intervals(iv_c, iv_a, iv_b)
                assert (gi == oi).all() and gpu.labels_hit() == ora.labels_hit(), "intervals"
            if C >= 2 and all(int(l) > 0 for l in ora.lengths()):
                perm = rng.permutation(C).tolist()
                cut = sorted(rng.choice(range(1, C), size=min(C - 1, rng.randint(1, 3)), replace=False).tolist()) if C > 2 else [1]
                units = [perm[a:b] for a, b in zip([0] + cut, cut + [C])]
                sgs = [[u for u in units]] if len(units) >= 2 else [[[perm[0]], perm[1:]]]
                args = (float(rng.choice([1.0, 1.5, 2.0, 3.0])), int(rng.choice([1, -1])), float(rng.choice([1, 3, 20])), 1e9,
                        float(rng.choice([0.5, 1.0])))
                csr = sets_to_csr(sgs, list(range(C)))
                _tr("filter %s %s" % (sgs, args))
                res = []
                for ctx in (gpu, ora):
                    try:
                        nu, nr, nh = ctx.filter(*csr, *args)
                        kk, cc, ff, tt = ctx.filter_fetch(nr)
                        res.append((nu, nr, nh, kk, cc, ff, tt, np.sort(ctx.filter_hist(nh))))
                    except ValueError as e:
                        res.append(("err", str(e)[:40]))
                if res[0][0] == "err" or res[1][0] == "err":
                    assert res[0][0] == res[1][0], "filter error mismatch %s %s" % (res[0], res[1])
                else:
                    assert res[0][:3] == res[1][:3], "filter counts %s %s" % (res[0][:3], res[1][:3])
                    for a, b in zip(res[0][3:], res[1][3:]):
                        assert a.shape == b.shape and (a == b).all(), "filter rows"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", tag, "->", e)
            if bad >= 5:
                break
    if verbose:
        print("fuzz%s seed %d: %d iterations, %d mismatches" % (" (streams)" if streams else "", seed, it + 1, bad), flush=True)
    return bad


if __name__ == "__main__":
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    st = len(sys.argv) > 3 and sys.argv[3] == "streams"
    sys.exit(1 if run(n_it, sd, _native.Context(0), OracleContext(nthreads=4), streams=st) else 0)
