#!/usr/bin/env python3
"""What would a locality-addressed pair filter cost in false positives?  (round 6, review item 2; needs a GPU for the label set)
Takes the differential k-mers of a synthetic genome (the label set the map stage works with), and simulates on the host
  today   the blocked Bloom filter of sp_map.h: block = one 32-bit word addressed by the hashed (k-1)-mer, 3 bits
  core-B  block of B bits addressed by the hashed canonical (k-3)-mer CORE the two pairs of a quad share, `nb` bits per (k-1)-mer
          taken from the hashed (k-1)-mer: ONE gather answers both pairs, every (k-1)-mer is entered under both of its cores
  core5   the same with the (k-5)-mer core three pairs share (one gather per six starts, three insertions per (k-1)-mer)
at several filter sizes: fill, block-load skew, false-positive rate per probed (k-1)-mer and per probed unit (quad / sextet) on
uniform random sequence (what the genome is outside the planted repeats).
usage: filter_sim.py [config=wheat] [scale=1.0] [k=15]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

U = np.uint64


def revcomp(x, k):
    x = ~x
    x = ((x >> U(2)) & U(0x3333333333333333)) | ((x & U(0x3333333333333333)) << U(2))
    x = ((x >> U(4)) & U(0x0F0F0F0F0F0F0F0F)) | ((x & U(0x0F0F0F0F0F0F0F0F)) << U(4))
    x = ((x >> U(8)) & U(0x00FF00FF00FF00FF)) | ((x & U(0x00FF00FF00FF00FF)) << U(8))
    x = ((x >> U(16)) & U(0x0000FFFF0000FFFF)) | ((x & U(0x0000FFFF0000FFFF)) << U(16))
    x = (x >> U(32)) | (x << U(32))
    return x >> U(64 - 2 * k)


def canon(x, k):
    return np.minimum(x, revcomp(x, k))


def h32(x, mult):
    x = (x ^ (x >> U(32))) & U(0xFFFFFFFF)
    x = (x * U(mult)) & U(0xFFFFFFFF)
    x ^= x >> U(15)
    x = (x * U(0x2C1B3C6D)) & U(0xFFFFFFFF)
    x ^= x >> U(12)
    return x


def bits_of(x, nb, B):
    """nb bit positions inside a block of B bits from the hashed key -> one mask per key (as B/64 uint64 columns)"""
    cols = B // 64
    m = np.zeros((x.size, cols), U)
    for i in range(nb):
        pos = h32(x, (0x85EBCA6B + 2 * i * 0x9E3779B1) & 0xFFFFFFFF) % U(B)
        bit = U(1) << (pos % U(64))
        for col in range(cols):
            m[:, col] |= np.where(pos // U(64) == U(col), bit, U(0))
    return m


def labels_of(config, scale, k):
    from subphaser_amd import _native
    from subphaser_amd.config import sets_to_csr
    from subphaser_amd.synth import SynthGenome
    gen = SynthGenome(config, scale)
    ctx = _native.Context(0)
    ctx.genome_reset(len(gen.chroms))
    for i, c in enumerate(gen.chroms):
        p = ctx.dev_alloc(c["length"])
        ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
        ctx.genome_add_device(i, p, c["length"])
        ctx.dev_free(p)
    ctx.count(k, 3, 0)
    nu, nr, nh = ctx.filter(*sets_to_csr(gen.sgs, gen.labels), 2.0, 1, 200 * scale if scale < 1 else 200, 1e9, 1.0)
    keys = ctx.filter_fetch(nr, want_freqs=False)[0]
    ctx.close()
    return np.asarray(keys, U)


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "wheat"
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    if config == "fake":       # (no GPU: runs of neighbouring k-mers of random sequence, to try the tool)
        rs = np.random.RandomState(1)
        n_runs = int(200000 * scale)
        seq = rs.randint(0, 4, size=(n_runs, k + 9)).astype(U)
        ks_ = []
        for off in range(10):
            v = np.zeros(n_runs, U)
            for j in range(k):
                v = (v << U(2)) | seq[:, off + j]
            ks_.append(canon(v, k))
        keys = np.unique(np.concatenate(ks_))
    else:
        keys = labels_of(config, scale, k)
    n = keys.size
    m1 = U((1 << (2 * (k - 1))) - 1)
    # the (k-1)-mers of the labelled k-mers, canonical, distinct
    X = np.unique(np.concatenate([canon(keys >> U(2), k - 1), canon(keys & m1, k - 1)]))
    print("%s x%g k=%d: %d labelled k-mers, %d distinct canonical (k-1)-mers (%.2f per k-mer)" % (config, scale, k, n, X.size, X.size / n))
    rng = np.random.RandomState(7)
    NP = 4_000_000

    def cores_of(x, kk, core):      # the canonical cores of length `core` at the offsets a unit can share (0, 2, .. kk - core)
        out = []
        for off in range(0, kk - core + 1, 2):
            c = (x >> U(2 * (kk - core - off))) & U((1 << (2 * core)) - 1)
            out.append(canon(c, core))
        return out

    # ---- today: 32-bit word per (k-1)-mer, 3 bits
    for bits in (24, 25):
        words = 1 << (bits - 5)
        w = (h32(X, 0x9E3779B1) % U(words)).astype(np.int64)
        mk = bits_of(X, 3, 64)[:, 0]
        mk = (mk | (mk >> U(32))) & U(0xFFFFFFFF)      # three positions in a 32-bit word: the 64-bit mask folded
        F = np.zeros(words, U)
        np.bitwise_or.at(F, w, mk)
        fill = sum(bin(int(v)).count("1") for v in F[:: max(1, words // 65536)]) / (32.0 * len(F[:: max(1, words // 65536)]))
        q = rng.randint(0, 1 << 62, size=NP).astype(U) & m1
        q = canon(q, k - 1)
        qm = bits_of(q, 3, 64)[:, 0]
        qm = (qm | (qm >> U(32))) & U(0xFFFFFFFF)
        hit = (F[(h32(q, 0x9E3779B1) % U(words)).astype(np.int64)] & qm) == qm
        fp = hit.mean()
        print("today   2^%d bits (%4.1f MiB), word-blocked, 3 bits:            fill %.3f  FP per (k-1)-mer %.4f  FP per quad %.4f  [2 gathers per quad]"
              % (bits, (1 << bits) / 8 / 2 ** 20, fill, fp, 1 - (1 - fp) ** 2))

    # ---- core-addressed blocks
    for core, per_unit, name in ((k - 3, 2, "core-%d  (quad)  "), (k - 5, 3, "core5-%d (sextet)")):
        cs = cores_of(X, k - 1, core)      # per (k-1)-mer: its cores at offsets 0, 2 (, 4)
        allc = np.unique(np.concatenate(cs))
        print("   %d-mer cores: %d distinct (%.2f per (k-1)-mer); insertions = %d" % (core, allc.size, allc.size / X.size, len(cs) * X.size))
        for B in (64, 128):
            for nb in (3, 4):
                for bits in (24, 25, 26, 27):
                    blocks = (1 << bits) // B
                    F = np.zeros((blocks, B // 64), U)
                    mk = bits_of(X, nb, B)
                    load = np.zeros(blocks, np.int64)
                    for c in cs:
                        b = (h32(c, 0x9E3779B1) % U(blocks)).astype(np.int64)
                        for col in range(B // 64):
                            np.bitwise_or.at(F[:, col], b, mk[:, col])
                        np.add.at(load, b, 1)
                    sample = F[:: max(1, blocks // 65536)]
                    fill = sum(bin(int(v)).count("1") for v in sample.ravel()) / (64.0 * sample.size)
                    # random units: a window of core + 2 * per_unit bases; its per_unit (k-1)-mers all contain the core
                    wlen = core + 2 * per_unit
                    win = (rng.randint(0, 1 << 62, size=NP).astype(U)) & U((1 << (2 * wlen)) - 1)
                    # (quad: x1 = bases [0, k-1), x2 = [2, k+1): core = [2, k-1).  sextet: x1 [0,k-1), x2 [2,k+1), x3 [4,k+3): core [4,k-1))
                    if per_unit == 2:
                        xs = [(win >> U(4)) & m1, win & m1]
                        cq = canon((win >> U(4)) & U((1 << (2 * core)) - 1), core)
                    else:
                        xs = [(win >> U(8)) & m1, (win >> U(4)) & m1, win & m1]
                        cq = canon((win >> U(8)) & U((1 << (2 * core)) - 1), core)
                    bq = (h32(cq, 0x9E3779B1) % U(blocks)).astype(np.int64)
                    anyhit = np.zeros(NP, bool)
                    fps = []
                    for x in xs:
                        qm = bits_of(canon(x, k - 1), nb, B)
                        ok = np.ones(NP, bool)
                        for col in range(B // 64):
                            ok &= (F[bq, col] & qm[:, col]) == qm[:, col]
                        fps.append(ok.mean())
                        anyhit |= ok
                    print((name % B) + " 2^%d bits (%4.1f MiB), %d bits per entry: fill %.3f  load mean %.1f p99 %d max %d  FP per (k-1)-mer %.4f  FP per unit %.4f  [1 gather per %d starts]"
                          % (bits, (1 << bits) / 8 / 2 ** 20, nb, fill, load.mean(), int(np.percentile(load, 99)), int(load.max()),
                             float(np.mean(fps)), anyhit.mean(), 2 * per_unit))


if __name__ == "__main__":
    main()
