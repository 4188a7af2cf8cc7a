"""GPU suite (`-m gpu`): libsubphaser_hip.so through the C-ABI vs the CPU oracle
and the golden vectors.  Integer work is compared bit-exactly; p-values within
the 1e-6 the north star states (they agree to ~1e-9 in practice)."""
import os

import numpy as np
import pytest

import parity_cases as pc
import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["tables", "lists"])
def count_engine(request, monkeypatch):
    """k <= 15 counts live either in byte tables (engines 1 / 2 + the dense filter) or, for small genomes, in sorted
    (slot, count) lists (engine 3 + the list filter).  engine = 0 picks by table occupancy -- every toy genome of this
    suite would take the lists -- so the tests that go through the auto engine run once with each, forced."""
    monkeypatch.setenv("SP_LIST_ENGINE", "1" if request.param == "lists" else "0")
    return request.param


def _rand_seq(rng, n, p_other=0.01, lower=0.1):
    a = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, size=n)].copy()
    m = rng.random_sample(n) < lower
    a[m] |= 0x20
    m = rng.random_sample(n) < p_other
    a[m] = np.frombuffer(b"NRYKMSWnx-", np.uint8)[rng.randint(0, 10, size=int(m.sum()))]
    return a


def _count_both(ctx, seqs, k, lower, engine=1):
    ctx.genome_reset(len(seqs))
    for i, s in enumerate(seqs):
        ctx.genome_add(i, s)
    ctx.count(k, lower, engine)
    out = []
    for i, s in enumerate(seqs):
        gk, gc = ctx.dump(i)
        ok, oc = po.count(s, k, lower, nthreads=4)
        assert gk.shape == ok.shape, (k, i, gk.shape, ok.shape)
        assert (gk == ok).all() and (gc == oc).all(), (k, i)
        out.append((gk, gc))
    lens = ctx.lengths()
    assert lens.tolist() == [int(c.astype(np.int64).sum()) for _, c in out]
    return out


def test_library_is_native(gpu_ctx):
    import os
    from subphaser_amd import _native
    assert os.path.exists(_native.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libsubphaser_hip.so" in maps


def test_pack_roundtrip(gpu_ctx):
    rng = np.random.RandomState(1)
    seqs = [_rand_seq(rng, n) for n in (0, 1, 15, 31, 32, 33, 63, 64, 65, 1000, 4097, 100003)]
    gpu_ctx.genome_reset(len(seqs))
    for i, s in enumerate(seqs):
        gpu_ctx.genome_add(i, s)
    for i, s in enumerate(seqs):
        got = gpu_ctx.genome_unpack(i)
        up = s & 0xDF
        ok = (up == 65) | (up == 67) | (up == 71) | (up == 84)
        exp = np.where(ok, up, ord("N")).astype(np.uint8)
        assert got.shape == exp.shape and (got == exp).all(), i


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 9, 11, 12, 13, 14, 15])
def test_count_random_k(gpu_ctx, k):
    rng = np.random.RandomState(100 + k)
    seqs = [_rand_seq(rng, n) for n in (5000, 70001, 333)]
    _count_both(gpu_ctx, seqs, k, 1)
    _count_both(gpu_ctx, seqs, k, 3)


@pytest.mark.parametrize("k", [9, 10, 11, 12, 13, 14, 15])
def test_count_engine2_random_k(gpu_ctx, k):
    """LDS radix-partition engine vs the oracle (and, transitively, engine 1)."""
    rng = np.random.RandomState(200 + k)
    seqs = [_rand_seq(rng, n) for n in (5000, 1_200_001, 64, 333)]
    seqs.append(np.concatenate([np.frombuffer(b"A" * 30000, np.uint8), _rand_seq(rng, 50000),
                                np.frombuffer(b"TTTAGGG" * 5000, np.uint8)]))
    _count_both(gpu_ctx, seqs, k, 1, engine=2)
    _count_both(gpu_ctx, seqs, k, 3, engine=2)


def test_count_engine2_edges(gpu_ctx):
    seqs = [b"", b"ACGT", b"N" * 100, b"A" * 50, b"ACGTACGTACGTACGnACGTACGTACGTAC", b"TTTAGGG" * 3000,
            b"A" * 300000, (b"ACGT" * 20 + b"N") * 500]
    _count_both(gpu_ctx, seqs, 15, 1, engine=2)
    _count_both(gpu_ctx, seqs, 13, 2, engine=2)


@pytest.mark.parametrize("k", [13, 15])
def test_count_engine2_hot_slots(gpu_ctx, k):
    """Hot k-mers with 70 K - 400 K copies (beyond 16 bits): homopolymers of both strands' representatives, every
    rotation of a 13-mer unit (neighbouring slots both hot), a dinucleotide repeat, next to ordinary sequence; then
    one slot with 18 M copies.  (Written for a 16-bit packed-counter variant of c2_count that was not kept; the
    byte table + overflow list must carry these counts exactly whatever the LDS counter width.)"""
    rng = np.random.RandomState(77 + k)
    unit = _rand_seq(rng, 13, 0, 0)
    hot = [np.frombuffer(b"A" * 300000, np.uint8), np.frombuffer(b"C" * 140000, np.uint8), np.tile(unit, 70000),
           np.frombuffer(b"AC" * 200000, np.uint8), np.frombuffer(b"AAAAAAAAAAAAAAC" * 66000, np.uint8)]
    sep = np.frombuffer(b"N", np.uint8)
    parts = [_rand_seq(rng, 200000)]
    for h in hot:
        parts += [sep, h, sep, _rand_seq(rng, 1000)]
    seqs = [np.concatenate(parts), _rand_seq(rng, 5000)]
    _count_both(gpu_ctx, seqs, k, 1, engine=2)
    _count_both(gpu_ctx, seqs, k, 3, engine=2)
    seqs = [np.concatenate([_rand_seq(rng, 50000), sep, np.frombuffer(b"T" * 18_000_000, np.uint8), sep,
                            np.tile(unit, 70000)]), _rand_seq(rng, 5000)]
    _count_both(gpu_ctx, seqs, k, 2, engine=2)


def test_count_engine2_big_buckets_on_16_bit_counters(gpu_ctx):
    """Round 6: c2_count16 counts the fine buckets of 65536 records and more as well (their adds check for a wrapped half; a bucket
    with a wrapped counter is handed to c2_count) and writes the pairs of saturated slots (>= 255) straight to the bucket's segment
    of the staging list.  k = 9 on 34 Mb of random sequence: four fine buckets of ~8 M records, about half of their 32768 slots at
    255 or more; then the same with a 100 K-copy homopolymer (a wrapped counter) in one of them."""
    rng = np.random.RandomState(909)
    big = _rand_seq(rng, 34_000_000)
    _count_both(gpu_ctx, [big, _rand_seq(rng, 5000)], 9, 3, engine=2)
    big[1_000_000:1_100_000] = ord("A")
    _count_both(gpu_ctx, [big, _rand_seq(rng, 70_000)], 9, 1, engine=2)


def test_count_engine2_sampled_sizing_and_recount(gpu_ctx, monkeypatch):
    """Engine 2 lays its partition buckets out from a 1-in-16 stripe sample (round 3).  (a) on ordinary sequence
    nothing overruns and nothing is recounted; (b) with the slack removed and the multiplier halved (test hooks)
    buckets do overrun: the chromosome must be recounted from the exact histogram and the dump must still be the
    oracle's, bit for bit; (c) a local repeat array that the sample cannot see in proportion (a 25-kb array of one
    13-mer unit, a 300-kb homopolymer) is either absorbed by the slack or recounted -- exact either way; (d)
    SP_C2_EXACT=1 is the round-2 path."""
    rng = np.random.RandomState(4242)
    unit = _rand_seq(rng, 13, 0, 0)
    plain = [_rand_seq(rng, 3_000_001), _rand_seq(rng, 70_000)]
    before = gpu_ctx.count_recounts()
    _count_both(gpu_ctx, plain, 15, 3, engine=2)
    _count_both(gpu_ctx, plain, 13, 1, engine=2)
    assert gpu_ctx.count_recounts() == before                       # (a)
    monkeypatch.setenv("SP_C2_SLACK", "0")
    monkeypatch.setenv("SP_C2_MULT8", "64")                         # regions half of what the sample predicts
    _count_both(gpu_ctx, plain, 13, 1, engine=2)
    _count_both(gpu_ctx, plain, 11, 3, engine=2)
    assert gpu_ctx.count_recounts() >= before + 2                   # (b) both chromosomes, at least once
    monkeypatch.delenv("SP_C2_SLACK")
    monkeypatch.delenv("SP_C2_MULT8")
    arr = [np.concatenate([_rand_seq(rng, 400_000), np.tile(unit, 2000), _rand_seq(rng, 100_000),
                           np.frombuffer(b"A" * 300_000, np.uint8), _rand_seq(rng, 9_000), np.tile(unit[::-1], 1500)])]
    _count_both(gpu_ctx, arr, 15, 3, engine=2)                      # (c)
    _count_both(gpu_ctx, arr, 12, 1, engine=2)
    monkeypatch.setenv("SP_C2_EXACT", "1")
    n = gpu_ctx.count_recounts()
    _count_both(gpu_ctx, arr, 15, 2, engine=2)                      # (d)
    assert gpu_ctx.count_recounts() == n


@pytest.mark.parametrize("k", [9, 10, 12, 13, 14, 15])
def test_count_engine3_lists(gpu_ctx, k):
    """Engine 3 (small genomes): the partition chain ends in (slot, count >= lower) LISTS instead of byte tables.
    Dumps (converted back to canonical k-mers), lengths and dump sizes against the oracle; hot slots far beyond 255
    and beyond 16 bits; buckets with more kept slots than the LDS stage holds (k = 9: every slot of a bucket)."""
    rng = np.random.RandomState(300 + k)
    unit = _rand_seq(rng, 13, 0, 0)
    seqs = [_rand_seq(rng, n) for n in (5000, 1_200_001, 64, 333)]
    seqs.append(np.concatenate([np.frombuffer(b"A" * 300000, np.uint8), _rand_seq(rng, 50000), np.tile(unit, 70000),
                                np.frombuffer(b"TTTAGGG" * 5000, np.uint8), np.frombuffer(b"N", np.uint8),
                                np.frombuffer(b"AC" * 100000, np.uint8)]))
    seqs += [np.frombuffer(b"", np.uint8), np.frombuffer(b"N" * 100, np.uint8)]
    for lower in (1, 3):
        out = _count_both(gpu_ctx, seqs, k, lower, engine=3)
        assert [gpu_ctx.dump_size(i) for i in range(len(seqs))] == [len(kk) for kk, _ in out]
    with pytest.raises(Exception):
        gpu_ctx.count(5, 3, 3)            # 2^9 slots: no partition plan


@pytest.mark.parametrize("counters", ["16-bit", "32-bit"])
def test_count_engine3_list_counter_widths(gpu_ctx, monkeypatch, counters):
    """The list counter of small genomes (round 6): c2_count_list16 -- two 512-thread workgroups per CU on 16-bit counters -- hands
    a chromosome with a bucket of 65536 keys or more (300 K copies of one k-mer) back through the overrun flag, and the exact
    recount takes the 32-bit kernel; SP_C2_LIST16=0 runs the 32-bit kernel everywhere.  Batched (several chromosomes per launch) and
    per-chromosome chains, dumps against the oracle."""
    if counters == "32-bit":
        monkeypatch.setenv("SP_C2_LIST16", "0")
    rng = np.random.RandomState(4242)
    hot = np.concatenate([_rand_seq(rng, 40000), np.frombuffer(b"A" * 300000, np.uint8), _rand_seq(rng, 40000)])
    seqs = [_rand_seq(rng, 900_001), hot, _rand_seq(rng, 333), _rand_seq(rng, 250_000)]
    for batch in ("1", "0"):
        monkeypatch.setenv("SP_C2_BATCH", batch)
        r0 = gpu_ctx.count_recounts()
        _count_both(gpu_ctx, seqs, 15, 2, engine=3)
        if counters == "16-bit":
            assert gpu_ctx.count_recounts() > r0      # the hot chromosome went through the exact recount


def test_count_engine2_unsupported_small_k(gpu_ctx):
    gpu_ctx.genome_reset(1)
    gpu_ctx.genome_add(0, b"ACGT" * 100)
    with pytest.raises(ValueError):
        gpu_ctx.count(5, 1, 2)


@pytest.mark.parametrize("engine", [0, 1])
@pytest.mark.parametrize("k", [16, 17, 21, 22, 25, 26, 31, 32])
def test_count_sparse_engine(gpu_ctx, k, engine):
    """k > 15: 64-bit keys; engine 0 = MSD partition + in-LDS sort (sp_sparse2.hip), engine 1 = device-wide
    radix sort + run-length encode (sp_sparse.hip), both vs the oracle."""
    rng = np.random.RandomState(300 + k)
    seqs = [_rand_seq(rng, n) for n in (5000, 300_001, 64, 40)]
    rep = _rand_seq(rng, 400, 0, 0)
    s = _rand_seq(rng, 120000)
    for _ in range(60):
        p = rng.randint(0, s.size - 500)
        s[p:p + 400] = rep
    seqs += [s, np.frombuffer(b"A" * 5000 + b"TTTAGGG" * 800, np.uint8), np.empty(0, np.uint8),
             np.frombuffer(b"N" * 100, np.uint8)]
    _count_both(gpu_ctx, seqs, k, 1, engine)
    _count_both(gpu_ctx, seqs, k, 3, engine)


def test_count_sparse_engine_sampled_sizing_and_recount(gpu_ctx, monkeypatch):
    """k > 15 (round 3): the level-2 regions are laid out from a 1-in-8 sample of the level-1 buffer.  (a) sampled sizes
    on plain + repeat-rich sequence: exact results; (b) regions forced far too small (test hooks): keys are dropped, the
    flag goes up, the chromosome is counted again from the exact histogram -- same results, recounts reported;
    (c) SP_S3_EXACT=1 (the round-2 path) never recounts."""
    rng = np.random.RandomState(77)
    unit = _rand_seq(rng, 31, 0, 0)
    seqs = [np.concatenate([_rand_seq(rng, 300_000), np.tile(unit, 3000), _rand_seq(rng, 50_000)]), _rand_seq(rng, 180_000)]
    before = gpu_ctx.count_recounts()
    for k in (16, 17, 21, 32):
        _count_both(gpu_ctx, seqs, k, 2, 0)
    assert gpu_ctx.count_recounts() == before                       # (a)
    monkeypatch.setenv("SP_S3_MULT", "2")
    monkeypatch.setenv("SP_S3_SLACK", "0")
    for k in (17, 21, 32):
        _count_both(gpu_ctx, seqs, k, 1, 0)
    assert gpu_ctx.count_recounts() >= before + 3                   # (b)
    monkeypatch.delenv("SP_S3_MULT")
    monkeypatch.delenv("SP_S3_SLACK")
    # level 1 is sampled on chromosomes of 2^24 bases or more: plain, then with regions half the estimate
    big = [np.concatenate([_rand_seq(rng, 9_000_000), np.tile(unit, 200_000), _rand_seq(rng, 2_000_000),
                           np.frombuffer(b"N" * 5000, np.uint8), _rand_seq(rng, 300_000)])]
    assert big[0].size >= 1 << 24
    n0 = gpu_ctx.count_recounts()
    _count_both(gpu_ctx, big, 17, 3, 0)
    assert gpu_ctx.count_recounts() == n0
    monkeypatch.setenv("SP_S3_MULT1", "16")
    monkeypatch.setenv("SP_S3_SLACK1", "0")
    _count_both(gpu_ctx, big, 21, 3, 0)
    assert gpu_ctx.count_recounts() == n0 + 1
    monkeypatch.delenv("SP_S3_MULT1")
    monkeypatch.delenv("SP_S3_SLACK1")
    monkeypatch.setenv("SP_S3_EXACT", "1")
    n = gpu_ctx.count_recounts()
    _count_both(gpu_ctx, seqs, 17, 3, 0)                            # (c)
    assert gpu_ctx.count_recounts() == n


def test_count_sparse_engine_lanes_and_oversized_paths(gpu_ctx, monkeypatch):
    """round 4: (a) several chromosome chains in flight (three phases on SP_LANES_SPARSE streams) against one chain at a
    time; (b) every bucket above one sort's worth of keys forced through the oversized path (test hook): cut by hash
    class (s3_big_class + s3_big_sort) and, as the cross-check, through the library sort (SP_S3_BIG=sort); (c) a kept
    list beyond the class path's capacity must fall back to the library path.  All against the oracle."""
    rng = np.random.RandomState(404)
    unit = _rand_seq(rng, 29, 0, 0)
    seqs = [np.concatenate([_rand_seq(rng, 250_000), np.tile(unit, 4000), _rand_seq(rng, 40_000)]),
            _rand_seq(rng, 120_000), np.empty(0, np.uint8), _rand_seq(rng, 90_000),
            np.concatenate([np.tile(_rand_seq(rng, 3000, 0, 0), 40), _rand_seq(rng, 60_000)]), _rand_seq(rng, 30)]
    for lanes in ("0", "2", "3", "7"):
        monkeypatch.setenv("SP_LANES_SPARSE", lanes)
        for k in (16, 21, 27):
            _count_both(gpu_ctx, seqs, k, 2, 0)
    monkeypatch.delenv("SP_LANES_SPARSE")
    monkeypatch.setenv("SP_S3_FORCE_BIG", "1")
    for k in (17, 21, 26, 32):
        _count_both(gpu_ctx, seqs, k, 1, 0)
        _count_both(gpu_ctx, seqs, k, 3, 0)
    monkeypatch.setenv("SP_S3_BIG", "sort")
    for k in (17, 21, 32):
        _count_both(gpu_ctx, seqs, k, 2, 0)
    monkeypatch.delenv("SP_S3_BIG")
    monkeypatch.delenv("SP_S3_FORCE_BIG")
    # (c) ONE fine bucket with thousands of distinct residuals of count 2: blocks `AAAAAAAAAA + k - 10 random bases + N`
    # hold one valid k-mer each, all with the same leading 20 bits.  3000 distinct: more than the finish kernels' tables
    # take -> the hash classes; 6000: more kept pairs than the class path's list holds -> the library path
    for k in (21, 31):
        for distinct in (3000, 6000):
            body = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, size=(distinct, k - 10))]
            blocks = np.concatenate([np.full((distinct, 10), ord("A"), np.uint8), body, np.full((distinct, 1), ord("N"), np.uint8)], axis=1)
            seq = np.concatenate([blocks.reshape(-1), blocks.reshape(-1), _rand_seq(rng, 50_000)])
            _count_both(gpu_ctx, [seq, _rand_seq(rng, 20_000)], k, 2, 0)


def test_count_chains_side_by_side_at_k15(gpu_ctx, monkeypatch):
    """k <= 15: the byte-table chains (engine 2) and the list chains (engine 3) on several streams -- forced here, the
    default takes one stream below 2^24 bases -- against the oracle, with chromosomes that are empty, shorter than k,
    repeat-rich and N-rich in the same call"""
    rng = np.random.RandomState(1515)
    unit = _rand_seq(rng, 23, 0, 0)
    seqs = [np.concatenate([_rand_seq(rng, 150_000), np.tile(unit, 5000)]), _rand_seq(rng, 90_000, 0.05), np.empty(0, np.uint8),
            _rand_seq(rng, 9), _rand_seq(rng, 200_000), np.tile(_rand_seq(rng, 1, 0, 0), 70_000), _rand_seq(rng, 40_000)]
    for var, engine in (("SP_LANES_DENSE", 2), ("SP_LANES", 3)):
        for lanes in ("0", "1", "3", "7"):
            monkeypatch.setenv(var, lanes)
            for k in (11, 14, 15):
                _count_both(gpu_ctx, seqs, k, 2, engine)
        monkeypatch.delenv(var)
    # engine 3, round 5: one launch per kernel type and GROUP of chromosomes (sp_c2batch.h; the default below 2^26
    # bases) against the per-chromosome chains (SP_C2_BATCH=0), one to four groups, incl. the empty chromosome
    for batch in ("1", "0"):
        monkeypatch.setenv("SP_C2_BATCH", batch)
        for lanes in ("0", "3"):
            monkeypatch.setenv("SP_LANES", lanes)
            _count_both(gpu_ctx, seqs, 15, 3, 3)
            _count_both(gpu_ctx, seqs, 13, 1, 3)
    monkeypatch.delenv("SP_C2_BATCH")
    monkeypatch.delenv("SP_LANES")


def test_count_sparse_engine_hot_buckets(gpu_ctx):
    """Buckets beyond one workgroup's sort capacity (a k-mer repeated > 4096 times, and many distinct keys
    sharing the 18-19 partition bits) take the device-wide fallback of the MSD engine."""
    rng = np.random.RandomState(41)
    k = 17
    unit = _rand_seq(rng, 23, 0, 0)
    hot = np.tile(unit, 9000)                                   # 23 distinct k-mers x 9000 copies
    pre = _rand_seq(rng, 9, 0, 0)                               # many distinct k-mers behind one 9-base prefix
    crowd = np.concatenate([np.concatenate([pre, _rand_seq(rng, 8, 0, 0), np.frombuffer(b"N", np.uint8)])
                            for _ in range(12000)])
    seqs = [np.concatenate([_rand_seq(rng, 50000), hot, _rand_seq(rng, 1000), crowd])]
    _count_both(gpu_ctx, seqs, k, 1, 0)
    _count_both(gpu_ctx, seqs, k, 3, 0)
    # k = 16 / 17 finish their buckets with the bitmap kernel (<= 16 residual bits): few distinct residuals with
    # many copies stream through it (keys beyond the registers), > 2048 distinct residuals or > 2^22 keys in one
    # bucket are handed to the sort kernel / the device-wide fallback through kept[] = punt
    for k in (16, 17):
        polya = np.frombuffer(b"A" * 4_300_000, np.uint8)
        seqs = [np.concatenate([_rand_seq(rng, 30000), hot[:23 * 6000], _rand_seq(rng, 500), crowd[:18 * 9000], polya,
                                _rand_seq(rng, 200)])]
        _count_both(gpu_ctx, seqs, k, 1, 0)
        _count_both(gpu_ctx, seqs, k, 3, 0)
    # k = 32: 5000 distinct k-mers (each twice) behind one 14-base prefix share all partition and split
    # bits, so the piece exceeds the LDS hash table as well -> device-wide fallback for that bucket
    k = 32
    pre = np.frombuffer(b"AAAACAAAAGAAAC", np.uint8)
    tails = [_rand_seq(rng, 18, 0, 0) for _ in range(5000)]
    sep = np.frombuffer(b"N", np.uint8)
    crowd = np.concatenate([np.concatenate([pre, t, sep]) for t in tails + tails])
    seqs = [np.concatenate([_rand_seq(rng, 20000), sep, crowd, _rand_seq(rng, 3000)])]
    _count_both(gpu_ctx, seqs, k, 1, 0)
    _count_both(gpu_ctx, seqs, k, 2, 0)
    _count_both(gpu_ctx, seqs, k, 3, 0)


def test_list_filter_handful_of_kmers(gpu_ctx, oracle_ctx):
    """A fuzz find of round 3 (tools/fuzz_parity.py seed 9301, iteration 47): k = 32 and lists of 2 and 1 k-mers.  The
    join filter cut the 64-bit key space into ONE range, i.e. shifted keys right by 64 -- not a shift on the device --
    and wrote range edges out of bounds.  Also: the same lists through every k that leaves the key space 64 - 2k
    bits short of a machine word."""
    from subphaser_amd.config import sets_to_csr
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "fuzz_case_k32_join.npz"))
    seqs = [d["s0"], d["s1"]]
    for k, lower in ((32, 3), (32, 1), (31, 3), (24, 2), (17, 3)):
        for ctx in (gpu_ctx, oracle_ctx):
            ctx.genome_reset(2)
            for i, s in enumerate(seqs):
                ctx.genome_add(i, s)
            ctx.count(k, lower)
        assert gpu_ctx.lengths().tolist() == oracle_ctx.lengths().tolist()
        csr = sets_to_csr([[[0], [1]]], [0, 1])
        res = []
        for ctx in (gpu_ctx, oracle_ctx):
            nu, nr, nh = ctx.filter(*csr, 1.5, -1, 3.0, 1e9, 1.0)
            keys, counts, freqs, tot = ctx.filter_fetch(nr)
            res.append((nu, nr, nh, keys, counts, freqs, tot, np.sort(ctx.filter_hist(nh))))
        g, o = res
        assert g[:3] == o[:3], (k, lower, g[:3], o[:3])
        for a, b in zip(g[3:], o[3:]):
            assert a.shape == b.shape and (a == b).all(), (k, lower)


@pytest.mark.parametrize("map_engine", ["pairs", "per-kmer", "pairs+sort-filter", "pairs-hash", "pairs-crowded"])
@pytest.mark.parametrize("k", [17, 21, 32])
def test_filter_and_map_sparse_engine(gpu_ctx, oracle_ctx, k, map_engine, monkeypatch):
    """k = 17 / 21 (BASELINE config 5) and 32: matrix rows and bin counts bit-exact vs the oracle, with the
    pair-keyed label table (one look-up per pair of starts, <= 7 subgenomes; the rolled walk k5_map_sparse2) and with the
    per-k-mer table (k5_map_sparse_lab); the third variant runs the list filter's independent cross-check
    (SP_LIST_FILTER=sort: device-wide sort + run evaluation instead of the workgroup-per-range join sps_join_blk)."""
    if map_engine == "per-kmer":
        monkeypatch.setenv("SP_MAP_ENGINE", "1")
    if map_engine == "pairs+sort-filter":
        monkeypatch.setenv("SP_LIST_FILTER", "sort")
    # "pairs" = the quad-bucket table since round 6 (<= 3 subgenomes: one 32-byte bucket per candidate QUAD of starts);
    # "pairs-hash" = the pair-keyed hash table it replaced (SP_CTAB=0, still the table for 4..7 subgenomes); "pairs-crowded" = the
    # quad buckets at ~4 keys per bucket of four, so that many keys live in the overflow table
    if map_engine == "pairs-hash":
        monkeypatch.setenv("SP_CTAB", "0")
    if map_engine == "pairs-crowded":
        monkeypatch.setenv("SP_CTAB_FACTOR", "1")
    rng = np.random.RandomState(400 + k)
    reps = [_rand_seq(rng, 350, 0, 0) for _ in range(6)]
    seqs = []
    for c in range(6):
        s = _rand_seq(rng, 40000 + 500 * c)
        for _ in range(50):
            r = reps[rng.randint(0, 3) + (3 if c % 2 else 0)]
            p = rng.randint(0, s.size - 400)
            s[p:p + r.size] = r
        seqs.append(s)
    from subphaser_amd.config import sets_to_csr
    for ctx in (gpu_ctx, oracle_ctx):
        ctx.genome_reset(len(seqs))
        for i, s in enumerate(seqs):
            ctx.genome_add(i, s)
        ctx.count(k, 2)
    assert gpu_ctx.lengths().tolist() == oracle_ctx.lengths().tolist()
    sgs = [[[0], [1]], [[2], [3]], [[4], [5]]]
    csr = sets_to_csr(sgs, list(range(6)))
    res = []
    for ctx in (gpu_ctx, oracle_ctx):
        nu, nr, nh = ctx.filter(*csr, 2.0, 1, 10, 1e9, 1.0)
        keys, counts, freqs, tot = ctx.filter_fetch(nr)
        res.append((nu, nr, nh, keys, counts, freqs, tot, np.sort(ctx.filter_hist(nh))))
    g, o = res
    assert g[:3] == o[:3] and g[1] > 0
    for a, b in zip(g[3:], o[3:]):
        assert a.shape == b.shape and (a == b).all()
    keys, counts = g[3], g[4]
    # the same rows through the copy stream (the list engines' rows travel while the map stage runs since round 5)
    ak, ac, at = gpu_ctx.filter_fetch_async(g[1])
    gpu_ctx.filter_fetch_wait()
    oa, og = np.argsort(ak, kind="stable"), np.argsort(keys, kind="stable")
    assert (ak[oa] == keys[og]).all() and (ac[oa] == counts[og]).all() and (at[oa] == g[6][og]).all()
    sg = (counts[:, 1::2].sum(axis=1) > counts[:, 0::2].sum(axis=1)).astype(np.uint8)
    gpu_ctx.labels_set(keys, sg, 2)
    for i, s in enumerate(seqs):
        for bin_size, chunk in ((1000, 10000), (10000, 10_000_000), (64, 0)):
            got, n = gpu_ctx.map_bins(i, bin_size, chunk)
            exp, hit, n2 = po.map_bins(s, k, keys, sg, 2, bin_size, chunk, nthreads=2)
            assert (got == exp).all() and n == n2, (i, bin_size, chunk)
    allb, nm = gpu_ctx.map_bins_all(1000, 10000)
    assert int(nm.sum()) == sum(int(po.map_bins(s, k, keys, sg, 2, 1000, 10000)[2]) for s in seqs)
    assert gpu_ctx.labels_hit() > 0
    feats = [bytes(seqs[0][100:900]), bytes(seqs[1][5:2000]), b"ACGT", b""]
    fc = gpu_ctx.map_features(feats)
    for f, row in zip(feats, fc):
        if len(f) == 0:
            assert row.sum() == 0
            continue
        exp, _, _ = po.map_bins(f, k, keys, sg, 2, max(len(f), 1), 0)
        assert (row == exp.sum(axis=0)).all()
    gpu_ctx.count(15, 3, 1)      # back to the dense engine on the same context
    assert int(gpu_ctx.lengths()[0]) == int(po.count(seqs[0], 15, 3)[1].astype(np.int64).sum())


@pytest.mark.parametrize("variant", ["walk", "generic", "baseline2", "sort"])
def test_list_join_decision_paths(gpu_ctx, oracle_ctx, variant, monkeypatch):
    """The workgroup-per-range join (sps_join_blk) decides a k-mer with a uniform fp32 walk over row descriptors
    (baselines 1 / -1), with the generic fp64 code (SP_JOIN_GENERIC=1, or a baseline the walk does not cover), and the
    sort-based list filter stays as the independent cross-check: all bit-exact vs the oracle, on a set structure with units of several
    chromosomes, a singleton set, thresholds that keep hundreds of rows per round (several row chunks per round) and a
    ratio below one (k-mers missing from a set are decided, not screened out)."""
    if variant == "generic":
        monkeypatch.setenv("SP_JOIN_GENERIC", "1")
    if variant == "sort":
        monkeypatch.setenv("SP_LIST_FILTER", "sort")
    rng = np.random.RandomState(977)
    C = 11
    reps = [_rand_seq(rng, 4000, 0, 0) for _ in range(8)]
    seqs = []
    for c in range(C):
        s = _rand_seq(rng, 60000 + 700 * c)
        for _ in range(20 + 3 * (c % 3)):
            r = reps[rng.randint(0, 8)] if c % 2 else reps[rng.randint(0, 4)]
            p = rng.randint(0, s.size - r.size)
            s[p:p + r.size] = r
        seqs.append(s)
    from subphaser_amd.config import sets_to_csr
    sgs = [[[0, 1], [2], [3]], [[4], [5, 6]], [[7], [8], [9]], [[10]]]
    if variant == "baseline2":      # four units per set: baseline 2 is neither the second largest nor the smallest
        sgs = [[[0, 1], [2], [3], [4]], [[5], [6], [7, 8], [9]], [[10]]]
    csr = sets_to_csr(sgs, list(range(C)))
    for k in (17, 24):
        for ctx in (gpu_ctx, oracle_ctx):
            ctx.genome_reset(C)
            for i, s in enumerate(seqs):
                ctx.genome_add(i, s)
            ctx.count(k, 2)
        for fold, baseline, q, ratio in ((1.2, 1, 3, 1.0), (2.0, -1, 10, 0.5), (1.0, 2 if variant == "baseline2" else 1, 1, 0.6)):
            res = []
            for ctx in (gpu_ctx, oracle_ctx):
                nu, nr, nh = ctx.filter(*csr, fold, baseline, q, 1e9, ratio)
                keys, counts, freqs, tot = ctx.filter_fetch(nr)
                res.append((nu, nr, nh, keys, counts, freqs, tot, np.sort(ctx.filter_hist(nh))))
            g, o = res
            assert g[:3] == o[:3], (variant, k, fold, baseline, g[:3], o[:3])
            for a, b in zip(g[3:], o[3:]):
                assert a.shape == b.shape and (a == b).all(), (variant, k, fold, baseline)
        assert g[1] > 15000     # (the last configuration keeps ~50 rows per range on average: rounds of one and of two row chunks)


@pytest.mark.parametrize("layout", ["32 sets", "33 sets"])
def test_list_join_many_chromosomes(gpu_ctx, oracle_ctx, layout):
    """64 chromosomes (the list filter's limit): 32 non-singleton sets fill the join's 32-bit screen mask to its last bit; 33
    sets (chromosome 0 paired with each of 33 others) switch the screen off -- every owner is decided."""
    rng = np.random.RandomState(4242)
    C = 64
    reps = [_rand_seq(rng, 1500, 0, 0) for _ in range(6)]
    seqs = []
    for c in range(C):
        s = _rand_seq(rng, 6000 + 37 * c)
        for _ in range(3):
            r = reps[rng.randint(0, 6)] if c % 2 else reps[rng.randint(0, 3)]
            p = rng.randint(0, s.size - r.size)
            s[p:p + r.size] = r
        seqs.append(s)
    from subphaser_amd.config import sets_to_csr
    if layout == "32 sets":
        sgs = [[[2 * i], [2 * i + 1]] for i in range(32)]
    else:
        sgs = [[[0], [i]] for i in range(1, 34)] + [[[i]] for i in range(34, C)]
    csr = sets_to_csr(sgs, list(range(C)))
    for ctx in (gpu_ctx, oracle_ctx):
        ctx.genome_reset(C)
        for i, s in enumerate(seqs):
            ctx.genome_add(i, s)
        ctx.count(19, 1)
    for fold, baseline, q, ratio in ((1.5, 1, 4, 0.5), (1.0, -1, 1, 0.1)):
        res = []
        for ctx in (gpu_ctx, oracle_ctx):
            nu, nr, nh = ctx.filter(*csr, fold, baseline, q, 1e9, ratio)
            keys, counts, freqs, tot = ctx.filter_fetch(nr)
            res.append((nu, nr, nh, keys, counts, freqs, tot, np.sort(ctx.filter_hist(nh))))
        g, o = res
        assert g[:3] == o[:3], (layout, fold, g[:3], o[:3])
        for a, b in zip(g[3:], o[3:]):
            assert a.shape == b.shape and (a == b).all(), (layout, fold)
    assert g[1] > 100


def _tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(__file__), "..", "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("k", [15, 19])
def test_count_with_streams_equals_single_stream(k):
    """Tripwire for timing-dependent miscounts (tools/stress_lanes.py): 100 counts of a 21-chromosome synthetic genome with
    the default streams, each compared chromosome by chromosome with a single-stream count.  (Round 5: a missing barrier
    in s3_part1 miscounted one k > 15 pass in a few hundred when three chains were in flight; no parity test saw it.
    Round 6: 100 counts instead of 300 -- the oracle-compared loops below carry the weight now.)"""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_lanes.py")
    out = subprocess.run([sys.executable, tool, "wheat", str(k), "100", "0.01"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "100 iterations, 0 bad" in out.stdout, out.stdout[-2000:]


def test_stress_medium_with_streams(gpu_ctx):
    """tools/stress_medium.py in the suite (round 6): the medium-scale scenario (three 6-Mb chromosomes with planted repeat
    families) at k = 15 / 17 / 19 / 22, every pass with 3, 5 or 7 chains forced in flight and a second context busy on the
    same GPU, FULL dumps, matrix rows, histogram totals and bin counts against the oracle."""
    sm = _tool("stress_medium")
    assert sm.run(12, [15, 17, 19, 22], gpu_ctx, lanes=[3, 5, 7], busy=True, verbose=False) == 0


@pytest.mark.parametrize("k", [15, 17, 19, 22])
def test_stress_pass_with_streams(gpu_ctx, k):
    """tools/stress_pass.py in the suite (round 6): count -> filter -> labels -> map of the wheat-like genome at 1 / 100
    scale (21 chromosomes), six passes with 3 / 5 / 7 chains in flight and a busy neighbour, every chromosome's full dump, the
    matrix rows, totals, histogram and bins hashed and compared with a single-stream pass."""
    sp = _tool("stress_pass")
    bad, rows = sp.run("wheat", k, 6, 0.01, gpu_ctx, lanes=[3, 5, 7], busy=True, verbose=False)
    assert bad == 0 and rows > 100


def test_count_edges(gpu_ctx):
    k = 15
    seqs = [b"", b"ACGT", b"N" * 100, b"A" * 50, b"T" * 50, b"ACGTACGTACGTACnACGTACGTACGTAC",
            b"ACGTACGTACGTACGnACGTACGTACGTAC", b"acgtacgtacgtacgtacgtacgt", b"TTTAGGG" * 3000,
            b"A" * 100000, (b"ACGT" * 20 + b"N") * 500, b"ACGTACGTACGTACG", b"ACGTACGTACGTAC"]
    _count_both(gpu_ctx, seqs, k, 1)
    _count_both(gpu_ctx, seqs, k, 3)


def test_count_unit_boundaries(gpu_ctx):
    """N runs and valid windows placed around the 64-start unit and 16/32-base word edges."""
    rng = np.random.RandomState(5)
    base = _rand_seq(rng, 4096, p_other=0, lower=0)
    seqs = []
    for pos in (47, 48, 49, 62, 63, 64, 65, 78, 79, 80, 127, 128, 129, 4080, 4095):
        s = base.copy()
        s[pos] = ord("N")
        seqs.append(s)
    for n in (64 + 14, 64 + 15, 128 + 14, 128, 127, 129):
        seqs.append(base[:n].copy())
    _count_both(gpu_ctx, seqs, 15, 1)
    _count_both(gpu_ctx, seqs, 13, 1)


def test_count_deterministic(gpu_ctx):
    rng = np.random.RandomState(9)
    seqs = [_rand_seq(rng, 200000)]
    a = _count_both(gpu_ctx, seqs, 15, 2)
    b = _count_both(gpu_ctx, seqs, 15, 2)
    assert (a[0][0] == b[0][0]).all() and (a[0][1] == b[0][1]).all()


def test_count_rejects_bad_k(gpu_ctx):
    gpu_ctx.genome_reset(1)
    gpu_ctx.genome_add(0, b"ACGT" * 10)
    with pytest.raises(ValueError):
        gpu_ctx.count(0, 1)
    with pytest.raises(ValueError):
        gpu_ctx.count(33, 1)
    gpu_ctx.count(32, 1)         # the largest supported k


def test_toy_dumps(gpu_ctx, golden, toy, count_engine):
    pc.check_toy_dumps(gpu_ctx, golden, toy, engine=1)


def test_filter_cases(gpu_ctx, golden, toy, count_engine):
    pc.check_filter_cases(gpu_ctx, golden, toy)


def test_filter_vs_oracle_random(gpu_ctx, oracle_ctx, count_engine):
    """Random multi-chromosome genomes, random set structures: matrix rows bit-exact vs the oracle."""
    rng = np.random.RandomState(21)
    k = 11
    rep = [_rand_seq(rng, 300, 0, 0) for _ in range(6)]
    seqs = []
    for c in range(7):
        s = _rand_seq(rng, 30000 + 1000 * c)
        for _ in range(40):
            r = rep[rng.randint(0, 3) + (3 if c % 2 else 0)]
            p = rng.randint(0, s.size - 400)
            s[p:p + r.size] = r
        seqs.append(s)
    cfgs = [
        ([[[0], [1]], [[2], [3]], [[4], [5]], [[6]]], dict(min_fold=2, baseline=1, min_freq=10, max_freq=1e9, ratio=1)),
        ([[[0, 2], [1, 3]], [[4], [5], [6]]], dict(min_fold=1.5, baseline=-1, min_freq=5, max_freq=500, ratio=0.5)),
        ([[[0], [1], [2], [3], [4], [5], [6]]], dict(min_fold=3, baseline=3, min_freq=1, max_freq=1e9, ratio=1)),
    ]
    from subphaser_amd.config import sets_to_csr
    for ctx in (gpu_ctx, oracle_ctx):
        ctx.genome_reset(len(seqs))
        for i, s in enumerate(seqs):
            ctx.genome_add(i, s)
        ctx.count(k, 2, 1)
    assert gpu_ctx.lengths().tolist() == oracle_ctx.lengths().tolist()
    for sgs, kw in cfgs:
        csr = sets_to_csr(sgs, list(range(7)))
        res = []
        for ctx in (gpu_ctx, oracle_ctx):
            nu, nr, nh = ctx.filter(*csr, kw["min_fold"], kw["baseline"], kw["min_freq"], kw["max_freq"], kw["ratio"])
            keys, counts, freqs, tot = ctx.filter_fetch(nr)
            hist = np.sort(ctx.filter_hist(nh))
            res.append((nu, nr, nh, keys, counts, freqs, tot, hist))
        g, o = res
        assert g[:3] == o[:3], (sgs, g[:3], o[:3])
        for a, b in zip(g[3:], o[3:]):
            assert a.shape == b.shape and (a == b).all()
        assert g[1] > 0


def test_filter_near_threshold_screen(gpu_ctx, count_engine):
    """The fold test is screened in fp32 and decided in fp64 only near the threshold: hand-made count
    tables whose fold sits within 1e-9 .. 1e-3 (relative) of min_fold on both sides, exactly on it, and
    far from it, through the slot-range view (caller-owned tables + caller-given lengths)."""
    from subphaser_amd import kmer as km
    from subphaser_amd.config import sets_to_csr
    k, lower, C = 7, 1, 6
    n = km.dense_slots(k)
    rng = np.random.RandomState(77)
    base = 1_000_000_007
    # near-equal lengths so that small integer count ratios land a hair above / below the threshold
    lengths = np.array([base, base + 1, base + 1000, base - 3, 2 * base + 1, 2 * base - 1], np.int64)
    tabs = np.zeros((C, n), np.uint32)
    occ = rng.rand(C, n) < 0.5
    tabs[occ] = rng.randint(1, 50, size=int(occ.sum())).astype(np.uint32)
    # rows built to straddle fold == 2 and fold == 1.5: counts (2m, m), (3m, 2m), and +-1 perturbations
    for i in range(0, n, 3):
        m = int(rng.randint(1, 1 << 20))
        a, b = [(2 * m, m), (2 * m + 1, m), (2 * m - 1, m), (3 * m, 2 * m), (3 * m + 1, 2 * m), (4 * m, 2 * m + 1)][(i // 3) % 6]
        tabs[0, i], tabs[1, i] = a, b
        tabs[2, i], tabs[3, i] = b, a
        tabs[4, i], tabs[5, i] = 2 * a, 2 * b
    d_tabs, d_ovf, n_ovf = [], [], []     # byte tables + overflow pairs (most counts here are >= 255)
    for c in range(C):
        d_tabs.append(gpu_ctx.dev_alloc(n))
        gpu_ctx.host_to_dev(d_tabs[-1], np.minimum(tabs[c], 255).astype(np.uint8))
        big = np.flatnonzero(tabs[c] >= 255)
        pairs = np.stack([big.astype(np.uint32), tabs[c][big]], axis=1).astype(np.uint32)
        d_ovf.append(gpu_ctx.dev_alloc(max(8 * len(big), 8)))
        if len(big):
            gpu_ctx.host_to_dev(d_ovf[-1], np.ascontiguousarray(pairs))
        n_ovf.append(len(big))
    slots = np.arange(n, dtype=np.uint64)
    keys_all = km.keys_of_slots(slots, k)
    dumps = []
    for c in range(C):
        nz = np.flatnonzero(tabs[c] >= lower)
        kk = keys_all[nz]
        o = np.argsort(kk, kind="stable")
        dumps.append((kk[o], tabs[c][nz][o]))
    gpu_ctx.genome_reset(C)
    for c in range(C):
        gpu_ctx.genome_add(c, b"ACGTACGTACGT")
    gpu_ctx.count(k, lower, 1)
    for sgs, kw in (
        ([[[0], [1]], [[2], [3]], [[4], [5]]], dict(min_fold=2, baseline=1, min_freq=1, max_freq=1e12, ratio=1)),
        ([[[0], [1]], [[2], [3]], [[4], [5]]], dict(min_fold=1.5, baseline=-1, min_freq=1, max_freq=1e12, ratio=0.6)),
        ([[[0], [1], [2]], [[3], [4, 5]]], dict(min_fold=2, baseline=1, min_freq=10, max_freq=1e12, ratio=0.5)),
        ([[[0, 4], [1, 5]], [[2], [3]]], dict(min_fold=2.0000001, baseline=-1, min_freq=1, max_freq=1e12, ratio=1)),
    ):
        labels = list(range(C))
        exp = po.filter_dumps(dumps, sgs, labels, lengths=lengths, **kw)
        gpu_ctx.filter_view(d_tabs, 0, n, lengths, k, lower, d_ovf, n_ovf)
        try:
            nu, nr, nh = gpu_ctx.filter(*sets_to_csr(sgs, labels), kw["min_fold"], kw["baseline"], kw["min_freq"],
                                        kw["max_freq"], kw["ratio"])
            keys, counts, freqs, tot = gpu_ctx.filter_fetch(nr)
        finally:
            gpu_ctx.filter_view(None, 0, 0, None, 0, 0)
        assert (nu, nr, nh) == (exp.n_union, len(exp.keys), len(exp.hist)), (sgs, kw, nu, nr, nh, exp.n_union, len(exp.keys), len(exp.hist))
        assert (keys == exp.keys).all() and (counts == exp.counts).all()
        assert nr > 0 and nh > nr // 2
    for d in d_tabs + d_ovf:
        gpu_ctx.dev_free(d)


def test_map_pair_filter_adversarial(gpu_ctx):
    """The map pre-filter screens starts 2i and 2i+1 with one probe of their shared (k-1)-mer: labelled
    k-mers at even and odd starts, next to N runs, at the chromosome ends, as isolated single hits, on
    both strands, for odd and even k (dense) and 64-bit keys (sparse)."""
    rng = np.random.RandomState(5)
    for k in (2, 3, 8, 13, 15, 16, 21, 32):
        s = _rand_seq(rng, 6000, 0.02, 0.2)
        n_pos = s.size - k + 1
        gpu_ctx.genome_reset(1)
        gpu_ctx.genome_add(0, s)
        gpu_ctx.count(k, 1, 1)
        keys, cnts = gpu_ctx.dump(0)
        assert keys.size
        # label a scattered third of the k-mers -> isolated hits at both parities, plus the first/last ones
        sel = keys[rng.rand(keys.size) < 0.3]
        sg = (np.arange(sel.size) % 2).astype(np.uint8)
        gpu_ctx.labels_set(sel, sg, 2)
        for bin_size, chunk in ((1, 0), (7, 100), (1000, 0)):
            got, nmap = gpu_ctx.map_bins(0, bin_size, chunk)
            exp, hit, n2 = po.map_bins(s, k, sel, sg, 2, bin_size, chunk, nthreads=2)
            assert got.shape == exp.shape and (got == exp).all() and nmap == n2, (k, bin_size, chunk)
        assert gpu_ctx.labels_hit() == int(hit.sum())
        # one labelled k-mer only, then none
        gpu_ctx.labels_set(sel[:1], sg[:1], 2)
        got, nmap = gpu_ctx.map_bins(0, 1, 0)
        exp, hit, n2 = po.map_bins(s, k, sel[:1], sg[:1], 2, 1, 0, nthreads=1)
        assert (got == exp).all() and nmap == n2 and nmap >= 1
        gpu_ctx.labels_set(sel[:0], sg[:0], 2)
        got, nmap = gpu_ctx.map_bins(0, 1, 0)
        assert nmap == 0 and not got.any()


def test_sparse_key_range_view(gpu_ctx, oracle_ctx):
    """k > 15 multi-GPU building blocks through the C-ABI: cut every chromosome's sorted list at
    common splitters, export the pieces to caller-owned device buffers, filter each key range through
    sp_sparse_view -- the concatenated ranges must equal the one-shot filter (and the oracle's)."""
    from subphaser_amd.config import sets_to_csr
    rng = np.random.RandomState(19)
    k, lower = 19, 2
    rep = [_rand_seq(rng, 400, 0, 0) for _ in range(4)]
    seqs = []
    for c in range(4):
        s = _rand_seq(rng, 20000 + 500 * c)
        for _ in range(30):
            r = rep[rng.randint(0, 2) + (2 if c % 2 else 0)]
            p = rng.randint(0, s.size - 500)
            s[p:p + r.size] = r
        seqs.append(s)
    sgs = [[[0], [1]], [[2], [3]]]
    csr = sets_to_csr(sgs, list(range(4)))
    args = (2.0, 1, 5, 1e9, 1.0)
    for ctx in (gpu_ctx, oracle_ctx):
        ctx.genome_reset(4)
        for i, s in enumerate(seqs):
            ctx.genome_add(i, s)
        ctx.count(k, lower, 0)
    nu, nr, nh = gpu_ctx.filter(*csr, *args)
    keys, counts, freqs, tot = gpu_ctx.filter_fetch(nr)
    onu, onr, onh = oracle_ctx.filter(*csr, *args)
    okeys, ocounts, ofreqs, otot = oracle_ctx.filter_fetch(onr)
    assert (nu, nr, nh) == (onu, onr, onh) and nr > 0
    assert (keys == okeys).all() and (counts == ocounts).all() and (freqs == ofreqs).all()
    lengths = gpu_ctx.lengths()
    sizes = gpu_ctx.sparse_sizes()
    assert sizes.tolist() == [len(oracle_ctx.dump(i)[0]) for i in range(4)]
    smp = gpu_ctx.sparse_sample(0, 64)
    full = oracle_ctx.dump(0)[0]
    assert (smp == full[::len(full) // 64][:64]).all()
    assert (gpu_ctx.sparse_sample(0, 10 ** 9) == full).all()          # short list: all of it
    splitters = smp[[21, 42]]
    bounds = [gpu_ctx.sparse_split(i, splitters) for i in range(4)]
    for i in range(4):
        kk = oracle_ctx.dump(i)[0]
        assert bounds[i].tolist() == [0] + np.searchsorted(kk, splitters, side="left").tolist() + [len(kk)]
    parts, tot_nu, tot_nh = [], 0, 0
    for r in range(3):
        bufs, pk, pc, n = [], [], [], []
        for i in range(4):
            lo, hi = int(bounds[i][r]), int(bounds[i][r + 1])
            dk, dc = gpu_ctx.dev_alloc(max(hi - lo, 1) * 8), gpu_ctx.dev_alloc(max(hi - lo, 1) * 4)
            gpu_ctx.sparse_export(i, lo, hi - lo, dk, dc)
            bufs += [dk, dc]
            pk.append(dk), pc.append(dc), n.append(hi - lo)
        gpu_ctx.sync()
        kk = gpu_ctx.dev_to_host(pk[1], n[1] * 8).view(np.uint64)
        assert (kk == oracle_ctx.dump(1)[0][int(bounds[1][r]):int(bounds[1][r + 1])]).all()
        gpu_ctx.sparse_view(pk, pc, n, lengths, k, lower)
        try:
            a, b, c_ = gpu_ctx.filter(*csr, *args)
            parts.append(gpu_ctx.filter_fetch(b, sort=False))
            dkeys = gpu_ctx.dev_alloc(max(b, 1) * 8)
            gpu_ctx.filter_fetch_device(dkeys, None, None, b)
            assert (gpu_ctx.dev_to_host(dkeys, b * 8).view(np.uint64) == parts[-1][0]).all()
            gpu_ctx.dev_free(dkeys)
        finally:
            gpu_ctx.sparse_view(None, None, None, None, 0, 0)
        tot_nu += a
        tot_nh += c_
        for d in bufs:
            gpu_ctx.dev_free(d)
    assert (tot_nu, tot_nh) == (nu, nh)
    assert (np.concatenate([p[0] for p in parts]) == keys).all()
    assert (np.concatenate([p[1] for p in parts]) == counts).all()
    assert (np.concatenate([p[2] for p in parts]) == freqs).all()
    nu2, nr2, nh2 = gpu_ctx.filter(*csr, *args)        # back on the local lists
    assert (nu2, nr2, nh2) == (nu, nr, nh)


@pytest.mark.parametrize("engine", [1, 2])
def test_byte_table_and_overflow_list(gpu_ctx, engine):
    """The count table is one byte per slot (raw count saturated at 255) + an overflow list of
    (slot, count) pairs in ascending slot order; this is also what travels between GPUs.  Both engines,
    a caller-bound table (sp_tables_bind), hot keys far above 255 and many buckets with overflow."""
    rng = np.random.RandomState(9)
    k, lower = 11, 3
    s = _rand_seq(rng, 5_000_000, 0.001, 0.1)
    s[1000:61000] = ord("A")                                   # poly-A: one count far above 255
    s[100000:160000] = np.frombuffer(b"ACGTTGCA" * 7500, np.uint8)   # 8 k-mers ~7500 times each
    fam = _rand_seq(rng, 3000, 0, 0)
    for p in rng.randint(200000, 4_900_000, size=400):         # a repeat family: ~3000 slots with counts ~400
        s[p:p + 3000] = fam
    from subphaser_amd import kmer as km
    n = km.dense_slots(k)
    d8 = gpu_ctx.dev_alloc(n)
    try:
        gpu_ctx.genome_reset(1)
        gpu_ctx.tables_bind(0, d8)
        gpu_ctx.genome_add(0, s)
        gpu_ctx.count(k, lower, engine)
        okeys, ocnts = po.count(s, k, 1, nthreads=4)           # raw counts
        expect = np.zeros(n, np.uint32)
        expect[km.slots_of_keys(okeys, k).astype(np.int64)] = ocnts
        n_big = int((expect >= 255).sum())
        assert n_big >= 1000
        assert (gpu_ctx.dev_to_host(d8, n) == np.minimum(expect, 255)).all()
        assert gpu_ctx.table_overflow(0) == n_big
        dov = gpu_ctx.dev_alloc(8 * n_big)
        gpu_ctx.table_overflow(0, dov, n_big)
        gpu_ctx.sync()
        pairs = gpu_ctx.dev_to_host(dov, 8 * n_big).view(np.uint32).reshape(-1, 2)
        gpu_ctx.dev_free(dov)
        assert (pairs[:, 0] == np.flatnonzero(expect >= 255)).all()      # ascending slots, no sort needed
        assert (pairs[:, 1] == expect[expect >= 255]).all()
        keys, cnts = gpu_ctx.dump(0)                                     # threshold applied on read
        assert (keys == okeys[ocnts >= lower]).all() and (cnts == ocnts[ocnts >= lower]).all()
        assert int(gpu_ctx.lengths()[0]) == int(ocnts[ocnts >= lower].astype(np.int64).sum())
    finally:
        gpu_ctx.tables_bind(0, None)
        gpu_ctx.dev_free(d8)


def test_filter_errors(gpu_ctx, count_engine):
    rng = np.random.RandomState(2)
    gpu_ctx.genome_reset(2)
    gpu_ctx.genome_add(0, _rand_seq(rng, 5000))
    gpu_ctx.genome_add(1, b"ACGT")           # no k-mers at all
    gpu_ctx.count(11, 1, 1)
    from subphaser_amd.config import sets_to_csr
    csr = sets_to_csr([[[0], [1]]], [0, 1])
    with pytest.raises(ValueError, match="have only 0 kmers"):
        gpu_ctx.filter(*csr, 2, 1, 1, 1e9, 1)
    with pytest.raises(ValueError, match="should be lower than"):
        gpu_ctx.filter(*csr, 2, 1, 100, 50, 1)
    csr1 = sets_to_csr([[[0]], [[1]]], [0, 1])
    with pytest.raises(ValueError, match="All singletons"):
        gpu_ctx.filter(*csr1, 2, 1, 1, 1e9, 1)


@pytest.mark.parametrize("engine", [1, 2, 3])
@pytest.mark.parametrize("shape", ["wheat", "peanut", "ara"])
def test_baseline_shapes(gpu_ctx, golden, shape, engine):
    """21 / 7 x 3, 20 / 10 x 2 and 13 comma-grouped chromosomes against reference-generated fixtures (G10)."""
    pc.check_shape(gpu_ctx, golden, shape, engine=engine)


@pytest.mark.parametrize("name", ["wheat_k17", "wheat_k21", "peanut_k17"])
def test_baseline_shapes_k17_k21(gpu_ctx, golden, golden_k17, name):
    """G14: the same whole-path fixtures, generated by the imported reference at k = 17 and 21 (BASELINE config 5):
    lists + join filter + hashed pair table against the reference's matrix rows, significant k-mers, `.bin.count`
    text, windows and enrichment calls."""
    ent = golden_k17[name]
    pc.check_shape(gpu_ctx, golden, ent["shape"], k=ent["k"], ent=ent)


def test_dump_roundtrip(gpu_ctx, golden, toy, tmp_path, count_engine):
    pc.check_dump_roundtrip(gpu_ctx, golden, toy, tmp_path)


def test_kmer_mat_text(gpu_ctx, golden, toy, count_engine):
    pc.check_kmer_mat_text(gpu_ctx, golden, toy)


def test_map_cases(gpu_ctx, golden, toy, count_engine):
    pc.check_map_cases(gpu_ctx, golden, toy)


def test_map_features(gpu_ctx, golden, toy, tmp_path):
    pc.check_map_features(gpu_ctx, golden, toy, tmp_path)


def test_long_feature(gpu_ctx, golden, toy, tmp_path):
    pc.check_long_feature(gpu_ctx, golden, toy, tmp_path)


def test_map_dict_labels(gpu_ctx, golden, toy):
    pc.check_dict_labels(gpu_ctx, golden, toy)


def test_hotpath_stack(gpu_ctx, golden, toy, count_engine):
    pc.check_hotpath_stack(gpu_ctx, golden, toy)


def test_pipeline_cli(gpu_ctx, golden, toy, tmp_path, count_engine):
    pc.check_pipeline_cli(gpu_ctx, golden, toy, tmp_path)


def test_pipeline_cli_bed_features(gpu_ctx, golden, toy, tmp_path):
    pc.check_pipeline_cli_bed(gpu_ctx, golden, toy, tmp_path)


def test_pipeline_cli_background_writers(gpu_ctx, golden, toy, tmp_path, monkeypatch):
    """the same CLI run with every text output forced onto the background writer threads (at toy size they are
    written inline): identical files, checkpoints recorded after the writers"""
    monkeypatch.setenv("SP_BG_MIN_ROWS", "1")
    pc.check_pipeline_cli(gpu_ctx, golden, toy, tmp_path)


@pytest.mark.parametrize("k,S,flt", [(13, 3, "core"), (13, 3, "x"), (15, 2, "core"), (15, 5, "x"), (4, 2, "core"), (17, 3, "core"), (17, 3, "x"),
                                     (13, 9, "core"), (21, 9, "core")])
def test_map_vs_oracle_random(gpu_ctx, monkeypatch, k, S, flt):
    """Bins, n_mapped, labels_hit against the oracle through every map engine (compact / direct pair table, label table for
    S > 7, k > 15 pair-keyed and per-k-mer tables) with both addressings of the pair filter: "core" = the word of a (k-1)-mer is
    addressed by its smaller-hashed (k-3)-mer core (round 6: a lane re-uses the previous pair's word for a third of its pairs),
    "x" = by the (k-1)-mer itself (SP_MAP_FILTER=x, the cross-check); k = 4: cores too short, the old addressing either way."""
    if flt == "x":
        monkeypatch.setenv("SP_MAP_FILTER", "x")
    rng = np.random.RandomState(33)
    s = _rand_seq(rng, 250000)
    rep = _rand_seq(rng, 500, 0, 0)
    for _ in range(100):
        p = rng.randint(0, s.size - 600)
        s[p:p + 500] = rep
    gpu_ctx.genome_reset(1)
    gpu_ctx.genome_add(0, s)
    gpu_ctx.count(k, 1, 1)
    keys, cnts = gpu_ctx.dump(0)
    sel = keys[cnts >= 20]
    sg = (np.arange(sel.size) % S).astype(np.uint8)
    gpu_ctx.labels_set(sel, sg, S)
    for bin_size, chunk in ((10000, 10_000_000), (100, 2000), (7, 0), (333, 1000), (1, 0), (50000, 100000)):
        got, n = gpu_ctx.map_bins(0, bin_size, chunk)
        exp, hit, n2 = po.map_bins(s, k, sel, sg, S, bin_size, chunk, nthreads=4)
        assert got.shape == exp.shape, (bin_size, chunk)
        assert (got == exp).all() and n == n2, (bin_size, chunk)
    allb, nm = gpu_ctx.map_bins_all(50000, 100000)       # batched entry point, same numbers
    assert (allb[0] == exp).all() and int(nm[0]) == n2
    assert gpu_ctx.labels_hit() == int(hit.sum())
    assert int(got.sum()) == int(cnts[cnts >= 20].astype(np.int64).sum())   # every occurrence mapped once


@pytest.mark.parametrize("k", [13, 15, 21])
def test_labels_set_device(gpu_ctx, k):
    """sp_labels_set_device (the label set handed over in device memory, KmerLabels.on_device) maps exactly like
    sp_labels_set; a label >= n_sg is caught on the device."""
    from subphaser_amd.seqs import KmerLabels
    rng = np.random.RandomState(77 + k)
    s = np.concatenate([_rand_seq(rng, 60_000), np.tile(_rand_seq(rng, 300, 0, 0), 30)])
    gpu_ctx.genome_reset(1)
    gpu_ctx.genome_add(0, s)
    gpu_ctx.count(k, 1, 0)
    keys, cnts = gpu_ctx.dump(0)
    sel = keys[(cnts >= 2) | (rng.rand(keys.size) < 0.2)]
    sg = (np.arange(sel.size) % 3).astype(np.uint8)
    gpu_ctx.labels_set(sel, sg, 3)
    exp, n_exp = gpu_ctx.map_bins(0, 777, 5000)
    hit_exp = gpu_ctx.labels_hit()
    lab = KmerLabels(sel, sg, ["A", "B", "C"], k)
    gpu_ctx.labels_set(sel[:3], sg[:3], 3)            # something else in between
    gpu_ctx.labels_set_from(lab, 3)
    got, n = gpu_ctx.map_bins(0, 777, 5000)
    assert (got == exp).all() and n == n_exp and gpu_ctx.labels_hit() == hit_exp
    oref, _, n2 = po.map_bins(s, k, sel, sg, 3, 777, 5000, nthreads=2)
    assert (got == oref).all() and n == n2
    d_keys, d_sg = lab.on_device(gpu_ctx)
    with pytest.raises(ValueError, match="label 2 >= n_sg 2"):
        gpu_ctx.labels_set_device(d_keys, d_sg, sel.size, 2)
    lab.release_device()


@pytest.mark.parametrize("k,mode", [(k, "direct") for k in (2, 5, 9, 13, 14, 15)] +
                         [(k, m) for k in (13, 14, 15) for m in ("compact", "compact-crowded")])
def test_map_compact_pair_table(gpu_ctx, monkeypatch, capfd, k, mode):
    """The compact exact pair table (S <= 3: QUAD buckets of four tagged entries addressed by the (k-3)-mer two pairs share +
    overflow table, sp_map.h) against the oracle and against the direct table, with a load that sends many keys to the
    overflow table, with palindromic (k-1)-mers, N runs, both strands; bins, n_mapped, labels_hit go through the same look-up.
    The compact layout needs k >= 5 and 2^bb buckets with 2 (k - 3) - 2 <= bb <= 2 (k - 3): with this test's ~10^5 labels
    that is k = 13 .. 15 (smaller k silently take the direct table -- advisor r05), and the test ASSERTS which table the
    library chose from its SP_DEBUG_FILTER line."""
    monkeypatch.setenv("SP_CTAB", "0" if mode == "direct" else "1")
    monkeypatch.setenv("SP_DEBUG_FILTER", "1")
    if mode == "compact-crowded":
        monkeypatch.setenv("SP_CTAB_FACTOR", "1")      # ~4 keys per bucket of four: many keys overflow
    capfd.readouterr()
    rng = np.random.RandomState(4100 + k)
    pal = np.frombuffer(b"ACGTACGTACGTACGTACGTAATTAATTAATTAATTGGCCGGCCGGCC", dtype=np.uint8)
    s = np.concatenate([_rand_seq(rng, 90_000, 0.01, 0.1), np.tile(pal, 40), _rand_seq(rng, 30_000)])
    s2 = np.concatenate([s[20_000:50_000][::-1].copy(), np.tile(pal, 10), _rand_seq(rng, 20_000)])
    gpu_ctx.genome_reset(2)
    gpu_ctx.genome_add(0, s)
    gpu_ctx.genome_add(1, s2)
    gpu_ctx.count(k, 1, 1)
    keys, cnts = gpu_ctx.dump(0)
    sel = keys[(cnts >= 2) | (rng.rand(keys.size) < 0.4)]
    assert sel.size
    for S in (1, 2, 3):
        sg = (np.arange(sel.size) % S).astype(np.uint8)
        gpu_ctx.labels_set(sel, sg, S)
        assert ("compact pair table" in capfd.readouterr().err) == (mode != "direct"), (k, mode, S)
        hit_all = np.zeros(sel.size, bool)
        for bin_size, chunk in ((1000, 10_000), (7, 0)):
            for ci, seq in enumerate((s, s2)):
                got, n = gpu_ctx.map_bins(ci, bin_size, chunk)
                exp, hit, n2 = po.map_bins(seq, k, sel, sg, S, bin_size, chunk, nthreads=4)
                assert got.shape == exp.shape and (got == exp).all() and n == n2, (S, bin_size, chunk, ci)
                hit_all |= hit.astype(bool)
        assert gpu_ctx.labels_hit() == int(hit_all.sum())
    # a second, smaller label set over the same context: nothing of the first one may survive
    sel2 = sel[::5]
    sg2 = (np.arange(sel2.size) % 3).astype(np.uint8)
    gpu_ctx.labels_set(sel2, sg2, 3)
    got, n = gpu_ctx.map_bins(0, 500, 0)
    exp, hit, n2 = po.map_bins(s, k, sel2, sg2, 3, 500, 0, nthreads=4)
    assert (got == exp).all() and n == n2 and gpu_ctx.labels_hit() == int(hit.sum())


@pytest.mark.parametrize("k,S", [(13, 8), (15, 9), (15, 8), (13, 9)])
def test_map_label_table_engine_many_subgenomes(gpu_ctx, k, S):
    """More than 7 subgenomes: the pair table's 3-bit label does not fit, the dense per-k-mer label table
    (`k5_map_lab`) maps instead.  Bins, n_mapped, labels_hit and the batched entry point against the oracle."""
    rng = np.random.RandomState(900 + 10 * k + S)
    unit = _rand_seq(rng, 40, 0, 0)
    s = np.concatenate([_rand_seq(rng, 120_000), np.tile(unit, 300), _rand_seq(rng, 60_000, p_other=0.05)])
    s2 = np.concatenate([s[50_000:90_000], _rand_seq(rng, 30_000)])
    gpu_ctx.genome_reset(2)
    gpu_ctx.genome_add(0, s)
    gpu_ctx.genome_add(1, s2)
    gpu_ctx.count(k, 1, 0)
    keys, cnts = gpu_ctx.dump(0)
    sel = keys[(cnts >= 2) | (np.arange(keys.size) % 7 == 0)]       # the repeats + every seventh k-mer
    assert sel.size > 500
    sg = (np.arange(sel.size) % S).astype(np.uint8)
    gpu_ctx.labels_set(sel, sg, S)
    hit_all = np.zeros(sel.size, bool)
    for bin_size, chunk in ((10000, 10_000_000), (100, 2000), (333, 1000)):
        for ci, seq in enumerate((s, s2)):
            got, n = gpu_ctx.map_bins(ci, bin_size, chunk)
            exp, hit, n2 = po.map_bins(seq, k, sel, sg, S, bin_size, chunk, nthreads=4)
            assert got.shape == exp.shape and (got == exp).all() and n == n2, (bin_size, chunk, ci)
            hit_all |= hit.astype(bool)
    assert (got.sum(axis=0) > 0).sum() == S          # every subgenome column is exercised
    allb, nm = gpu_ctx.map_bins_all(333, 1000)
    assert (allb[1] == exp).all() and int(nm[1]) == n2
    assert gpu_ctx.labels_hit() == int(hit_all.sum())


def test_enrich_bin(gpu_ctx, golden, tmp_path):
    pc.check_enrich_bin(gpu_ctx, golden, tmp_path)


def test_enrich_features(gpu_ctx, golden, tmp_path):
    pc.check_enrich_features(gpu_ctx, golden, tmp_path)


def test_fisher_cells_and_tails(gpu_ctx, golden):
    pc.check_fisher_cells_and_tails(gpu_ctx, golden)


def test_enrich_vs_oracle_random(gpu_ctx):
    rng = np.random.RandomState(44)
    for S, W, scale in ((2, 300, 30), (3, 500, 200), (5, 200, 5), (3, 64, 3e6), (9, 50, 50)):
        t = rng.poisson(scale, size=(W, S)).astype(np.int64)
        t[: W // 5, 0] += rng.poisson(scale * 2 + 5, W // 5)
        t[W // 5: W // 4] = 0
        with np.errstate(all="ignore"):
            gp, ga, gs, gr = gpu_ctx.enrich(t, 0.05, 0.5)
            op, oa, os_, orr = po.enrich(t, 0.05, 0.5)
        assert np.allclose(gp, op, rtol=1e-7, atol=1e-300), (S, np.abs(gp - op).max())
        assert np.abs(gp - op).max() <= 1e-6
        same = np.isclose(gp, op, rtol=1e-12, atol=0).all(axis=1)   # decisions can only differ on p ties
        assert (ga[same] == oa[same]).all()
        assert (gs[same] == os_[same]).all()
        assert ((gr == orr) | (np.isnan(gr) & np.isnan(orr))).all()


def test_enrich_needs_two_columns(gpu_ctx):
    with pytest.raises(ValueError):
        gpu_ctx.enrich(np.ones((3, 1), np.int64))


def test_full_size_properties(gpu_ctx):
    """Size-independent invariants on a chromosome too large for the pure-Python checks:
    the sum of all counts equals the number of valid windows; counting twice is idempotent;
    mapping with every dumped k-mer labelled maps exactly sum(counts) positions."""
    n = 20_000_000
    d = gpu_ctx.dev_alloc(n)
    try:
        gpu_ctx.synth_chrom(d, n, seed=5, set_id=0, sg_id=0, n_sg=2, chrom_id=0)
        gpu_ctx.genome_reset(1)
        gpu_ctx.genome_add_device(0, d, n)
        host = gpu_ctx.dev_to_host(d, n)
    finally:
        gpu_ctx.sync()
        gpu_ctx.dev_free(d)
    k = 15
    gpu_ctx.count(k, 1, 1)
    total = int(gpu_ctx.lengths()[0])
    up = host & 0xDF
    ok = ((up == 65) | (up == 67) | (up == 71) | (up == 84)).astype(np.int64)
    run = np.zeros(n, np.int64)
    c = np.cumsum(ok)
    reset = np.where(ok == 0, c, 0)
    np.maximum.accumulate(reset, out=reset)
    run = c - reset
    assert total == int((run >= k).sum())
    keys, cnts = gpu_ctx.dump(0)
    ok_keys, ok_cnts = po.count(host, k, 1, nthreads=8)
    assert (keys == ok_keys).all() and (cnts == ok_cnts).all()
    gpu_ctx.count(k, 3, 2)            # engine 2 on the same chromosome: identical tables
    assert int(gpu_ctx.lengths()[0]) == int(cnts[cnts >= 3].astype(np.int64).sum())
    keys3, cnts3 = gpu_ctx.dump(0)
    assert (keys3 == keys[cnts >= 3]).all() and (cnts3 == cnts[cnts >= 3]).all()
    gpu_ctx.labels_set(keys3, np.zeros(keys3.size, np.uint8), 1)
    got, nmap = gpu_ctx.map_bins(0, 10000, 10_000_000)
    assert nmap == int(cnts3.astype(np.int64).sum()) == int(got.sum())
    assert gpu_ctx.labels_hit() == keys3.size


def test_peanut_sized_chromosomes_list_engine(gpu_ctx):
    """Two chromosomes of 2^26 + bases (the peanut-like size class: ~8 K keys per fine bucket) counted as LISTS (engine 3: the
    four-quads-per-thread geometry of c2_count_list, round 6, chains side by side) and as byte tables (engine 2): same lengths,
    identical dumps at two thresholds."""
    n = (1 << 26) + 5_000_000
    ds = [gpu_ctx.dev_alloc(n), gpu_ctx.dev_alloc(n)]
    try:
        for i, d in enumerate(ds):
            gpu_ctx.synth_chrom(d, n, seed=2, set_id=i, sg_id=i, n_sg=2, chrom_id=i)
        gpu_ctx.genome_reset(2)
        for i, d in enumerate(ds):
            gpu_ctx.genome_add_device(i, d, n)
        for lower in (2, 3):
            res = {}
            for eng in (3, 2):
                gpu_ctx.count(15, lower, eng)
                res[eng] = (gpu_ctx.lengths().tolist(), [gpu_ctx.dump(i) for i in range(2)])
            assert res[3][0] == res[2][0] and min(res[3][0]) > 100_000, (lower, res[3][0], res[2][0])
            for i in range(2):
                assert (res[3][1][i][0] == res[2][1][i][0]).all() and (res[3][1][i][1] == res[2][1][i][1]).all(), (lower, i)
    finally:
        for d in ds:
            gpu_ctx.dev_free(d)


def test_wheat_sized_chromosome_properties(gpu_ctx):
    """A chromosome of the size BASELINE.json's headline config uses (667 Mb) is far beyond what the
    CPU oracle finishes in seconds, so the check is by size-independent properties:
    engine 1 and engine 2 build the same table (same totals at two thresholds, identical dump of the
    high-count k-mers); sum of counts == number of valid windows (run-length identity on a 64-Mb
    prefix copied back); every occurrence of a labelled k-mer is mapped exactly once; windows sum to bins."""
    n = 667_000_000
    d = gpu_ctx.dev_alloc(n)
    try:
        gpu_ctx.synth_chrom(d, n, seed=3, set_id=1, sg_id=2, n_sg=3, chrom_id=5)
        gpu_ctx.genome_reset(1)
        gpu_ctx.genome_add_device(0, d, n)
        res = {}
        for eng in (1, 2):
            gpu_ctx.count(15, 1, eng)
            tot1 = int(gpu_ctx.lengths()[0])
            gpu_ctx.count(15, 200, eng)
            tot200 = int(gpu_ctx.lengths()[0])
            keys, cnts = gpu_ctx.dump(0)
            res[eng] = (tot1, tot200, keys, cnts)
        assert res[1][0] == res[2][0] and res[1][1] == res[2][1]
        assert (res[1][2] == res[2][2]).all() and (res[1][3] == res[2][3]).all()
        tot1, tot200, keys, cnts = res[2]
        assert len(keys) > 1000 and int(cnts.astype(np.int64).sum()) == tot200
        # valid-window identity: N runs of 1000 every 50 Mb, nothing else invalid in the generator
        m = 64_000_000
        host = gpu_ctx.dev_to_host(d, m)
        up = host & 0xDF
        ok = (up == 65) | (up == 67) | (up == 71) | (up == 84)
        n_runs = (n - 1) // 50_000_000                  # runs at 50M, 100M, ... < n
        assert int((~ok).sum()) == 1000                 # the prefix holds exactly one N run
        expected = (n - 14) - n_runs * (1000 + 14)      # each run kills 1000 + (k-1) windows
        assert tot1 == expected, (tot1, expected)
        # map: label every dumped k-mer, alternate subgenomes
        sg = (np.arange(keys.size) % 3).astype(np.uint8)
        gpu_ctx.labels_set(keys, sg, 3)
        slots, nm = gpu_ctx.map_bins_all(10000, 10_000_000)
        assert int(nm[0]) == tot200 == int(slots[0].astype(np.int64).sum())
        per_sg = [int(cnts[sg == j].astype(np.int64).sum()) for j in range(3)]
        assert slots[0].astype(np.int64).sum(axis=0).tolist() == per_sg
        win, woff = gpu_ctx.stack_windows(10000, 10_000_000, 1_000_000, [n])
        assert win.sum(axis=0).tolist() == per_sg
        assert gpu_ctx.labels_hit() == keys.size
    finally:
        gpu_ctx.sync()
        gpu_ctx.dev_free(d)


@pytest.mark.parametrize("k", [17, 21, 27])
def test_wheat_sized_chromosome_properties_sparse(gpu_ctx, k):
    """k > 15 at the headline chromosome size (667 Mb), by size-independent properties: the MSD-partition
    engine and the device-wide radix-sort engine produce the same sorted list (same totals at two
    thresholds, identical dump of the high-count k-mers); sum of counts == number of valid windows;
    keys strictly ascending and canonical; every occurrence of a labelled k-mer is mapped exactly once."""
    from subphaser_amd import kmer as km
    n = 667_000_000
    d = gpu_ctx.dev_alloc(n)
    try:
        gpu_ctx.synth_chrom(d, n, seed=3, set_id=1, sg_id=2, n_sg=3, chrom_id=5)
        gpu_ctx.genome_reset(1)
        gpu_ctx.genome_add_device(0, d, n)
        res = {}
        for eng in (0, 1):
            gpu_ctx.count(k, 1, eng)
            tot1 = int(gpu_ctx.lengths()[0])
            gpu_ctx.count(k, 200, eng)
            tot200 = int(gpu_ctx.lengths()[0])
            keys, cnts = gpu_ctx.dump(0, sort=False)
            res[eng] = (tot1, tot200, keys, cnts)
        assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
        assert (res[0][2] == res[1][2]).all() and (res[0][3] == res[1][3]).all()
        tot1, tot200, keys, cnts = res[0]
        assert len(keys) > 1000 and int(cnts.astype(np.int64).sum()) == tot200 and int(cnts.min()) >= 200
        assert (keys[1:] > keys[:-1]).all()                       # the engine itself emits ascending keys
        assert (km.canonical(keys, k) == keys).all()
        n_runs = (n - 1) // 50_000_000
        assert tot1 == (n - (k - 1)) - n_runs * (1000 + (k - 1))  # each N run kills 1000 + (k-1) windows
        sg = (np.arange(keys.size) % 3).astype(np.uint8)
        gpu_ctx.labels_set(keys, sg, 3)
        slots, nm = gpu_ctx.map_bins_all(10000, 10_000_000)
        assert int(nm[0]) == tot200 == int(slots[0].astype(np.int64).sum())
        per_sg = [int(cnts[sg == j].astype(np.int64).sum()) for j in range(3)]
        assert slots[0].astype(np.int64).sum(axis=0).tolist() == per_sg
        assert gpu_ctx.labels_hit() == keys.size
    finally:
        gpu_ctx.sync()
        gpu_ctx.dev_free(d)


@pytest.mark.parametrize("mode", ["one-stream", "streams"])
def test_fuzz_against_oracle(gpu_ctx, oracle_ctx, count_engine, mode):
    """150 random cases of tools/fuzz_parity.py (random genomes, k in 1..32, thresholds, engines, labels,
    bin / chunk sizes, set structures): counts, matrix rows, bin counts, feature totals, bit-exact.  "streams" (round 6):
    3..7 chains forced in flight at every count, chromosomes of several tiles, a second context busy on the same GPU."""
    mod = _tool("fuzz_parity")
    assert mod.run(150, 12345, gpu_ctx, oracle_ctx, verbose=False, streams=(mode == "streams")) == 0


@pytest.mark.parametrize("k", [15, 17, 22])
def test_medium_scale_count_filter_map_vs_oracle(gpu_ctx, oracle_ctx, k):
    """Three 6-Mb chromosomes with planted repeat families, default engines (k = 15: engine 2 counts, pair
    filter sized from the label set, 768-thread map blocks, several block iterations per kernel; k = 17 / 22:
    MSD-partition engine with u32 / u64 residuals, sort-join filter, hash-table map): dumps, matrix rows and
    bin counts bit-exact against the oracle."""
    from subphaser_amd.config import sets_to_csr
    rng = np.random.RandomState(2024)
    lower = 3
    fams = [[_rand_seq(rng, 600, 0, 0) for _ in range(8)] for _ in range(2)]
    seqs = []
    for c in range(3):
        s = _rand_seq(rng, 6_000_000 + 4099 * c, 0.0005, 0.1)
        lib = fams[c % 2]
        for _ in range(1500):
            r = lib[rng.randint(0, len(lib))].copy()
            mut = rng.rand(r.size) < 0.02
            r[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, int(mut.sum()))]
            p = rng.randint(0, s.size - 700)
            s[p:p + r.size] = r
        seqs.append(s)
    for ctx in (gpu_ctx, oracle_ctx):
        ctx.genome_reset(3)
        for i, s in enumerate(seqs):
            ctx.genome_add(i, s)
        ctx.count(k, lower, 0)
    assert gpu_ctx.lengths().tolist() == oracle_ctx.lengths().tolist()
    for i in range(3):
        gk, gc = gpu_ctx.dump(i)
        ok, oc = oracle_ctx.dump(i)
        assert gk.shape == ok.shape and (gk == ok).all() and (gc == oc).all(), i
    csr = sets_to_csr([[[0], [1], [2]]], [0, 1, 2])
    res = []
    for ctx in (gpu_ctx, oracle_ctx):
        nu, nr, nh = ctx.filter(*csr, 2.0, 1, 50, 1e9, 1.0)
        res.append((nu, nr, nh) + tuple(ctx.filter_fetch(nr)))
    assert res[0][:3] == res[1][:3] and res[0][1] > 100
    for a, b in zip(res[0][3:], res[1][3:]):
        assert (a == b).all()
    keys = res[0][3]
    sg = (np.arange(keys.size) % 2).astype(np.uint8)
    for ctx in (gpu_ctx, oracle_ctx):
        ctx.labels_set(keys, sg, 2)
    for i in range(3):
        for bs, ch in ((10000, 10_000_000), (10000, 1_000_000), (333, 50_000)):
            g, gn = gpu_ctx.map_bins(i, bs, ch)
            o, on = oracle_ctx.map_bins(i, bs, ch)
            assert g.shape == o.shape and (g == o).all() and gn == on, (i, bs, ch)
    allb, nm = gpu_ctx.map_bins_all(10000, 10_000_000)
    for i in range(3):
        o, on = oracle_ctx.map_bins(i, 10000, 10_000_000)
        assert (allb[i] == o).all() and int(nm[i]) == on
    assert gpu_ctx.labels_hit() == oracle_ctx.labels_hit()


@pytest.mark.parametrize("config,scale", [("wheat", 0.003), ("peanut", 0.012), ("ara", 0.08)])
def test_synth_baseline_shapes_vs_oracle(gpu_ctx, config, scale, count_engine):
    """bench.py's own generator at the BASELINE chromosome / set structures (21 / 7 x 3, 20 / 10 x 2, 13
    comma-grouped), scaled down until the oracle finishes in seconds: HotPath on the HIP context against the
    oracle step by step -- lengths, (n_union, n_rows, n_hist), matrix rows, bins, windows, p-values, calls."""
    from oracle_ctx import OracleContext
    from subphaser_amd import cluster
    from subphaser_amd.hotpath import HotPath
    from subphaser_amd.synth import SynthGenome
    gen = SynthGenome(config, scale)
    C, S = len(gen.chroms), gen.S
    ptrs, host = [], []
    try:
        for c in gen.chroms:
            p = gpu_ctx.dev_alloc(c["length"])
            gpu_ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], S, c["chrom_id"], c["exchange"])
            ptrs.append(p)
        gpu_ctx.sync()
        host = [gpu_ctx.dev_to_host(p, c["length"]) for p, c in zip(ptrs, gen.chroms)]
        lens = [c["length"] for c in gen.chroms]
        kw = dict(k=15, lower_count=3, min_freq=20, bin_size=10000, chunk_size=10_000_000, window_size=250_000)
        hp = HotPath(gpu_ctx, gen.labels, lens, gen.sgs, **kw)
        a = hp.count_and_filter(ptrs, sort=True)
    finally:
        gpu_ctx.sync()
        for p in ptrs:
            gpu_ctx.dev_free(p)
    octx = OracleContext(nthreads=min(32, len(os.sched_getaffinity(0))))
    octx.genome_reset(C)
    for i, s in enumerate(host):
        octx.genome_add(i, s)
    octx.count(15, 3)
    assert a.kmer_lengths.tolist() == octx.lengths().tolist()
    ho = HotPath(octx, gen.labels, lens, gen.sgs, **kw)
    nu, nr, nh = octx.filter(*ho.csr, ho.min_fold, ho.baseline, ho.min_freq, ho.max_freq, ho.ratio)
    assert (a.n_union, a.n_rows, a.n_hist) == (nu, nr, nh) and nr > 50, (a.n_union, a.n_rows, a.n_hist, nu, nr, nh)
    okeys, ocounts, ofreqs, otot = octx.filter_fetch(nr)
    assert (a.keys == okeys).all() and (a.counts == ocounts).all() and (a.tot == otot).all()

    class _Mat:
        pass
    mat = _Mat()
    mat.labels, mat.keys, mat.k = gen.labels, a.keys, 15
    mat.freqs = a.counts.astype(np.float64) / np.asarray(a.kmer_lengths, np.float64)
    assert (mat.freqs == ofreqs).all()
    cl = cluster.Cluster(mat, n_clusters=S, sg_assigned=gen.sg_assigned)
    labels = cl.output_kmers(open(os.devnull, "w"), max_pval=0.05)
    assert len(labels.keys) > 20
    b = hp.map_and_enrich(labels, S)
    o = ho.map_and_enrich(labels, S)
    assert b.n_mapped == o.n_mapped and b.n_mapped > 0
    for x, y in zip(b.bins, o.bins):
        assert x.shape == y.shape and (x == y).all()
    assert b.coords == o.coords and (b.window_counts == o.window_counts).all() and len(b.window_counts) > C
    assert (b.argmin == o.argmin).all() and (b.sig == o.sig).all() and b.sig.any()
    assert np.allclose(b.pvals, o.pvals, rtol=1e-6, atol=1e-6)      # north star: p-values within 1e-6
    assert (b.ratios == o.ratios).all() or np.allclose(b.ratios, o.ratios, rtol=1e-15, atol=0, equal_nan=True)


def test_shared_host_segment_copy(gpu_ctx):
    """The multi-GPU matrix hand-over: a POSIX shared-memory segment is page-locked (sp_host_register) and the
    device rows are copied straight into it at a row offset (what every rank does with its own rows)."""
    import ctypes
    from multiprocessing import shared_memory
    rng = np.random.RandomState(3)
    rows = rng.randint(0, 1 << 31, size=(5000, 21)).astype(np.uint32)
    d = gpu_ctx.dev_alloc(rows.nbytes)
    shm = shared_memory.SharedMemory(create=True, size=4 * rows.nbytes)
    try:
        gpu_ctx.host_to_dev(d, rows)
        addr = ctypes.addressof(ctypes.c_char.from_buffer(shm.buf))
        gpu_ctx.host_register(addr, 4 * rows.nbytes)
        gpu_ctx.dev_to_host_ptr(addr + rows.nbytes, d, rows.nbytes)
        got = np.frombuffer(shm.buf, np.uint32, rows.size, rows.nbytes).reshape(rows.shape).copy()
        gpu_ctx.host_unregister(addr)
        assert (got == rows).all()
    finally:
        gpu_ctx.dev_free(d)
        got = None
        shm.close()
        shm.unlink()


@pytest.mark.parametrize("k,n_sg", [(15, 2), (17, 2), (21, 2), (15, 9), (17, 9)])
def test_feature_join_at_reduced_scale(gpu_ctx, k, n_sg):
    """(k = 21: BASELINE config 5's second k; 9 subgenomes: the label-table / per-k-mer hash engines.)
    BASELINE config 5's feature-window join at 1/100 of its size: 20,000 features of U[0.2, 10] kb cut from a
    synthetic chromosome (N runs, soft-masked repeats, features that overlap each other), mapped back to back in one
    upload (sp_map_features: k-mers across feature boundaries rejected in the kernel) -- per-feature totals and the
    number of labelled k-mers seen against the oracle, which maps every feature on its own."""
    from oracle_ctx import OracleContext
    n, n_feat = 40_000_000, 20000
    d = gpu_ctx.dev_alloc(n)
    try:
        gpu_ctx.synth_chrom(d, n, seed=5, set_id=0, sg_id=1, n_sg=2, chrom_id=3)
        gpu_ctx.genome_reset(1)
        gpu_ctx.genome_add_device(0, d, n)
        host = gpu_ctx.dev_to_host(d, n)
    finally:
        gpu_ctx.sync()
        gpu_ctx.dev_free(d)
    gpu_ctx.count(k, 3, 0)
    keys, cnts = gpu_ctx.dump(0)
    sel = np.flatnonzero(cnts >= 40)[::3]                    # labelled: every third k-mer seen >= 40 times
    lab_keys, lab_sg = keys[sel], (np.arange(sel.size) % n_sg).astype(np.uint8)      # 9 subgenomes: the per-k-mer label engines
    assert lab_keys.size > 1000
    rng = np.random.RandomState(4)
    ln = rng.randint(200, 10001, size=n_feat)
    ln[:50] = rng.randint(1, k + 3, size=50)                # shorter than / around k
    st = rng.randint(0, n - 10001, size=n_feat)
    off = np.concatenate(([0], np.cumsum(ln))).astype(np.int64)
    cat = np.concatenate([host[s:s + l] for s, l in zip(st.tolist(), ln.tolist())])
    gpu_ctx.labels_set(lab_keys, lab_sg, n_sg)
    got = gpu_ctx.map_features_cat(cat, off)
    octx = OracleContext(nthreads=1)
    octx.k = k
    octx.labels_set(lab_keys, lab_sg, n_sg)
    exp = octx.map_features_cat(cat, off)
    assert got.shape == exp.shape and (got == exp).all()
    assert int(got.sum()) > 10 * n_feat
    assert gpu_ctx.labels_hit() == octx.labels_hit()
    # the same features as BED intervals over the resident chromosome (sp_map_intervals): no sequence is uploaded;
    # totals and the set of labelled k-mers seen must equal the FASTA path's
    gpu_ctx.labels_set(lab_keys, lab_sg, n_sg)
    iv = gpu_ctx.map_intervals(np.zeros(n_feat, np.int32), st, st + ln)
    assert iv.shape == exp.shape and (iv == exp).all()
    assert gpu_ctx.labels_hit() == octx.labels_hit()
    with pytest.raises(Exception):
        gpu_ctx.map_intervals([0], [5], [n + 1])            # beyond the chromosome
    with pytest.raises(Exception):
        gpu_ctx.map_intervals([1], [0], [10])               # no such chromosome
    assert gpu_ctx.map_intervals([], [], []).shape == (0, n_sg)


def test_kmer_ttest_device_vs_scipy(gpu_ctx):
    """f-1: the per-k-mer Student test on the device against scipy.stats.ttest_ind row by row, incl. constant
    groups (0 / 0 -> NaN, kept like the reference keeps it), a perfect separation (t = inf -> p = 0), ties between
    group means (subgenome order decides) and unequal group sizes."""
    from scipy import stats as st
    rng = np.random.RandomState(12)
    C, groups = 13, [[0, 3, 4, 9, 12], [1, 2, 5, 6, 7, 8, 10, 11]]        # the Arabidopsis suecica 5 + 8 split
    lengths = rng.randint(10**6, 10**8, size=C).astype(np.int64)
    counts = rng.poisson(30, size=(4000, C)).astype(np.uint32)
    counts[:500, groups[0]] += rng.poisson(200, size=(500, 5)).astype(np.uint32)
    counts[500:900, groups[1]] += rng.poisson(90, size=(400, 8)).astype(np.uint32)
    counts[900:950] = 0                                   # all zero: 0 / 0
    counts[950:960] = 7                                   # constant row
    counts[960:970, :] = 0
    counts[960:970, groups[0]] = 5                        # zero variance in both groups, different means: t = inf
    lengths_eq = lengths.copy()
    top, second, pvals, means = gpu_ctx.kmer_ttest(counts, lengths, groups)
    X = counts.astype(np.float64) / lengths.astype(np.float64)
    for r in list(range(0, 4000, 37)) + list(range(895, 975)):
        m = [X[r, g].mean() for g in groups]
        order = sorted(range(2), key=lambda g: (-(X[r, groups[g]].sum() / len(groups[g])), g))
        assert top[r] == order[0] and second[r] == order[1], r
        assert np.allclose(means[r], m, rtol=1e-14, atol=0)
        with np.errstate(all="ignore"):
            exp = st.ttest_ind(X[r, groups[order[0]]], X[r, groups[order[1]]]).pvalue
        if np.isnan(exp):
            assert np.isnan(pvals[r]), r
        else:
            assert np.isclose(pvals[r], exp, rtol=1e-9, atol=1e-300), (r, pvals[r], exp)
    three = [[0, 1, 2], [3, 4, 5, 6], [7, 8, 9, 10, 11, 12]]
    top3, sec3, p3, _ = gpu_ctx.kmer_ttest(counts[:300], lengths, three)
    for r in range(0, 300, 11):
        order = sorted(range(3), key=lambda g: (-(X[r, three[g]].sum() / len(three[g])), g))
        assert (top3[r], sec3[r]) == (order[0], order[1])
        exp = st.ttest_ind(X[r, three[order[0]]], X[r, three[order[1]]]).pvalue
        assert np.isclose(p3[r], exp, rtol=1e-9, atol=1e-300)
