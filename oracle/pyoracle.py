"""ctypes binding of the CPU oracle (oracle/sp_oracle.c) + definition-level brute force.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under subphaser_amd/ may import this.
"""
import ctypes as C
import os
import subprocess
from collections import Counter

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsp_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("sp_oracle.c", "sp_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, i64, i32, u32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_double
    L.spo_last_error.restype = C.c_char_p
    L.spo_count.restype = vp
    L.spo_count.argtypes = [vp, i64, C.c_int, C.c_int]
    L.spo_counts_n.restype = i64
    L.spo_counts_n.argtypes = [vp, u32]
    L.spo_counts_fetch.restype = i64
    L.spo_counts_fetch.argtypes = [vp, u32, vp, vp]
    L.spo_counts_free.argtypes = [vp]
    L.spo_filter.restype = vp
    L.spo_filter.argtypes = [C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, dbl, C.c_int, dbl, dbl, dbl, vp]
    for name in ("spo_filtered_n_union", "spo_filtered_n_rows", "spo_filtered_n_hist"):
        getattr(L, name).restype = i64
        getattr(L, name).argtypes = [vp]
    L.spo_filtered_lengths.argtypes = [vp, vp]
    L.spo_filtered_fetch.argtypes = [vp, vp, vp, vp, vp]
    L.spo_filtered_hist.argtypes = [vp, vp]
    L.spo_filtered_free.argtypes = [vp]
    L.spo_map_bins.restype = i64
    L.spo_map_bins.argtypes = [vp, i64, C.c_int, vp, vp, i64, C.c_int, i64, i64, vp, i64, vp, C.c_int]
    L.spo_hypergeom_right_tail.restype = dbl
    L.spo_hypergeom_right_tail.argtypes = [i64, i64, i64, i64]
    L.spo_fisher_cells.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.spo_enrich.argtypes = [vp, i64, C.c_int, dbl, dbl, vp, vp, vp, vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _ascii(seq):
    if isinstance(seq, str):
        seq = seq.encode()
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


# ---------------------------------------------------------------- k-mer codec
_ENC = {"A": 0, "C": 1, "G": 2, "T": 3}
_DEC = "ACGT"


def encode(kmer):
    v = 0
    for ch in kmer.upper():
        v = (v << 2) | _ENC[ch]
    return v


def decode(key, k):
    return "".join(_DEC[(int(key) >> (2 * (k - 1 - i))) & 3] for i in range(k))


def revcomp_key(key, k):
    v = 0
    key = int(key)
    for _ in range(k):
        v = (v << 2) | (3 - (key & 3))
        key >>= 2
    return v


# ---------------------------------------------------------------- D1
def count(seq, k, lower=1, nthreads=1):
    """Canonical k-mer dump of one chromosome: (keys ascending uint64, counts uint32)."""
    a = _ascii(seq)
    L = lib()
    h = L.spo_count(_p(a), a.size, k, nthreads)
    if not h:
        raise ValueError(L.spo_last_error().decode())
    try:
        n = L.spo_counts_n(h, lower)
        keys = np.empty(n, np.uint64)
        cnts = np.empty(n, np.uint32)
        L.spo_counts_fetch(h, lower, _p(keys), _p(cnts))
    finally:
        L.spo_counts_free(h)
    return keys, cnts


def count_bruteforce(seq, k, lower=1):
    """Definition-level counter (pure Python): strand-collapsed k-mer multiset,
    window broken by any non-ACGT byte, case-insensitive."""
    if isinstance(seq, (bytes, bytearray)):
        seq = seq.decode()
    elif not isinstance(seq, str):
        seq = bytes(np.asarray(seq, dtype=np.uint8)).decode()
    s = seq.upper()
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    cnt = Counter()
    for i in range(len(s) - k + 1):
        w = s[i:i + k]
        if any(ch not in comp for ch in w):
            continue
        rc = "".join(comp[ch] for ch in reversed(w))
        cnt[min(w, rc)] += 1
    items = sorted((encode(w), c) for w, c in cnt.items() if c >= lower)
    keys = np.array([x for x, _ in items], dtype=np.uint64)
    cnts = np.array([c for _, c in items], dtype=np.uint32)
    return keys, cnts


# ---------------------------------------------------------------- D2
class Filtered:
    pass


def sets_to_csr(sgs, labels):
    """sgs: list of sets, each a list of units, each a list of chromosome ids
    (the structure SGConfig yields, __main__.py:752-782)."""
    idx = {lab: i for i, lab in enumerate(labels)}
    set_off, unit_off, unit_chrom = [0], [0], []
    for sg in sgs:
        for chrs in sg:
            unit_chrom += [idx[c] for c in chrs]
            unit_off.append(len(unit_chrom))
        set_off.append(len(unit_off) - 1)
    return (np.array(set_off, np.int32), np.array(unit_off, np.int32),
            np.array(unit_chrom, np.int32))


def filter_dumps(dumps, sgs, labels, min_fold=2, baseline=1, min_freq=200, max_freq=1e9,
                 ratio=1, min_prop=None, max_prop=None, lengths=None):
    """dumps: list (per chromosome) of (keys ascending, counts)."""
    L = lib()
    Cn = len(dumps)
    off = np.zeros(Cn + 1, np.int64)
    for i, (k_, _) in enumerate(dumps):
        off[i + 1] = off[i] + len(k_)
    keys_all = np.concatenate([np.asarray(d[0], np.uint64) for d in dumps]) if Cn else np.empty(0, np.uint64)
    cnts_all = np.concatenate([np.asarray(d[1], np.uint32) for d in dumps]) if Cn else np.empty(0, np.uint32)
    keys_all = np.ascontiguousarray(keys_all)
    cnts_all = np.ascontiguousarray(cnts_all)
    tot_lens = int(cnts_all.astype(np.int64).sum())
    if min_prop is not None:   # Jellyfish.py:467-470
        min_freq = min_prop * tot_lens
    if max_prop is not None:   # Jellyfish.py:471-473
        max_freq = max_prop * tot_lens
    set_off, unit_off, unit_chrom = sets_to_csr(sgs, labels)
    h = L.spo_filter(Cn, _p(off), _p(keys_all), _p(cnts_all), len(set_off) - 1, _p(set_off),
                     _p(unit_off), _p(unit_chrom), float(min_fold), int(baseline),
                     float(min_freq), float(max_freq), float(ratio),
                     _p(np.ascontiguousarray(lengths, np.int64)) if lengths is not None else None)
    if not h:
        raise ValueError(L.spo_last_error().decode())
    try:
        r = Filtered()
        r.n_union = L.spo_filtered_n_union(h)
        M = L.spo_filtered_n_rows(h)
        nh = L.spo_filtered_n_hist(h)
        r.lengths = np.empty(Cn, np.int64)
        L.spo_filtered_lengths(h, _p(r.lengths))
        r.keys = np.empty(M, np.uint64)
        r.counts = np.empty((M, Cn), np.uint32)
        r.freqs = np.empty((M, Cn), np.float64)
        r.tot = np.empty(M, np.uint64)
        L.spo_filtered_fetch(h, _p(r.keys), _p(r.counts), _p(r.freqs), _p(r.tot))
        r.hist = np.empty(nh, np.uint64)
        L.spo_filtered_hist(h, _p(r.hist))
    finally:
        L.spo_filtered_free(h)
    return r


# ---------------------------------------------------------------- D3 / D4
def n_slots(length, bin_size, chunk_size, k):
    nb = (max(length, 1) + bin_size - 1) // bin_size
    nch = ((max(length, 1) + (k - 1)) // chunk_size + 1) if chunk_size else 1
    return int(nb + nch)


def map_bins(seq, k, lab_keys, lab_sg, S, bin_size=10000, chunk_size=10_000_000, nthreads=1):
    """Returns (slot_counts [nslots,S] int32, hit [n_lab] uint8, n_mapped)."""
    a = _ascii(seq)
    lab_keys = np.ascontiguousarray(lab_keys, np.uint64)
    lab_sg = np.ascontiguousarray(lab_sg, np.uint8)
    ns = n_slots(a.size, bin_size, chunk_size, k)
    out = np.zeros((ns, S), np.int32)
    hit = np.zeros(len(lab_keys), np.uint8)
    n = lib().spo_map_bins(_p(a), a.size, k, _p(lab_keys), _p(lab_sg), len(lab_keys), S,
                           bin_size, chunk_size, _p(out), ns, _p(hit), nthreads)
    return out, hit, n


def stack_windows(starts, counts, window_size):
    """Circos.stack_matrix for one chromosome (Circos.py:734-742,831-842):
    window = int(START // window_size), rows in order of first appearance."""
    starts = np.asarray(starts, np.int64)
    counts = np.asarray(counts, np.int64)
    win = starts // int(window_size)
    order = []
    acc = {}
    for w, row in zip(win.tolist(), counts):
        if w not in acc:
            acc[w] = row.copy()
            order.append(w)
        else:
            acc[w] += row
    coords = [(int(w * window_size), int(w * window_size + window_size)) for w in order]
    mat = np.array([acc[w] for w in order], np.int64).reshape(len(order), counts.shape[1] if counts.ndim == 2 else 0)
    return coords, mat


# ---------------------------------------------------------------- D5
def hypergeom_right_tail(a, b, c, d):
    return lib().spo_hypergeom_right_tail(int(a), int(b), int(c), int(d))


def fisher_cells(each, total, j):
    each = np.ascontiguousarray(each, np.int64)
    total = np.ascontiguousarray(total, np.int64)
    out = np.zeros(4, np.int64)
    lib().spo_fisher_cells(_p(each), _p(total), len(each), j, _p(out))
    return tuple(int(x) for x in out)


def enrich(counts, max_pval=0.05, min_ratio=0.5):
    counts = np.ascontiguousarray(counts, np.int64)
    W, S = counts.shape
    pvals = np.empty((W, S), np.float64)
    ratios = np.empty((W, S), np.float64)
    argmin = np.empty(W, np.int32)
    sig = np.empty(W, np.uint8)
    with np.errstate(all="ignore"):
        lib().spo_enrich(_p(counts), W, S, float(max_pval), float(min_ratio), _p(pvals),
                         _p(argmin), _p(sig), _p(ratios))
    return pvals, argmin, sig.astype(bool), ratios


def bh_correct(pvals):
    """statsmodels multipletests(method='fdr_bh')[1] (Stats.py:11-12): step-up BH."""
    p = np.asarray(pvals, np.float64)
    n = p.size
    if n == 0:
        return p.copy()
    order = np.argsort(p, kind="stable")
    ps = p[order]
    ecdf = np.arange(1, n + 1) / float(n)
    q = ps / ecdf
    q = np.minimum.accumulate(q[::-1])[::-1]
    q[q > 1] = 1
    out = np.empty(n, np.float64)
    out[order] = q
    return out
