"""ctypes binding of libsubphaser_hip.so (C-ABI declared in include/subphaser_hip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is
visible, every entry point raises.  PyTorch is never imported here; callers
that want torch-owned memory/streams pass raw pointers (tensor.data_ptr(),
torch.cuda.current_stream().cuda_stream).
"""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SUBPHASER_HIP_LIB") or os.path.join(_HERE, "lib", "libsubphaser_hip.so")

SP_OK, SP_EINVAL, SP_EUNSUP, SP_ENOMEM, SP_EHIP, SP_ENODEV, SP_ESTATE, SP_EIO = 0, -1, -2, -3, -4, -5, -6, -7

# every symbol include/subphaser_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "sp_version", "sp_last_error", "sp_ctx_create", "sp_ctx_destroy", "sp_sync", "sp_stream",
    "sp_genome_reset", "sp_genome_add", "sp_genome_add_device", "sp_genome_len", "sp_genome_unpack",
    "sp_count", "sp_count_range", "sp_count_recounts", "sp_nslots", "sp_tables_bind", "sp_table_overflow", "sp_table_merge", "sp_table_lengths", "sp_lengths", "sp_dump_size", "sp_dump",
    "sp_filter_view", "sp_filter", "sp_filter_fetch", "sp_filter_fetch_async", "sp_filter_fetch_wait", "sp_filter_fetch_device", "sp_filter_hist",
    "sp_labels_set", "sp_labels_set_device", "sp_map_nslots", "sp_map_bins", "sp_map_bins_all", "sp_stack_windows", "sp_stack_windows_dev", "sp_stack_enrich", "sp_map_features", "sp_map_intervals", "sp_labels_hit",
    "sp_enrich", "sp_enrich_dev", "sp_kmer_ttest",
    "sp_sparse_sizes", "sp_sparse_sample", "sp_sparse_split", "sp_sparse_export", "sp_sparse_view",
    "sp_prof_enable", "sp_prof_reset", "sp_prof_report",
    "sp_synth_chrom", "sp_synth_chrom_range", "sp_host_alloc", "sp_host_free", "sp_host_register", "sp_host_unregister", "sp_dev_alloc", "sp_dev_free", "sp_dev_copy_to_host", "sp_dev_copy_from_host",
    "sp_fasta_open", "sp_fasta_counts", "sp_fasta_fetch", "sp_fasta_close",
    "sp_text_kmer_matrix", "sp_text_sig_kmers", "sp_text_repr", "sp_text_table",
]


def csrc_fingerprint():
    """sha256 (first 16 hex digits) over the kernel sources: profiles/*_pmc.json are stamped with it, and bench.py
    quotes their measured HBM traffic only while the sources are the ones that were profiled"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class NativeError(RuntimeError):
    """HIP/runtime failure inside libsubphaser_hip.so."""


_lib = None


def load():
    """dlopen the library and declare prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "libsubphaser_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C subphaser_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, dbl, ci = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int
    P = C.POINTER
    L.sp_version.restype = ci
    L.sp_last_error.restype = C.c_char_p
    L.sp_last_error.argtypes = [vp]
    L.sp_ctx_create.argtypes = [ci, vp, P(vp)]
    L.sp_ctx_destroy.argtypes = [vp]
    L.sp_sync.argtypes = [vp]
    L.sp_stream.restype = vp
    L.sp_stream.argtypes = [vp]
    L.sp_genome_reset.argtypes = [vp, ci]
    L.sp_genome_add.argtypes = [vp, ci, vp, i64]
    L.sp_genome_add_device.argtypes = [vp, ci, vp, i64]
    L.sp_genome_len.argtypes = [vp, ci, P(i64)]
    L.sp_genome_unpack.argtypes = [vp, ci, vp, i64]
    L.sp_count.argtypes = [vp, ci, ci, ci]
    L.sp_count_range.argtypes = [vp, ci, ci, ci, ci, ci]
    L.sp_nslots.argtypes = [vp, ci, P(i64)]
    L.sp_count_recounts.argtypes = [vp, P(i64)]
    L.sp_tables_bind.argtypes = [vp, ci, vp]
    L.sp_table_overflow.argtypes = [vp, ci, vp, i64, P(i64)]
    L.sp_table_merge.argtypes = [vp, vp, vp, i64, vp, vp, i64, i64, i64, vp, i64, P(i64)]
    L.sp_table_lengths.argtypes = [vp, vp, vp, i64, i64, i64, ci, P(i64), P(i64)]
    L.sp_filter_view.argtypes = [vp, ci, vp, i64, i64, vp, ci, ci, vp, vp]
    L.sp_lengths.argtypes = [vp, vp]
    L.sp_dump_size.argtypes = [vp, ci, P(i64)]
    L.sp_dump.argtypes = [vp, ci, vp, vp, i64, P(i64)]
    L.sp_filter.argtypes = [vp, ci, vp, vp, vp, dbl, ci, dbl, dbl, dbl, P(i64), P(i64), P(i64)]
    L.sp_filter_fetch.argtypes = [vp, vp, vp, vp, vp, i64]
    L.sp_filter_fetch_device.argtypes = [vp, vp, vp, vp, i64]
    L.sp_filter_fetch_async.argtypes = [vp, vp, vp, vp, i64]
    L.sp_filter_fetch_wait.argtypes = [vp]
    L.sp_filter_hist.argtypes = [vp, vp, i64]
    L.sp_labels_set.argtypes = [vp, vp, vp, i64, ci]
    L.sp_labels_set_device.argtypes = [vp, vp, vp, i64, ci]
    L.sp_map_nslots.argtypes = [vp, ci, i64, i64, P(i64)]
    L.sp_map_bins.argtypes = [vp, ci, i64, i64, vp, i64, P(i64)]
    L.sp_map_bins_all.argtypes = [vp, i64, i64, vp, vp, vp]
    L.sp_stack_windows.argtypes = [vp, i64, i64, i64, vp, vp, vp]
    L.sp_stack_windows_dev.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp]
    L.sp_stack_enrich.argtypes = [vp, i64, i64, i64, vp, vp, dbl, dbl, vp, vp, vp, vp, vp]
    L.sp_map_features.argtypes = [vp, vp, vp, i64, vp]
    L.sp_map_intervals.argtypes = [vp, vp, vp, vp, i64, vp]
    L.sp_host_alloc.argtypes = [vp, i64, P(vp)]
    L.sp_host_free.argtypes = [vp, vp]
    L.sp_labels_hit.argtypes = [vp, P(i64)]
    L.sp_enrich.argtypes = [vp, vp, i64, ci, dbl, dbl, vp, vp, vp, vp]
    L.sp_enrich_dev.argtypes = [vp, vp, i64, ci, dbl, dbl, vp, vp, vp, vp]
    L.sp_kmer_ttest.argtypes = [vp, vp, i64, ci, vp, ci, vp, vp, vp, vp, vp, vp]
    L.sp_sparse_sizes.argtypes = [vp, vp]
    L.sp_sparse_sample.argtypes = [vp, ci, i64, vp, P(i64)]
    L.sp_sparse_split.argtypes = [vp, ci, vp, ci, vp]
    L.sp_sparse_export.argtypes = [vp, ci, i64, i64, vp, vp]
    L.sp_sparse_view.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci]
    L.sp_prof_enable.argtypes = [vp, ci]
    L.sp_prof_reset.argtypes = [vp]
    L.sp_prof_report.argtypes = [vp, C.c_char_p, i64]
    L.sp_synth_chrom.argtypes = [vp, vp, i64, C.c_uint64, ci, ci, ci, ci, ci]
    L.sp_synth_chrom_range.argtypes = [vp, vp, i64, i64, i64, C.c_uint64, ci, ci, ci, ci, ci]
    L.sp_dev_alloc.argtypes = [vp, i64, P(vp)]
    L.sp_dev_free.argtypes = [vp, vp]
    L.sp_dev_copy_to_host.argtypes = [vp, vp, vp, i64]
    L.sp_host_register.argtypes = [vp, vp, i64]
    L.sp_host_unregister.argtypes = [vp, vp]
    L.sp_dev_copy_from_host.argtypes = [vp, vp, vp, i64]
    L.sp_fasta_open.argtypes = [vp, i64, ci, P(vp)]
    L.sp_fasta_counts.argtypes = [vp, P(i64), P(i64)]
    L.sp_fasta_fetch.argtypes = [vp, vp, vp, vp, vp]
    L.sp_fasta_close.argtypes = [vp]
    L.sp_fasta_close.restype = None
    L.sp_text_kmer_matrix.argtypes = [vp, ci, vp, i64, ci, ci, ci, P(i64)]
    L.sp_text_sig_kmers.argtypes = [vp, ci, vp, C.c_char_p, ci, vp, vp, ci, i64, ci, ci, P(i64)]
    L.sp_text_repr.argtypes = [vp, i64, vp, vp]
    L.sp_text_table.argtypes = [vp, ci, i64, ci, ci, P(i64)]
    for name in SYMBOLS:
        if name not in ("sp_last_error", "sp_stream", "sp_fasta_close"):
            getattr(L, name).restype = ci
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def fasta_scan(data, threads=None, out=None):
    """Records of a FASTA file image (uint8 array / memmap / bytes) through the library's host-side scanner:
    returns (ids, cat, off) -- ids list of str (first token of each header), cat uint8 array of all sequences back
    to back without line breaks or blanks, off int64 [n + 1].  `out(n_bases)` may supply the buffer for cat (e.g.
    page-locked memory).  No GPU is involved."""
    L = load()
    data = np.frombuffer(data, np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data
    if data.dtype != np.uint8 or not data.flags.c_contiguous:
        data = np.ascontiguousarray(data, np.uint8)
    n = int(data.size)
    if threads is None:
        threads = min(32, len(os.sched_getaffinity(0)))
    h = C.c_void_p()
    rc = L.sp_fasta_open(C.c_void_p(data.ctypes.data) if n else None, n, int(threads), C.byref(h))
    if rc == SP_ENOMEM:
        raise MemoryError("sp_fasta_open")
    if rc:
        raise ValueError("sp_fasta_open: bad arguments")
    try:
        nr, nb = C.c_int64(), C.c_int64()
        L.sp_fasta_counts(h, C.byref(nr), C.byref(nb))
        nr, nb = nr.value, nb.value
        hs, he, off = np.empty(nr, np.int64), np.empty(nr, np.int64), np.empty(nr + 1, np.int64)
        cat = out(nb) if out is not None else np.empty(nb, np.uint8)
        if L.sp_fasta_fetch(h, _p(hs), _p(he), _p(off), _p(cat) if nb else None):
            raise ValueError("sp_fasta_fetch: bad arguments")
    finally:
        L.sp_fasta_close(h)
    ids = []
    for a, b in zip(hs.tolist(), he.tolist()):
        t = bytes(data[a:b]).split()
        ids.append(t[0].decode() if t else "")
    return ids, cat, off


def _fd_of(fout):
    """file descriptor of a real file object (flushed), or None (StringIO, sys.stdout wrappers without one...)"""
    try:
        fd = fout.fileno()
        fout.flush()
        return fd
    except Exception:
        return None


def _text_threads():
    return min(64, len(os.sched_getaffinity(0)))


def _text_fail(name, rc):
    msg = load().sp_last_error(None).decode(errors="replace")
    if rc == SP_ENOMEM:
        raise MemoryError("%s: %s" % (name, msg or "out of memory"))
    raise OSError("%s failed (%d): %s" % (name, rc, msg))


def text_kmer_matrix(fout, keys, k, freqs):
    """rows of `.kmer.mat` through the library's threaded writer; False when fout has no file descriptor"""
    fd = _fd_of(fout)
    if fd is None:
        return False
    keys = np.ascontiguousarray(keys, np.uint64)
    freqs = np.ascontiguousarray(freqs, np.float64)
    M, Cn = freqs.shape
    rc = load().sp_text_kmer_matrix(_p(keys), int(k), _p(freqs), M, Cn, _text_threads(), fd, None)
    if rc:
        _text_fail("sp_text_kmer_matrix", rc)
    return True


def text_sig_kmers(fout, keys, k, top, names, pvals, means):
    fd = _fd_of(fout)
    if fd is None:
        return False
    keys = np.ascontiguousarray(keys, np.uint64)
    top = np.ascontiguousarray(top, np.int32)
    pvals = np.ascontiguousarray(pvals, np.float64)
    means = np.ascontiguousarray(means, np.float64)
    blob = b"".join(n.encode() + b"\0" for n in names)
    rc = load().sp_text_sig_kmers(_p(keys), int(k), _p(top), blob, len(names), _p(pvals), _p(means), means.shape[1],
                                  len(keys), _text_threads(), fd, None)
    if rc:
        _text_fail("sp_text_sig_kmers", rc)
    return True


class _TextCol(C.Structure):
    _fields_ = [("kind", C.c_int), ("width", C.c_int), ("join", C.c_char), ("data", C.c_void_p),
                ("off", C.c_void_p), ("names", C.c_void_p)]


def str_blob(strings):
    """list of str -> (uint8 blob, int64 offsets [n + 1]) for a SP_COL_STR / SP_COL_NAME column"""
    enc = [s.encode() for s in strings]
    off = np.zeros(len(enc) + 1, np.int64)
    if enc:
        np.cumsum(np.fromiter((len(e) for e in enc), np.int64, len(enc)), out=off[1:])
    return np.frombuffer(b"".join(enc) or b"\0", np.uint8), off


def write_chunks(fout, n_rows, format_rows, chunk=50000):
    """Fallback of the threaded text writers below for targets that are not real files (StringIO, wrapped stdout):
    format_rows(lo, hi) -> str for rows [lo, hi), written to fout in order, in this process (a process that owns a
    HIP context must not fork worker pools: DESIGN.md section 7b)."""
    for lo in range(0, max(0, n_rows), chunk):
        fout.write(format_rows(lo, min(lo + chunk, n_rows)))


def text_table(fout, n_rows, cols):
    """Tab-separated rows through the library's threaded writer (sp_text_table).  cols: list of
    ("str", blob, off) | ("i64", array [n x w], join) | ("f64", array [n x w], join) | ("name", idx int32 [n], names)
    | ("ival", int64 [n x 3] = name index, start, end, names).
    Returns False when fout has no file descriptor (the caller formats in Python then)."""
    fd = _fd_of(fout)
    if fd is None:
        return False
    arr = (_TextCol * len(cols))()
    keep = []
    for c, col in zip(arr, cols):
        kind = col[0]
        if kind == "str":
            blob, off = np.ascontiguousarray(col[1], np.uint8), np.ascontiguousarray(col[2], np.int64)
            assert off.size == n_rows + 1
            c.kind, c.width, c.join, c.data, c.off = 0, 1, b"\t", blob.ctypes.data, off.ctypes.data
            keep += [blob, off]
        elif kind in ("i64", "f64"):
            a = np.ascontiguousarray(col[1], np.int64 if kind == "i64" else np.float64).reshape(n_rows, -1)
            c.kind, c.width, c.join, c.data = (1 if kind == "i64" else 2), max(1, a.shape[1]), col[2].encode(), a.ctypes.data
            keep.append(a)
        elif kind == "ival":      # (name index, start, end) rows printed `name:start-end`
            a = np.ascontiguousarray(col[1], np.int64).reshape(n_rows, 3)
            blob, off = str_blob(col[2])
            c.kind, c.width, c.join, c.data, c.off, c.names = 4, len(col[2]), b"\t", a.ctypes.data, off.ctypes.data, blob.ctypes.data
            keep += [a, blob, off]
        elif kind == "name":
            idx = np.ascontiguousarray(col[1], np.int32)
            blob, off = str_blob(col[2])
            assert idx.size == n_rows
            c.kind, c.width, c.join, c.data, c.off, c.names = 3, len(col[2]), b"\t", idx.ctypes.data, off.ctypes.data, blob.ctypes.data
            keep += [idx, blob, off]
        else:
            raise ValueError(kind)
    rc = load().sp_text_table(arr, len(cols), int(n_rows), _text_threads(), fd, None)
    if rc:
        _text_fail("sp_text_table", rc)
    return True


def text_repr(x):
    """[repr(v) for v in x] through the library (test hook)"""
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty(max(1, x.size * 40), np.uint8)
    off = np.empty(x.size + 1, np.int64)
    load().sp_text_repr(_p(x), x.size, _p(out), _p(off))
    b = out.tobytes()
    return [b[off[i]:off[i + 1]].decode() for i in range(x.size)]


def as_ascii(seq):
    """str / bytes / uint8 array -> contiguous uint8 array (no copy when possible)."""
    if isinstance(seq, str):
        seq = seq.encode()
    if isinstance(seq, (bytes, bytearray, memoryview)):
        return np.frombuffer(seq, dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


class Context:
    """One GPU context (one per process per GPU)."""

    def __init__(self, device=0, stream=None):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.sp_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != SP_OK:
            raise NativeError("sp_ctx_create failed (%d): %s" % (rc, self.L.sp_last_error(None).decode()))
        self.h = h
        self.n_chrom = 0
        self.k = None
        self._pinned = {}     # tag -> (ptr, nbytes): reusable page-locked staging buffers

    # -------------------------------------------------------------- plumbing
    def _ck(self, rc):
        if rc == SP_OK:
            return
        msg = self.L.sp_last_error(self.h).decode()
        if rc in (SP_ESTATE, SP_EINVAL, SP_EUNSUP):
            raise ValueError(msg)       # the reference raises ValueError for these
        if rc == SP_ENOMEM:
            raise MemoryError(msg)
        raise NativeError("libsubphaser_hip error %d: %s" % (rc, msg))

    def close(self):
        if getattr(self, "h", None):
            for d_keys, d_sg in getattr(self, "_label_copies", {}).values():     # device copies of label sets (seqs.KmerLabels)
                self.L.sp_dev_free(self.h, C.c_void_p(d_keys))
                self.L.sp_dev_free(self.h, C.c_void_p(d_sg))
            self._label_copies = {}
            for ptr, _ in self._pinned.values():
                self.L.sp_host_free(self.h, C.c_void_p(ptr))
            self._pinned = {}
            self.L.sp_ctx_destroy(self.h)
            self.h = None

    def pinned_empty(self, tag, shape, dtype):
        """numpy array over page-locked memory owned by the context.  The buffer named `tag` is
        reused (and overwritten) by the next request with the same tag."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        ptr, cap = self._pinned.get(tag, (None, 0))
        if n > cap or ptr is None:       # (a request for zero rows still gets a real buffer)
            if ptr:
                self.L.sp_host_free(self.h, C.c_void_p(ptr))
            p = C.c_void_p()
            want = n + n // 8 + 4096
            self._ck(self.L.sp_host_alloc(self.h, want, C.byref(p)))
            ptr, cap = p.value, want
            self._pinned[tag] = (ptr, cap)
        buf = (C.c_uint8 * max(n, 1)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._ck(self.L.sp_sync(self.h))

    @property
    def stream(self):
        return self.L.sp_stream(self.h)

    # -------------------------------------------------------------- genome
    def genome_reset(self, n_chrom):
        self._ck(self.L.sp_genome_reset(self.h, int(n_chrom)))
        self.n_chrom = int(n_chrom)

    def genome_add(self, chrom, seq):
        a = as_ascii(seq)
        self._ck(self.L.sp_genome_add(self.h, int(chrom), _p(a), a.size))

    def genome_add_device(self, chrom, d_ptr, length):
        self._ck(self.L.sp_genome_add_device(self.h, int(chrom), C.c_void_p(int(d_ptr)), int(length)))

    def genome_len(self, chrom):
        n = C.c_int64()
        self._ck(self.L.sp_genome_len(self.h, int(chrom), C.byref(n)))
        return n.value

    def genome_unpack(self, chrom):
        n = self.genome_len(chrom)
        out = np.empty(n, np.uint8)
        self._ck(self.L.sp_genome_unpack(self.h, int(chrom), _p(out), n))
        return out

    # -------------------------------------------------------------- count
    def count(self, k, lower_count=3, engine=0):
        self._ck(self.L.sp_count(self.h, int(k), int(lower_count), int(engine)))
        self.k = int(k)

    def count_range(self, k, lower_count, engine, first, last):
        self._ck(self.L.sp_count_range(self.h, int(k), int(lower_count), int(engine), int(first), int(last)))
        self.k = int(k)

    def count_recounts(self):
        """chromosomes engine 2 counted twice so far (a partition bucket outgrew its sampled region)"""
        n = C.c_int64()
        self._ck(self.L.sp_count_recounts(self.h, C.byref(n)))
        return n.value

    def nslots(self, k):
        n = C.c_int64()
        self._ck(self.L.sp_nslots(self.h, int(k), C.byref(n)))
        return n.value

    def tables_bind(self, chrom, d_ptr):
        self._ck(self.L.sp_tables_bind(self.h, int(chrom), C.c_void_p(int(d_ptr)) if d_ptr else None))

    def table_overflow(self, chrom, d_pairs=None, cap=0):
        """Overflow pairs (uint32 slot, uint32 raw count >= 255; ascending slot) of the byte table of `chrom`:
        their number; with d_pairs (device pointer, capacity `cap` pairs) they are copied there (async)."""
        n = C.c_int64()
        self._ck(self.L.sp_table_overflow(self.h, int(chrom), C.c_void_p(int(d_pairs)) if d_pairs else None, int(cap),
                                          C.byref(n)))
        return n.value

    def table_merge(self, d_dst, d_dst_ovf, n_dst_ovf, d_src, d_src_ovf, n_src_ovf, slot_base, n, d_out_ovf, cap):
        """dst += src over the byte slices of slots [slot_base, slot_base + n) (device pointers); returns the
        number of pairs of the merged overflow list written to d_out_ovf (MemoryError: capacity too small)."""
        m = C.c_int64()
        self._ck(self.L.sp_table_merge(self.h, C.c_void_p(int(d_dst)), C.c_void_p(int(d_dst_ovf)) if n_dst_ovf else None,
                                       int(n_dst_ovf), C.c_void_p(int(d_src)),
                                       C.c_void_p(int(d_src_ovf)) if n_src_ovf else None, int(n_src_ovf), int(slot_base),
                                       int(n), C.c_void_p(int(d_out_ovf)) if cap else None, int(cap), C.byref(m)))
        return m.value

    def table_lengths(self, d_tab, d_ovf, n_ovf, slot_base, n, lower_count):
        """(sum, number) of the counts >= lower_count of a byte slice."""
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.L.sp_table_lengths(self.h, C.c_void_p(int(d_tab)), C.c_void_p(int(d_ovf)) if n_ovf else None,
                                         int(n_ovf), int(slot_base), int(n), int(lower_count), C.byref(a), C.byref(b)))
        return a.value, b.value

    def filter_view(self, d_ptrs, slot_base, nslots_view, lengths, k, lower_count, d_ovf=None, n_ovf=None):
        """Point sp_filter at slot-range slices of byte tables (device pointers, one per chromosome of the
        whole genome) + their overflow lists (device pointers / pair counts).  d_ptrs=None returns to the
        local tables."""
        if d_ptrs is None:
            self._ck(self.L.sp_filter_view(self.h, 0, None, 0, 0, None, 0, 0, None, None))
            self._view_C = None
            return
        arr = (C.c_void_p * len(d_ptrs))(*[int(p) for p in d_ptrs])
        lengths = np.ascontiguousarray(lengths, np.int64)
        oarr, narr = None, None
        if d_ovf is not None:
            oarr = (C.c_void_p * len(d_ptrs))(*[int(p) if p else None for p in d_ovf])
            narr = _p(np.ascontiguousarray(n_ovf, np.int64))
        self._ck(self.L.sp_filter_view(self.h, len(d_ptrs), arr, int(slot_base), int(nslots_view), _p(lengths),
                                       int(k), int(lower_count), oarr, narr))
        self._view_C = len(d_ptrs)
        self.k = int(k)

    # -------------------------------------------------------------- multi-GPU, k > 15
    def sparse_sizes(self):
        out = np.zeros(self.n_chrom, np.int64)
        if self.n_chrom:
            self._ck(self.L.sp_sparse_sizes(self.h, _p(out)))
        return out

    def sparse_sample(self, chrom, n_samples):
        out = np.empty(int(n_samples), np.uint64)
        n = C.c_int64()
        self._ck(self.L.sp_sparse_sample(self.h, int(chrom), int(n_samples), _p(out), C.byref(n)))
        return out[:n.value]

    def sparse_split(self, chrom, splitters):
        sp = np.ascontiguousarray(splitters, np.uint64)
        bounds = np.zeros(sp.size + 2, np.int64)
        self._ck(self.L.sp_sparse_split(self.h, int(chrom), _p(sp), int(sp.size), _p(bounds)))
        return bounds

    def sparse_export(self, chrom, first, count, d_keys, d_counts):
        v = lambda p: C.c_void_p(int(p)) if p else None
        self._ck(self.L.sp_sparse_export(self.h, int(chrom), int(first), int(count), v(d_keys), v(d_counts)))

    def sparse_view(self, d_keys, d_counts, n, lengths, k, lower_count):
        """Point sp_filter at caller-owned device lists (one key range of every chromosome of the
        whole genome).  d_keys=None returns to the local chromosomes."""
        if d_keys is None:
            self._ck(self.L.sp_sparse_view(self.h, 0, None, None, None, None, 0, 0))
            self._view_C = None
            return
        ka = (C.c_void_p * len(d_keys))(*[int(p) for p in d_keys])
        ca = (C.c_void_p * len(d_counts))(*[int(p) for p in d_counts])
        n = np.ascontiguousarray(n, np.int64)
        lengths = np.ascontiguousarray(lengths, np.int64)
        self._ck(self.L.sp_sparse_view(self.h, len(d_keys), ka, ca, _p(n), _p(lengths), int(k), int(lower_count)))
        self._view_C = len(d_keys)
        self.k = int(k)

    def lengths(self):
        out = np.zeros(self.n_chrom, np.int64)
        self._ck(self.L.sp_lengths(self.h, _p(out)))
        return out

    def dump_size(self, chrom):
        """number of distinct canonical k-mers with count >= lower_count (lines of the jellyfish dump)"""
        n = C.c_int64()
        self._ck(self.L.sp_dump_size(self.h, int(chrom), C.byref(n)))
        return n.value

    def dump(self, chrom, sort=True):
        """(keys, counts) of one chromosome; canonical keys, ascending when sort=True."""
        n = C.c_int64()
        self._ck(self.L.sp_dump_size(self.h, int(chrom), C.byref(n)))
        keys = np.empty(n.value, np.uint64)
        cnts = np.empty(n.value, np.uint32)
        self._ck(self.L.sp_dump(self.h, int(chrom), _p(keys), _p(cnts), n.value, C.byref(n)))
        if sort and keys.size:
            o = np.argsort(keys, kind="stable")
            keys, cnts = keys[o], cnts[o]
        return keys, cnts

    # -------------------------------------------------------------- filter
    def filter(self, set_off, unit_off, unit_chrom, min_fold, baseline, min_freq, max_freq, ratio):
        set_off = np.ascontiguousarray(set_off, np.int32)
        unit_off = np.ascontiguousarray(unit_off, np.int32)
        unit_chrom = np.ascontiguousarray(unit_chrom, np.int32)
        nu, nr, nh = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.L.sp_filter(self.h, len(set_off) - 1, _p(set_off), _p(unit_off), _p(unit_chrom),
                                  float(min_fold), int(baseline), float(min_freq), float(max_freq),
                                  float(ratio), C.byref(nu), C.byref(nr), C.byref(nh)))
        return nu.value, nr.value, nh.value

    def filter_fetch(self, n_rows, want_freqs=True, sort=True, pinned=False):
        """pinned=True returns views of context-owned page-locked buffers (valid until the next
        filter_fetch): the copy then runs at PCIe speed instead of through pageable staging."""
        Cn = getattr(self, "_view_C", None) or self.n_chrom
        if pinned and not sort:
            keys = self.pinned_empty("ff_keys", (n_rows,), np.uint64)
            counts = self.pinned_empty("ff_counts", (n_rows, Cn), np.uint32)
            tot = self.pinned_empty("ff_tot", (n_rows,), np.uint64)
            freqs = self.pinned_empty("ff_freqs", (n_rows, Cn), np.float64) if want_freqs else None
        else:
            keys = np.empty(n_rows, np.uint64)
            counts = np.empty((n_rows, Cn), np.uint32)
            tot = np.empty(n_rows, np.uint64)
            freqs = np.empty((n_rows, Cn), np.float64) if want_freqs else None
        self._ck(self.L.sp_filter_fetch(self.h, _p(keys), _p(counts), _p(freqs), _p(tot), n_rows))
        if sort and n_rows:
            o = np.argsort(keys, kind="stable")
            keys, counts, tot = keys[o], counts[o], tot[o]
            if freqs is not None:
                freqs = freqs[o]
        return keys, counts, freqs, tot

    def filter_fetch_async(self, n_rows):
        """Rows into page-locked buffers through the copy stream; valid after filter_fetch_wait().
        Returns (keys, counts, tot) views."""
        Cn = getattr(self, "_view_C", None) or self.n_chrom
        keys = self.pinned_empty("ff_keys", (n_rows,), np.uint64)
        counts = self.pinned_empty("ff_counts", (n_rows, Cn), np.uint32)
        tot = self.pinned_empty("ff_tot", (n_rows,), np.uint64)
        self._ck(self.L.sp_filter_fetch_async(self.h, _p(keys), _p(counts), _p(tot), n_rows))
        return keys, counts, tot

    def filter_fetch_async_ptr(self, h_keys, h_counts, n_rows):
        """The same copy into page-locked host memory given by ADDRESS (a registered shared-memory segment: the
        multi-GPU matrix hand-over); the row totals go to a scratch buffer of the context."""
        tot = self.pinned_empty("ff_tot", (max(int(n_rows), 1),), np.uint64)
        self._ck(self.L.sp_filter_fetch_async(self.h, C.c_void_p(int(h_keys)), C.c_void_p(int(h_counts)), _p(tot), int(n_rows)))

    def filter_fetch_wait(self):
        self._ck(self.L.sp_filter_fetch_wait(self.h))

    def filter_fetch_device(self, d_keys, d_counts, d_tot, n_rows):
        """Surviving rows into caller-owned device buffers (raw pointers or None), slot order."""
        v = lambda p: C.c_void_p(int(p)) if p else None
        self._ck(self.L.sp_filter_fetch_device(self.h, v(d_keys), v(d_counts), v(d_tot), int(n_rows)))

    def filter_hist(self, n_hist):
        tot = np.empty(n_hist, np.uint64)
        self._ck(self.L.sp_filter_hist(self.h, _p(tot), n_hist))
        return tot

    # -------------------------------------------------------------- labels / map
    def labels_set(self, keys, sg, n_sg):
        keys = np.ascontiguousarray(keys, np.uint64)
        sg = np.ascontiguousarray(sg, np.uint8)
        assert keys.size == sg.size
        self._ck(self.L.sp_labels_set(self.h, _p(keys), _p(sg), keys.size, int(n_sg)))
        self.n_sg = int(n_sg)

    def labels_set_device(self, d_keys, d_sg, n, n_sg):
        """keys (uint64) and labels (uint8) already in device memory (KmerLabels.on_device)"""
        self._ck(self.L.sp_labels_set_device(self.h, C.c_void_p(int(d_keys)), C.c_void_p(int(d_sg)), int(n), int(n_sg)))
        self.n_sg = int(n_sg)

    def labels_set_from(self, labels, n_sg):
        """a KmerLabels object: through its device copy when it offers one"""
        if hasattr(labels, "on_device"):
            d_keys, d_sg = labels.on_device(self)
            self.labels_set_device(d_keys, d_sg, len(labels.keys), n_sg)
        else:
            self.labels_set(labels.keys, labels.sg_idx, n_sg)

    def map_nslots(self, chrom, bin_size, chunk_size):
        n = C.c_int64()
        self._ck(self.L.sp_map_nslots(self.h, int(chrom), int(bin_size), int(chunk_size), C.byref(n)))
        return n.value

    def map_bins(self, chrom, bin_size=10000, chunk_size=10_000_000):
        ns = self.map_nslots(chrom, bin_size, chunk_size)
        out = np.empty((ns, self.n_sg), np.int32)
        n = C.c_int64()
        self._ck(self.L.sp_map_bins(self.h, int(chrom), int(bin_size), int(chunk_size), _p(out), ns,
                                    C.byref(n)))
        return out, n.value

    def map_bins_all(self, bin_size=10000, chunk_size=10_000_000):
        """All chromosomes in one call: (list of [nslots_c, n_sg] int32 views, n_mapped int64 [C]).
        The views alias one array (self.last_map = (array, slot offsets)) in page-locked memory."""
        off = np.zeros(self.n_chrom + 1, np.int64)
        for i in range(self.n_chrom):
            off[i + 1] = off[i] + self.map_nslots(i, bin_size, chunk_size)
        out = self.pinned_empty("map_all", (int(off[-1]), self.n_sg), np.int32)
        nm = np.zeros(self.n_chrom, np.int64)
        self._ck(self.L.sp_map_bins_all(self.h, int(bin_size), int(chunk_size), _p(off), _p(out), _p(nm)))
        self.last_map = (out, off)
        return [out[off[i]:off[i + 1]] for i in range(self.n_chrom)], nm

    def stack_windows(self, bin_size, chunk_size, window_size, lengths):
        """Window counts of the last map_bins_all, summed on the device.
        Returns (win_counts int64 [total, n_sg], win_off int64 [C+1])."""
        out, off = self.last_map
        woff = np.zeros(self.n_chrom + 1, np.int64)
        for i, n in enumerate(lengths):
            woff[i + 1] = woff[i] + (int(n) + int(window_size) - 1) // int(window_size) + 1
        win = self.pinned_empty("win_all", (int(woff[-1]), self.n_sg), np.int64)
        self._ck(self.L.sp_stack_windows(self.h, int(bin_size), int(chunk_size), int(window_size), _p(off),
                                         _p(woff), _p(win)))
        return win, woff

    def stack_windows_dev(self, bin_size, chunk_size, window_size, win_off, seg_start, d_win):
        """Window counts of the last map_bins_all accumulated into a DEVICE table (d_win: int64 [total, n_sg],
        cleared by the caller).  win_off[i] = first window row of the chromosome local chromosome i is (a piece
        of), seg_start[i] = position of its base 0 inside that chromosome."""
        out, off = self.last_map
        win_off = np.ascontiguousarray(win_off, np.int64)
        seg = np.ascontiguousarray(seg_start, np.int64) if seg_start is not None else None
        self._ck(self.L.sp_stack_windows_dev(self.h, int(bin_size), int(chunk_size), int(window_size), _p(off),
                                             _p(win_off), _p(seg) if seg is not None else None, C.c_void_p(int(d_win))))

    def stack_enrich(self, bin_size, chunk_size, window_size, lengths, max_pval=0.05, min_ratio=0.5):
        """stack_windows + enrich fused on the device.  Returns (win int64 [W, S], win_off [C+1], pvals, argmin,
        sig, ratios) over EVERY window row (empty rows included)."""
        out, off = self.last_map
        woff = np.zeros(self.n_chrom + 1, np.int64)
        for i, n in enumerate(lengths):
            woff[i + 1] = woff[i] + (int(n) + int(window_size) - 1) // int(window_size) + 1
        W, S = int(woff[-1]), self.n_sg
        win = self.pinned_empty("win_all", (W, S), np.int64)
        pvals = self.pinned_empty("enr_p", (W, S), np.float64)
        ratios = self.pinned_empty("enr_q", (W, S), np.float64)
        argmin = self.pinned_empty("enr_a", (W,), np.int32)
        sig = self.pinned_empty("enr_s", (W,), np.uint8)
        self._ck(self.L.sp_stack_enrich(self.h, int(bin_size), int(chunk_size), int(window_size), _p(off), _p(woff),
                                        float(max_pval), float(min_ratio), _p(win), _p(pvals), _p(argmin), _p(sig),
                                        _p(ratios)))
        return win, woff, pvals, argmin, sig, ratios

    def enrich_dev(self, d_counts, W, S, max_pval=0.05, min_ratio=0.5):
        """enrich() for a window table that already lives in device memory (int64 [W, S])."""
        pvals = np.empty((W, S), np.float64)
        ratios = np.empty((W, S), np.float64)
        argmin = np.empty(W, np.int32)
        sig = np.empty(W, np.uint8)
        self._ck(self.L.sp_enrich_dev(self.h, C.c_void_p(int(d_counts)), int(W), int(S), float(max_pval),
                                      float(min_ratio), _p(pvals), _p(argmin), _p(sig), _p(ratios)))
        return pvals, argmin, sig.astype(bool), ratios

    def map_features(self, seqs):
        """seqs: list of str/bytes.  Returns int64 [n_feat, n_sg] totals."""
        arrs = [as_ascii(s) for s in seqs]
        off = np.zeros(len(arrs) + 1, np.int64)
        for i, a in enumerate(arrs):
            off[i + 1] = off[i] + a.size
        cat = np.concatenate(arrs) if arrs else np.empty(0, np.uint8)
        cat = np.ascontiguousarray(cat)
        out = np.zeros((len(arrs), self.n_sg), np.int64)
        self._ck(self.L.sp_map_features(self.h, _p(cat), _p(off), len(arrs), _p(out)))
        return out

    def map_features_cat(self, cat, off):
        """Features lying back to back in `cat` (uint8), feature f = cat[off[f]:off[f+1]].
        Returns int64 [n_feat, n_sg] totals."""
        cat = np.ascontiguousarray(cat, np.uint8)
        off = np.ascontiguousarray(off, np.int64)
        n = off.size - 1
        out = np.zeros((n, self.n_sg), np.int64)
        self._ck(self.L.sp_map_features(self.h, _p(cat), _p(off), n, _p(out)))
        return out

    def map_intervals(self, chrom, start, end):
        """BED-style intervals over the resident genome: int64 [n, n_sg] counts of labelled k-mer starts s with
        start <= s and s + k <= end on chromosome index chrom (0-based, half-open) -- sp_map_features' totals for the
        same sub-sequences, without uploading them."""
        chrom = np.ascontiguousarray(chrom, np.int32)
        start = np.ascontiguousarray(start, np.int64)
        end = np.ascontiguousarray(end, np.int64)
        assert chrom.size == start.size == end.size
        out = np.zeros((chrom.size, self.n_sg), np.int64)
        self._ck(self.L.sp_map_intervals(self.h, _p(chrom), _p(start), _p(end), chrom.size, _p(out)))
        return out

    def labels_hit(self):
        n = C.c_int64()
        self._ck(self.L.sp_labels_hit(self.h, C.byref(n)))
        return n.value

    # -------------------------------------------------------------- enrich
    def enrich(self, counts, max_pval=0.05, min_ratio=0.5):
        counts = np.ascontiguousarray(counts, np.int64)
        if counts.ndim != 2:
            raise ValueError("counts must be W x S")
        W, S = counts.shape
        pvals = np.empty((W, S), np.float64)
        ratios = np.empty((W, S), np.float64)
        argmin = np.empty(W, np.int32)
        sig = np.empty(W, np.uint8)
        self._ck(self.L.sp_enrich(self.h, _p(counts), W, S, float(max_pval), float(min_ratio), _p(pvals),
                                  _p(argmin), _p(sig), _p(ratios)))
        return pvals, argmin, sig.astype(bool), ratios

    def kmer_ttest(self, counts, lengths, groups):
        """Cluster.output_kmers' per-k-mer t-test on the device.  counts: uint32 [M, C] (thresholded, as
        filter_fetch returns them), lengths: int64 [C], groups: list of chromosome-index lists in sorted
        subgenome-name order.  Returns (top, second, pvals, means [M, n_groups]).
        counts may also be (device pointer, M, C): rows staged on the device earlier (stage_rows)."""
        if isinstance(counts, tuple):
            d_ptr, M, Cn = counts
            cptr = C.c_void_p(int(d_ptr))
        else:
            counts = np.ascontiguousarray(counts, np.uint32)
            M, Cn = counts.shape
            cptr = _p(counts)
        lengths = np.ascontiguousarray(lengths, np.int64)
        goff = np.zeros(len(groups) + 1, np.int32)
        goff[1:] = np.cumsum([len(g) for g in groups])
        gch = np.ascontiguousarray(np.concatenate([np.asarray(g, np.int32) for g in groups]), np.int32)
        top, second = np.empty(M, np.int32), np.empty(M, np.int32)
        pvals, means = np.empty(M, np.float64), np.empty((M, len(groups)), np.float64)
        self._ck(self.L.sp_kmer_ttest(self.h, cptr, M, Cn, _p(lengths), len(groups), _p(goff), _p(gch),
                                      _p(top), _p(second), _p(pvals), _p(means)))
        return top, second, pvals, means

    def stage_rows(self, counts):
        """Copy a uint32 [M, C] matrix to a device buffer owned by the context (one at a time; the previous one is
        released) and return (device pointer, M, C) for kmer_ttest."""
        counts = np.ascontiguousarray(counts, np.uint32)
        old = getattr(self, "_staged_rows", None)
        if old:
            self.dev_free(old)
            self._staged_rows = None
        if counts.size == 0:
            return None
        self._staged_rows = self.dev_alloc(counts.nbytes)
        self.host_to_dev(self._staged_rows, counts)
        return (self._staged_rows, counts.shape[0], counts.shape[1])

    def release_rows(self):
        """free the matrix stage_rows left on the device (the k-mer test was its only reader)"""
        old = getattr(self, "_staged_rows", None)
        if old:
            self.dev_free(old)
            self._staged_rows = None

    # -------------------------------------------------------------- profiling / bench support
    def prof_enable(self, on=True):
        self._ck(self.L.sp_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self._ck(self.L.sp_prof_reset(self.h))

    def prof_report(self):
        buf = C.create_string_buffer(1 << 16)
        self._ck(self.L.sp_prof_report(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._ck(self.L.sp_dev_alloc(self.h, int(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        self._ck(self.L.sp_dev_free(self.h, C.c_void_p(ptr)))

    def dev_to_host(self, ptr, nbytes, offset=0):
        out = np.empty(nbytes, np.uint8)
        self._ck(self.L.sp_dev_copy_to_host(self.h, _p(out), C.c_void_p(ptr + offset), int(nbytes)))
        return out

    def host_register(self, h_ptr, nbytes):
        self._ck(self.L.sp_host_register(self.h, C.c_void_p(int(h_ptr)), int(nbytes)))

    def host_unregister(self, h_ptr):
        self._ck(self.L.sp_host_unregister(self.h, C.c_void_p(int(h_ptr))))

    def dev_to_host_ptr(self, h_ptr, d_ptr, nbytes):
        """device -> host copy into memory given by ADDRESS (e.g. a registered shared-memory segment)"""
        if nbytes:
            self._ck(self.L.sp_dev_copy_to_host(self.h, C.c_void_p(int(h_ptr)), C.c_void_p(int(d_ptr)), int(nbytes)))

    def host_to_dev(self, ptr, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        self._ck(self.L.sp_dev_copy_from_host(self.h, C.c_void_p(ptr + offset), _p(arr), int(arr.nbytes)))

    def synth_chrom_range(self, d_ptr, length, start, n, seed, set_id, sg_id, n_sg, chrom_id, exchange=0):
        self._ck(self.L.sp_synth_chrom_range(self.h, C.c_void_p(d_ptr), int(length), int(start), int(n), int(seed),
                                             int(set_id), int(sg_id), int(n_sg), int(chrom_id), int(exchange)))

    def synth_chrom(self, d_ptr, length, seed, set_id, sg_id, n_sg, chrom_id, exchange=0):
        self._ck(self.L.sp_synth_chrom(self.h, C.c_void_p(d_ptr), int(length), int(seed), int(set_id),
                                       int(sg_id), int(n_sg), int(chrom_id), int(exchange)))
