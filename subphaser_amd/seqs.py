"""Genome ingest and k-mer -> bin mapping.

Host-side mirror of the reference's subphaser/Seqs.py for the hot path:
  split_genomes  (Seqs.py:27-71)    select/rename target chromosomes
  map_kmer3      (Seqs.py:74-119)   subgenome-specific k-mers -> 10-kb bin counts
The per-position work (chunk_chromfiles / map_kmer_each4 / _get_kmer,
Seqs.py:121-153, 209-244) runs in the HIP kernel k5_map.
"""
import copy
import gzip
import os
import sys
from collections import OrderedDict

import numpy as np

from . import _native
from . import kmer as kmerlib
from .runtime import get_context, logger

_DELNL = bytes(range(256))


def _open_bytes(path):
    with open(path, "rb") as fh:
        magic = fh.read(2)
    if magic == b"\x1f\x8b":
        return gzip.open(path, "rb")
    return open(path, "rb")


def _strip_newlines(body):
    """Sequence bytes of a FASTA record body (uint8 array view) without line breaks.
    Fast path: fixed-width lines (every (w+1)-th byte is '\n') -> one strided copy."""
    n = body.size
    if n == 0:
        return body
    first = int(np.argmax(body == 10)) if (body[: min(n, 1 << 16)] == 10).any() else -1
    if first > 0:
        w = first
        rows = n // (w + 1)
        if rows > 0:
            grid = body[: rows * (w + 1)].reshape(rows, w + 1)
            if (grid[:, w] == 10).all():
                tail = body[rows * (w + 1):]
                head = grid[:, :w]
                # fixed width holds for the whole prefix; check the data columns carry no stray breaks
                if tail.size <= w + 1 and not (tail[:-1] == 10).any() if tail.size else True:
                    t = tail[tail != 10] if tail.size else tail
                    if not ((head <= 32).any() or (t <= 32).any()):      # no stray break/blank inside the lines
                        return np.concatenate([head.reshape(-1), t]) if t.size else np.ascontiguousarray(head).reshape(-1)
    keep = (body != 10) & (body != 13) & (body != 32) & (body != 9)
    return body[keep]


def read_gz(path):
    """Decompressed image of a gzip file as a uint8 array.  BGZF files (bgzip: a series of <= 64-KiB gzip members
    whose compressed size sits in a 'BC' extra field) are inflated block by block on a thread pool -- zlib releases
    the GIL -- straight into the output array; plain gzip is one deflate stream and stays one thread."""
    import zlib
    raw = np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.empty(0, np.uint8)
    n = int(raw.size)

    def bgzf_size(o):      # total size of the BGZF block at offset o, or 0
        if o + 18 > n or raw[o] != 31 or raw[o + 1] != 139 or raw[o + 2] != 8 or not (raw[o + 3] & 4):
            return 0
        xlen = int(raw[o + 10]) | (int(raw[o + 11]) << 8)
        p, e = o + 12, o + 12 + xlen
        while p + 4 <= e and e <= n:
            slen = int(raw[p + 2]) | (int(raw[p + 3]) << 8)
            if raw[p] == 66 and raw[p + 1] == 67 and slen == 2:
                return (int(raw[p + 4]) | (int(raw[p + 5]) << 8)) + 1
            p += 4 + slen
        return 0

    blocks, o = [], 0
    while o < n:
        b = bgzf_size(o)
        if b < 26 or o + b > n:
            blocks = None
            break
        blocks.append((o, b))
        o += b
    if not blocks:
        with gzip.open(path, "rb") as fh:
            return np.frombuffer(fh.read(), np.uint8)
    view = memoryview(raw)
    isize = np.array([int.from_bytes(view[a + b - 4:a + b], "little") for a, b in blocks], np.int64)
    off = np.concatenate(([0], np.cumsum(isize)))
    out = np.empty(int(off[-1]), np.uint8)
    from concurrent.futures import ThreadPoolExecutor

    def inflate(span):
        for j in range(*span):
            a, b = blocks[j]
            xlen = int(raw[a + 10]) | (int(raw[a + 11]) << 8)
            chunk = zlib.decompress(view[a + 12 + xlen:a + b - 8], wbits=-15)
            if len(chunk) != isize[j]:
                raise ValueError("corrupt BGZF block at offset {} of {}".format(a, path))
            out[off[j]:off[j + 1]] = np.frombuffer(chunk, np.uint8)
    step = 256
    with ThreadPoolExecutor(max_workers=min(32, len(os.sched_getaffinity(0)))) as pool:
        list(pool.map(inflate, [(j, min(j + step, len(blocks))) for j in range(0, len(blocks), step)]))
    return out


def _native_scanner():
    """The library's host-side scanner (sp_fasta.hip) unless SP_FASTA_NUMPY is set (the numpy twin below)."""
    return not os.environ.get("SP_FASTA_NUMPY")


def read_fasta(path, as_array=False):
    """Yield (id, sequence without line breaks) for every record of a (gz) FASTA file.
    as_array=True yields uint8 numpy arrays (no extra copy for multi-GB genomes), else bytes."""
    with open(path, "rb") as fh:
        magic = fh.read(2)
    if magic == b"\x1f\x8b":
        data = read_gz(path)
    else:
        if os.path.getsize(path) == 0:
            return
        data = np.memmap(path, dtype=np.uint8, mode="r")
    if data.size == 0:
        return
    if _native_scanner():
        ids, cat, off = _native.fasta_scan(data)
        for i, rid in enumerate(ids):
            seq = cat[off[i]:off[i + 1]]
            yield rid, (seq if as_array else seq.tobytes())
        return
    # numpy twin of the scanner (kept as its cross-check, tests/test_abi_and_host.py)
    # record starts: '>' at the beginning of a line (numpy passes run in threads: they release the GIL)
    from concurrent.futures import ThreadPoolExecutor
    n = int(data.size)
    step = 1 << 26
    workers = max(1, min(16, len(os.sched_getaffinity(0))))

    def find(lo):
        hits = np.flatnonzero(data[lo:lo + step] == 62) + lo
        return [int(i) for i in hits.tolist() if i == 0 or data[i - 1] == 10]

    def one(span):
        a, b = span
        rec = data[a:b]
        nl = -1
        for lo in range(0, rec.size, 1 << 16):       # header line: the first line break
            w = np.flatnonzero(rec[lo:lo + (1 << 16)] == 10)
            if w.size:
                nl = lo + int(w[0])
                break
        if nl < 0:
            nl = rec.size
        header = bytes(rec[1:nl]).decode().split()
        return (header[0] if header else ""), _strip_newlines(rec[nl + 1:])

    with ThreadPoolExecutor(max_workers=workers) as pool:
        starts = [i for part in pool.map(find, range(0, n, step)) for i in part]
        spans = list(zip(starts, starts[1:] + [n]))
        for rid, seq in pool.map(one, spans):
            yield rid, (seq if as_array else seq.tobytes())


_BULK_STEP = 1 << 26     # bytes per worker span of read_fasta_bulk


def read_fasta_bulk(path):
    """All records of a (gz) FASTA at once, for files with millions of short records (feature sets):
    returns (ids, cat, off) -- ids list of str, cat uint8 array of all sequences back to back without
    line breaks, off int64 [n + 1].  Pure numpy passes over the file, no per-base Python."""
    with open(path, "rb") as fh:
        magic = fh.read(2)
    if magic == b"\x1f\x8b":
        data = read_gz(path)
    elif os.path.getsize(path) == 0:
        data = np.empty(0, np.uint8)
    else:
        data = np.memmap(path, dtype=np.uint8, mode="r")
    n = int(data.size)
    empty = ([], np.empty(0, np.uint8), np.zeros(1, np.int64))
    if n == 0:
        return empty
    if _native_scanner():
        return _native.fasta_scan(data)
    from concurrent.futures import ThreadPoolExecutor
    step = _BULK_STEP
    spans = [(lo, min(lo + step, n)) for lo in range(0, n, step)]
    pool = ThreadPoolExecutor(max_workers=min(16, len(os.sched_getaffinity(0)), len(spans)))

    def scan(span):      # positions of '>' at line starts and of line breaks (numpy releases the GIL)
        lo, hi = span
        d = data[lo:hi]
        gt = np.flatnonzero(d == 62) + lo
        if gt.size:
            prev = np.where(gt > 0, data[np.maximum(gt - 1, 0)], 10)
            gt = gt[prev == 10]
        return gt, np.flatnonzero(d == 10) + lo
    parts = list(pool.map(scan, spans))
    starts = np.concatenate([p[0] for p in parts])
    nl = np.concatenate([p[1] for p in parts])
    del parts
    if starts.size == 0:
        pool.shutdown()
        return empty
    j = np.searchsorted(nl, starts)
    hdr_end = np.full(starts.size, n, np.int64)
    ok = j < nl.size
    hdr_end[ok] = nl[j[ok]]
    lens = hdr_end - starts
    tot = int(lens.sum())
    hdr_idx = np.repeat(starts - np.concatenate(([0], np.cumsum(lens)[:-1])), lens) + np.arange(tot)
    first = int(starts[0])

    def squeeze(span):   # sequence bytes of one span: no blanks / line breaks, no header bytes
        lo, hi = span
        keep = data[lo:hi] > 32
        a_, b_ = np.searchsorted(hdr_idx, [lo, hi])
        keep[hdr_idx[a_:b_] - lo] = False
        if lo < first:
            keep[:min(hi, first) - lo] = False
        # bytes kept before every record start that falls into this span
        ra, rb = np.searchsorted(starts, [lo, hi])
        cs = np.add.reduceat(keep.view(np.uint8), np.concatenate(([0], starts[ra:rb] - lo)), dtype=np.int64) \
            if rb > ra else np.array([int(keep.sum())], np.int64)
        if rb > ra and starts[ra] == lo:      # reduceat with a repeated index returns the element itself
            cs[0] = 0
        return data[lo:hi][keep], cs
    outs = list(pool.map(squeeze, spans))
    pool.shutdown()
    cat = np.concatenate([o[0] for o in outs])
    # cs pieces: [tail of the record open at the span start, record 1, record 2, ...] -> per-record totals
    rec_len = np.zeros(starts.size, np.int64)
    r = -1
    for (lo, hi), (_, cs) in zip(spans, outs):
        ra, rb = np.searchsorted(starts, [lo, hi])
        if r >= 0:
            rec_len[r] += cs[0]
        if rb > ra:
            rec_len[ra:rb] += cs[1:]
            r = rb - 1
    off = np.zeros(starts.size + 1, np.int64)
    off[1:] = np.cumsum(rec_len)
    assert off[-1] == cat.size
    ids = []
    for a_, b_ in zip(starts.tolist(), hdr_end.tolist()):
        h = bytes(data[a_ + 1:b_]).split()
        ids.append(h[0].decode() if h else "")
    return ids, cat, off


def write_fasta(path, rid, seq, width=60):
    seq = np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else np.asarray(seq, np.uint8)
    with open(path, "wb") as f:
        f.write(b">" + rid.encode() + b"\n")
        n = seq.size
        if n:
            full = (n // width) * width
            if full:
                b = np.empty((full // width, width + 1), np.uint8)
                b[:, :width] = seq[:full].reshape(-1, width)
                b[:, width] = 10
                f.write(b.data)
            if n > full:
                f.write(seq[full:].tobytes() + b"\n")


class ChromRecord:
    """In-memory twin of one tmp/chromosomes/<id>.fasta file."""
    __slots__ = ("rid", "seq", "index")

    def __init__(self, rid, seq):
        self.rid, self.seq, self.index = rid, seq, {}   # index: id(ctx) -> chromosome slot on that GPU


_REG = {}   # chromfile path -> ChromRecord


class PendingWrites:
    """Per-chromosome FASTA copies still being written by the writer threads of split_genomes(defer=True)."""

    def __init__(self, pool, futures):
        self.pool, self.futures = pool, futures

    def wait(self):
        try:
            for fut in self.futures:
                fut.result()        # re-raises a writer's exception
        finally:
            self.futures = []
            if self.pool is not None:
                self.pool.shutdown()
                self.pool = None


def split_genomes(genomes, prefixes, targets, outdir, d_targets=None, sep="|", write_files=True, defer=False):
    """Select target chromosomes, apply `new|old` renaming and label prefixes.
    Returns (chromfiles, labels, d_targets2, d_size) like the reference.  The
    sequences are also kept in memory so the counting step need not re-read them.
    defer=True: a fifth value, a PendingWrites -- the per-chromosome files (which nothing in modules 1-2 reads
    back) are still being written when the call returns; the caller waits before it records the checkpoint."""
    d_targets2 = OrderedDict()
    if not d_targets:
        d_targets = OrderedDict()
        for t in targets:
            tmp = t.split(sep, 1)
            d_targets[tmp[-1]] = tmp[0]
            d_targets2[t] = tmp[0]
    elif set(targets) - set(d_targets):
        for t in set(targets) - set(d_targets):
            tmp = t.split(sep, 1)
            d_targets[tmp[-1]] = tmp[0]
            d_targets2[t] = tmp[0]
    else:
        d_targets2 = copy.deepcopy(d_targets)
    outfas, labels, d_size, got = [], [], {}, set()
    from concurrent.futures import ThreadPoolExecutor
    writer = ThreadPoolExecutor(max_workers=4) if write_files else None   # file writes overlap the parsing
    pending, last_write = [], {}
    for genome, prefix in zip(genomes, prefixes):
        for old_id, seq in read_fasta(genome, as_array=True):
            new_id = "{}{}".format(prefix, old_id)
            if new_id in d_targets:
                rid = new_id
            elif old_id in d_targets:
                rid = old_id
            else:
                continue
            got.add(rid)
            rid = d_targets[rid]
            outfa = "{}{}.fasta".format(outdir, rid)
            if write_files:
                # two records that end up under one id (same name in two genomes, no label prefix): the later one
                # wins, like the reference's sequential loop -- never two writers on one path at a time
                prev = last_write.get(outfa)
                if prev is not None:
                    prev.result()
                last_write[outfa] = writer.submit(write_fasta, outfa, rid, seq)
                pending.append(last_write[outfa])
            _REG[outfa] = ChromRecord(rid, seq)
            outfas.append(outfa)
            labels.append(rid)
            d_size[rid] = len(seq)
    waiting = PendingWrites(writer, pending)
    if not defer:
        waiting.wait()
    missing = set(d_targets) - got
    if missing:
        logger.error("Chromosomes {} are not found in sequences files".format(missing))
    if defer:
        return outfas, labels, d_targets2, d_size, waiting
    return outfas, labels, d_targets2, d_size


def load_chromfile(chromfile):
    """ChromRecord of a per-chromosome FASTA (memory twin first, then disk)."""
    rec = _REG.get(chromfile)
    if rec is None:
        recs = list(read_fasta(chromfile))
        if not recs:
            raise ValueError("no FASTA record in {}".format(chromfile))
        # one record per file (Seqs.py:62-64); several records are joined with an N so no k-mer spans them
        rec = ChromRecord(recs[0][0], recs[0][1] if len(recs) == 1 else b"N".join(sq for _, sq in recs))
        _REG[chromfile] = rec
    return rec


class KmerLabels:
    """Array form of the reference's d_kmers dict (k-mer and its reverse
    complement -> subgenome name, Cluster.py:174-175): canonical keys + SG index."""

    def __init__(self, keys, sg_idx, sg_names, k):
        self.keys = np.ascontiguousarray(keys, np.uint64)
        self.sg_idx = np.ascontiguousarray(sg_idx, np.uint8)
        self.sg_names = list(sg_names)
        self.k = int(k)
        # the device copy below is keyed on the identity of these two arrays: they are frozen so that nobody edits them
        # in place behind it (assigning new arrays to .keys / .sg_idx is seen and makes a new copy) -- advisor r04
        for a in (self.keys, self.sg_idx):
            try:
                a.setflags(write=False)
            except ValueError:
                pass

    def __len__(self):           # the reference's dict holds both orientations
        return 2 * len(self.keys)

    # The labelled rows were tested on the device (Cluster.output_kmers -> sp_kmer_ttest); `on_device(ctx)` keeps a
    # device copy of (keys, sg_idx) for the context, made on first use, so that every later map stage hands the set
    # over with sp_labels_set_device instead of 9 bytes per k-mer of pageable host memory (2.2 of the 2.9 ms
    # `labels_set` cost per wheat-like pass).  The copy dies with the object or the context.
    _dev = None

    def on_device(self, ctx):
        stamp = (id(self.keys), id(self.sg_idx), len(self.keys))
        if self._dev is None or self._dev[0] is not ctx or not ctx.h or self._dev[3] != stamp:
            self.release_device()
            n = len(self.keys)
            d_keys = ctx.dev_alloc(max(n, 1) * 8)
            d_sg = ctx.dev_alloc(max(n, 1))
            if n:
                ctx.host_to_dev(d_keys, self.keys)
                ctx.host_to_dev(d_sg, self.sg_idx)
            self._dev = (ctx, d_keys, d_sg, stamp)
            # the context frees the copy if it is closed first (a closed context has no handle to free with)
            reg = getattr(ctx, "_label_copies", None)
            if reg is None:
                reg = ctx._label_copies = {}
            reg[id(self)] = (d_keys, d_sg)
        return self._dev[1], self._dev[2]

    def release_device(self):
        if self._dev is not None:
            ctx, d_keys, d_sg = self._dev[:3]
            self._dev = None
            reg = getattr(ctx, "_label_copies", None)
            if reg is not None and reg.pop(id(self), None) is not None and ctx.h:
                ctx.dev_free(d_keys)
                ctx.dev_free(d_sg)

    def __del__(self):
        try:
            self.release_device()
        except Exception:
            pass

    def values(self):
        for i in self.sg_idx:
            yield self.sg_names[i]
            yield self.sg_names[i]

    @classmethod
    def from_dict(cls, d_kmers, sg_names, k=None):
        kmers = list(d_kmers.keys())
        if k is None:
            k = len(kmers[0]) if kmers else 0
        keys = kmerlib.encode_many(kmers) if kmers else np.empty(0, np.uint64)
        canon = kmerlib.canonical(keys, k) if kmers else keys
        name_idx = {n: i for i, n in enumerate(sg_names)}
        sg = np.array([name_idx[d_kmers[x]] for x in kmers], np.uint8)
        canon, first = np.unique(canon, return_index=True)
        return cls(canon, sg[first], sg_names, k)


def _as_labels(d_kmers, sg_names, k):
    if isinstance(d_kmers, KmerLabels):
        return d_kmers
    return KmerLabels.from_dict(d_kmers, sg_names, k)


def bin_lines(rid, length, slot_counts, bin_size, chunk_size, k):
    """Reference-format lines for one chromosome from slot counts (see sp_map_bins):
    one line per non-empty (bin, chunk) slot, `end` clipped to the chunk end
    (Seqs.py:228-236).  Returns (starts, ends, counts) arrays of the emitted lines."""
    nz = np.flatnonzero(slot_counts.any(axis=1))
    if nz.size == 0:
        return nz, nz, slot_counts[:0]
    # invert slot -> (bin, chunk): slot = bin + chunk, chunk j owns starts >= j*W-(k-1)
    if chunk_size:
        nch = (length + (k - 1)) // chunk_size + 1
        j = np.arange(1, nch + 1, dtype=np.int64)
        first_start = j * chunk_size - (k - 1)                 # first start owned by chunk j
        first_slot = first_start // bin_size + j               # its slot
        chunk = np.searchsorted(first_slot, nz, side="right")  # number of chunks begun at/before slot
        bins = nz - chunk
        chunk_end = np.minimum((chunk + 1) * chunk_size, length)
    else:
        bins = nz
        chunk_end = np.full(nz.size, length, np.int64)
    starts = bins * bin_size
    ends = np.minimum(starts + bin_size, chunk_end)
    return starts, ends, slot_counts[nz]


def map_kmer3(chromfiles, d_kmers, fout=sys.stdout, k=None, window_size=10e6, bin_size=10000, sg_names=[],
              ncpu="autodetect", method="map", log=True, chunk=True, chunksize=None, ctx=None, collect=None):
    """Same arguments as the reference (ncpu/method/chunksize are accepted and ignored:
    the GPU replaces the process pool).  Writes `#chrom start end SG...` lines to fout.
    collect (chunk=False only): a list that receives, per feature file, the written lines as arrays
    (ids list, starts int64[n], counts int64[n, S]) so that a caller need not parse the text back."""
    ctx = ctx or get_context()
    labels = _as_labels(d_kmers, sg_names, k)
    if k is None:
        k = labels.k
    sg_names = list(sg_names) if sg_names else labels.sg_names
    window_size, bin_size = int(window_size), int(bin_size)
    if ctx.k != k:
        raise ValueError("map_kmer3: context was counted with k={} but k={} requested".format(ctx.k, k))
    ctx.labels_set(labels.keys, labels.sg_idx, len(sg_names))
    fout.write("\t".join(["#chrom", "start", "end"] + sg_names) + "\n")
    n_seq, mapped_seqs, mapped_num = 0, 0, 0
    if chunk:
        for ci, chromfile in enumerate(chromfiles):
            rec = load_chromfile(chromfile)
            seq, rid = rec.seq, rec.rid
            idx = rec.index.get(id(ctx))
            if idx is None:
                raise ValueError("chromosome file {} is not resident on the GPU; run run_jellyfish_dumps "
                                 "first (it uploads the genome)".format(chromfile))
            logger.info("Chunking chromsome {}: {:,} bp".format(rid, len(seq)))
            slots, c = ctx.map_bins(idx, bin_size, window_size)
            starts, ends, counts = bin_lines(rid, len(seq), slots, bin_size, window_size, k)
            _write_lines(fout, rid, starts, ends, counts)
            nchunks = max(1, -(-len(seq) // window_size))
            n_seq += nchunks
            mapped_num += c
            mapped_seqs += nchunks if c else 0
            if log:
                logger.info("Mapped {} kmers to chromsome {}".format(c, rid))
    else:
        from ._native import write_chunks
        for featfile in chromfiles:
            ids, cat, off = read_fasta_bulk(featfile)
            lens = np.diff(off)
            counts = ctx.map_features_cat(cat, off)
            big = {}
            for i in np.flatnonzero(lens > bin_size).tolist():   # rare: a feature longer than one bin
                ci = _map_long_feature(ctx, cat[off[i]:off[i + 1]], bin_size, k)
                nzb = np.flatnonzero(ci.any(axis=1))
                big[i] = (nzb * bin_size, np.minimum(nzb * bin_size + bin_size, int(lens[i])), ci[nzb])
                counts[i] = ci.sum(axis=0)
            tot = counts.sum(axis=1)
            n_seq += len(ids)
            mapped_num += int(tot.sum())
            mapped_seqs += int((tot > 0).sum())
            sel = np.flatnonzero(tot > 0)
            ends = np.minimum(bin_size, lens)

            def fmt(lo, hi, sel=sel, ids=ids, ends=ends, counts=counts, big=big):
                out = []
                for i in sel[lo:hi].tolist():
                    if i in big:
                        st, en, cc = big[i]
                        for a_, b_, row in zip(st.tolist(), en.tolist(), cc.tolist()):
                            out.append("%s\t%d\t%d\t%s\n" % (ids[i], a_, b_, "\t".join(map(str, row))))
                    else:
                        out.append("%s\t0\t%d\t%s\n" % (ids[i], ends[i], "\t".join(map(str, counts[i].tolist()))))
                return "".join(out)
            if collect is not None:
                if not big:
                    collect.append(([ids[i] for i in sel.tolist()], np.zeros(sel.size, np.int64),
                                    counts[sel].astype(np.int64)))
                else:
                    r_ids, r_st, r_cc = [], [], []
                    for i in sel.tolist():
                        if i in big:
                            st, _, cc = big[i]
                            r_ids += [ids[i]] * len(st)
                            r_st += st.tolist()
                            r_cc += cc.tolist()
                        else:
                            r_ids.append(ids[i])
                            r_st.append(0)
                            r_cc.append(counts[i].tolist())
                    collect.append((r_ids, np.asarray(r_st, np.int64),
                                    np.asarray(r_cc, np.int64).reshape(len(r_ids), counts.shape[1])))
            done = False
            if not big and sel.size:     # the common case: one line per feature, rows formatted by the library's threads
                sid = [ids[i] for i in sel.tolist()]
                blob, boff = _native.str_blob(sid)
                done = _native.text_table(fout, sel.size, [("str", blob, boff), ("i64", np.zeros(sel.size, np.int64), "\t"),
                                                           ("i64", ends[sel], "\t"), ("i64", counts[sel], "\t")])
            if not done:
                write_chunks(fout, len(sel), fmt)
    logger.info("Processed {} sequences".format(n_seq))
    total = len(labels.keys)
    if n_seq and total:
        hit = ctx.labels_hit()
        logger.info("{} ({:.2%}) sequences contain subgenome-specific kmers".format(mapped_seqs, mapped_seqs / n_seq))
        logger.info("{:.2%} of {} subgenome-specific kmers are mapped".format(hit / total, total))
    else:
        logger.warning("None sequences, please check.")
    return mapped_num


def is_bed(path):
    """a feature set given as BED intervals (chrom, start, end[, name ...]) rather than as a FASTA of sequences"""
    opener = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    with opener(path, "rt") as fh:
        for line in fh:
            if not line.strip() or line.startswith(("#", "track", "browser")):
                continue
            if line.startswith(">"):
                return False
            t = line.split()
            return len(t) >= 3 and t[1].isdigit() and t[2].isdigit()
    return False


def read_bed(path):
    """(names list, code int64[n], start int64[n], end int64[n]) of a (gz) BED file -- 0-based, half-open; code[i]
    indexes names.  Millions of lines: parsed by pandas' C reader when it is importable."""
    try:
        import pandas as pd
        df = pd.read_csv(path, sep=r"\s+", header=None, usecols=[0, 1, 2], comment="#", dtype={0: str, 1: str, 2: str},
                         engine="c", skip_blank_lines=True, names=["c", "s", "e"])
        junk = df["c"].str.startswith(("track", "browser"))
        if junk.any():
            df = df[~junk]
        # (a line with fewer than three fields gives NaN, which is truthy in an object Series: count it as not a digit)
        good = (df["s"].str.isdigit() == True) & (df["e"].str.isdigit() == True)      # noqa: E712 (NaN == True is False)
        if len(df) and not good.all():
            bad = df[~good].iloc[0]
            raise ValueError("{}: not a BED line: {!r}".format(path, "\t".join(str(x) for x in bad.tolist() if x == x)))
        code, names = pd.factorize(df["c"], sort=False)
        return list(names), code.astype(np.int64), df["s"].to_numpy().astype(np.int64), df["e"].to_numpy().astype(np.int64)
    except ImportError:
        pass
    opener = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    names, where, code, st, en = [], {}, [], [], []
    with opener(path, "rt") as fh:
        for ln, line in enumerate(fh, 1):
            if not line.strip() or line.startswith(("#", "track", "browser")):
                continue
            t = line.split()
            if len(t) < 3 or not (t[1].isdigit() and t[2].isdigit()):
                raise ValueError("{}:{}: not a BED line: {!r}".format(path, ln, line.rstrip()))
            if t[0] not in where:
                where[t[0]] = len(names)
                names.append(t[0])
            code.append(where[t[0]])
            st.append(int(t[1]))
            en.append(int(t[2]))
    return names, np.asarray(code, np.int64), np.asarray(st, np.int64), np.asarray(en, np.int64)


class IntervalRows:
    """The rows of an interval feature set as arrays: row i is `names[code[i]]:start[i]-end[i]`.  Stands in for the
    list of id strings where millions of ids would only be formatted to be written out again."""

    def __init__(self, names, code, start, end):
        self.names, self.code = list(names), np.asarray(code, np.int64)
        self.start, self.end = np.asarray(start, np.int64), np.asarray(end, np.int64)

    def __len__(self):
        return int(self.code.size)

    def ids(self):
        return ["%s:%d-%d" % (self.names[c], a, b) for c, a, b in zip(self.code.tolist(), self.start.tolist(), self.end.tolist())]

    def take(self, sel):
        return IntervalRows(self.names, self.code[sel], self.start[sel], self.end[sel])

    def column(self):
        """the id column for _native.text_table"""
        return ("ival", np.stack([self.code, self.start, self.end], axis=1), self.names)

    def merged(self, counts):
        """(rows, counts) with the records that name the same interval added up, in order of first appearance -- what
        Circos.stack_matrix (Circos.py:709-742) makes of FASTA lines that carry the same id: a BED file that lists an
        interval twice must give the same `.custom.enrich` row as the FASTA of the same sub-sequences."""
        if len(self) < 2:
            return self, counts
        key = np.stack([self.code, self.start, self.end], axis=1)
        _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
        if first.size == len(self):
            return self, counts
        order = np.argsort(first, kind="stable")            # groups in order of first appearance
        rank = np.empty(order.size, np.int64)
        rank[order] = np.arange(order.size)
        out = np.zeros((order.size, counts.shape[1]), counts.dtype)
        np.add.at(out, rank[inv.reshape(-1)], counts)
        return self.take(np.sort(first)), out

    @staticmethod
    def concat(parts):
        names, where, codes = [], {}, []
        for p_ in parts:
            remap = np.empty(len(p_.names), np.int64)
            for j, nm in enumerate(p_.names):
                if nm not in where:
                    where[nm] = len(names)
                    names.append(nm)
                remap[j] = where[nm]
            codes.append(remap[p_.code] if len(p_) else p_.code)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)
        return IntervalRows(names, cat(codes), cat([p_.start for p_ in parts]), cat([p_.end for p_ in parts]))


def map_intervals(bedfiles, d_kmers, chrom_index, fout=sys.stdout, k=None, bin_size=10000, sg_names=[], ctx=None,
                  collect=None, aliases=None):
    """`-custom_features` given as BED: the intervals are reduced over the genome that is already resident on the GPU
    (sp_map_intervals) instead of being uploaded as sequence (map_kmer3(chunk=False) on a FASTA of the same
    sub-sequences writes the same lines, with ids `chrom:start-end`; __main__.py:509-517, Seqs.py:228-244).
    chrom_index: {chromosome label: index in the context's genome}; aliases: {other name: label} (e.g. the ids
    before renaming).  Intervals on sequences that are not target chromosomes are skipped, like the records of a
    feature FASTA that carry no subgenome-specific k-mer.
    collect: a list that receives, per BED file, (IntervalRows, counts int64 [n, S]) of the features that carry a
    subgenome-specific k-mer (one row per feature: what stack_matrix makes of the lines)."""
    ctx = ctx or get_context()
    labels = _as_labels(d_kmers, sg_names, k)
    if k is None:
        k = labels.k
    if ctx.k != k:
        raise ValueError("map_intervals: context was counted with k={} but k={} requested".format(ctx.k, k))
    sg_names = list(sg_names) if sg_names else labels.sg_names
    bin_size = int(bin_size)
    ctx.labels_set(labels.keys, labels.sg_idx, len(sg_names))
    fout.write("\t".join(["#chrom", "start", "end"] + sg_names) + "\n")
    aliases = aliases or {}
    chrom_len = [int(ctx.genome_len(i)) for i in range(ctx.n_chrom)]
    n_seq = mapped_seqs = mapped_num = skipped = 0
    for bed in bedfiles:
        names, code, st, en = read_bed(bed)
        names = [aliases.get(c, c) for c in names]           # chromosome labels from here on (ids, subgenome look-ups)
        gidx = np.array([chrom_index.get(c, -1) for c in names], np.int64)   # per distinct name
        idx = gidx[code] if code.size else code
        keep = np.flatnonzero(idx >= 0)
        skipped += int(code.size - keep.size)
        code, idx, st, en = code[keep], idx[keep], st[keep], en[keep]
        # the library rejects the whole batch for one bad interval and can only name its index after splitting: name
        # the BED record here (an annotation of another assembly version is the usual cause)
        wrong = np.flatnonzero((st > en) | (en > np.asarray(chrom_len, np.int64)[idx])) if code.size else keep[:0]
        if wrong.size:
            w = int(wrong[0])
            raise ValueError("{}: record {} ({}:{}-{}) does not lie on its chromosome (length {}); {} such record(s)".format(
                bed, int(keep[w]) + 1, names[int(code[w])], int(st[w]), int(en[w]), chrom_len[int(idx[w])], wrong.size))
        lens = en - st
        n = int(code.size)
        if n and bool((lens > bin_size).any()):
            # a feature longer than one bin is reported per bin, bin j owning the k-mer starts [j * bin, (j + 1) * bin)
            nb = np.maximum(1, -(-lens // bin_size))
            piece_of = np.repeat(np.arange(n), nb)
            j = np.arange(piece_of.size) - np.repeat(np.cumsum(nb) - nb, nb)
            p_st = st[piece_of] + j * bin_size
            p_en = np.maximum(np.minimum(p_st + bin_size + (k - 1), en[piece_of]), p_st)
            pc = ctx.map_intervals(idx[piece_of], p_st, p_en)
            counts = np.zeros((n, len(sg_names)), np.int64)
            np.add.at(counts, piece_of, pc)
            many = nb > 1
        else:
            piece_of, j = np.arange(n), np.zeros(n, np.int64)
            pc = counts = ctx.map_intervals(idx, st, en) if n else np.zeros((0, len(sg_names)), np.int64)
            many = np.zeros(n, bool)
        tot = counts.sum(axis=1)
        n_seq += n
        mapped_num += int(tot.sum())
        mapped_seqs += int((tot > 0).sum())
        feats = IntervalRows(names, code, st, en)
        rows = np.flatnonzero((tot[piece_of] > 0) & (~many[piece_of] | pc.any(axis=1)))      # the lines, in file order
        r_feat = piece_of[rows]
        r_st = (j[rows] * bin_size).astype(np.int64)
        r_en = np.minimum(r_st + bin_size, lens[r_feat])
        r_cc = pc[rows]
        if collect is not None:
            sel = np.flatnonzero(tot > 0)
            collect.append((feats.take(sel), counts[sel]))
        if rows.size and not _native.text_table(fout, rows.size, [feats.take(r_feat).column(), ("i64", r_st, "\t"),
                                                                  ("i64", r_en, "\t"), ("i64", r_cc, "\t")]):
            for rid, a, b, cc in zip(feats.take(r_feat).ids(), r_st.tolist(), r_en.tolist(), r_cc.tolist()):
                fout.write("%s\t%d\t%d\t%s\n" % (rid, a, b, "\t".join(map(str, cc))))
    logger.info("Processed {} intervals".format(n_seq))
    if skipped:
        logger.info("{} intervals lie on sequences that are not target chromosomes: skipped".format(skipped))
    total = len(labels.keys)
    if n_seq and total:
        hit = ctx.labels_hit()
        logger.info("{} ({:.2%}) intervals contain subgenome-specific kmers".format(mapped_seqs, mapped_seqs / n_seq))
        logger.info("{:.2%} of {} subgenome-specific kmers are mapped".format(hit / total, total))
    else:
        logger.warning("None intervals, please check.")
    return mapped_num


def _write_lines(fout, rid, starts, ends, counts):
    if len(starts) == 0:
        return
    cols = [np.asarray(starts).astype(str), np.asarray(ends).astype(str)]
    cols += [counts[:, j].astype(str) for j in range(counts.shape[1])]
    lines = [rid + "\t" + "\t".join(t) for t in zip(*cols)]
    fout.write("\n".join(lines) + "\n")


def _map_long_feature(ctx, seq, bin_size, k):
    """A feature longer than one bin (rare: the reference then emits one line per 10-Mb bin of
    the record, Seqs.py:228-236).  Bin j owns the k-mer starts [j*bin, (j+1)*bin): those are
    exactly the k-mers of the piece seq[j*bin : (j+1)*bin + k-1], so the per-bin counts are the
    whole-piece totals of an ordinary feature batch."""
    nb = -(-len(seq) // bin_size)
    pieces = [seq[j * bin_size:(j + 1) * bin_size + k - 1] for j in range(nb)]
    return np.asarray(ctx.map_features(pieces), np.int64)
