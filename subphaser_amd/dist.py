"""Multi-GPU hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Sharding (DESIGN.md "Multi-GPU"), k <= 15:
  * K0-K2 (pack, count) and K5 (bin map): by GENOME POSITION.  The concatenated genome is cut into N
    contiguous slices of equal size; a cut inside a chromosome lies on a multiple of the 10-Mb chunk size
    and the left piece carries a k-1 base halo -- the reference's own chunking rule (Seqs.py:121-139), so
    every k-mer start is counted and mapped exactly once.  Perfect balance for any chromosome count
    (21 wheat chromosomes on 8 GPUs would otherwise leave 3 chromosomes on the busiest rank).
  * K3 (matrix + differential filter): by dense-table SLOT RANGE.  The one real exchange step of
    the path: every rank sends slice r of each of its byte count tables to rank r
    (`all_to_all_single`, one round per local piece), so rank r holds slots [r*n/N, (r+1)*n/N) of ALL
    pieces; pieces of one chromosome are added up exactly (`sp_table_merge`) and the range is filtered
    locally.  With N-1 direct xGMI links per GPU an all-to-all uses every link at once; a ring
    all-reduce of the same tables would be bound by one link.
  * windows: every rank stacks its pieces into a whole-genome window table on the device, one all-reduce
    (a few hundred KB) adds them up, and every rank then holds all windows and tests them (the Fisher
    stage is 40 us of kernel time: computing it everywhere is cheaper than gathering results).
    For k > 15 (64-bit keys, no dense tables) the same exchange is a KEY-RANGE partition: every
    rank cuts the sorted (key, count) lists of its chromosomes at common splitters (quantiles of one
    list, broadcast) and sends piece r to rank r (`all_to_all_single` with uneven splits); rank r then
    filters its key range of all chromosomes (`sp_sparse_view`).
  * small reductions: `lengths` (all_reduce), surviving rows and window rows (all_gather).

Everything that touches torch is passed in (`dist`, `torch`), so the same code runs on CPU
tensors over gloo in the tests (tests/test_dist_gloo.py) with an oracle-backed context.
"""
import time

import numpy as np

from .config import sets_to_csr
from .hotpath import HotPathResult


def lpt_assign(lengths, n_ranks):
    """Longest-processing-time-first assignment of chromosomes to ranks -> list of index lists."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * n_ranks
    owned = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda j: (load[j], j))
        owned[r].append(i)
        load[r] += lengths[i]
    return [sorted(o) for o in owned]


def plan_pieces(lengths, n_ranks, align):
    """Cut the concatenated genome into n_ranks contiguous slices of (almost) equal size.  A cut inside a
    chromosome is moved to the nearest multiple of `align` (the map stage's chunk size: the slot and window
    numbering of a piece is then the chromosome's numbering plus a constant).  Returns, per rank, a list of
    (chromosome, start, end): the rank owns the k-mer STARTS in [start, end)."""
    total = sum(lengths)
    offs = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
    cuts = [0]
    for r in range(1, n_ranks):
        g = total * r // n_ranks
        c = int(np.searchsorted(offs, g, side="right") - 1)
        c = min(c, len(lengths) - 1)
        p = int(g - offs[c])
        p = int(round(p / align)) * align if align > 0 else p
        p = min(max(p, 0), int(lengths[c]))
        cuts.append(max(int(offs[c]) + p, cuts[-1]))
    cuts.append(total)
    out = []
    for r in range(n_ranks):
        a, b, pieces = cuts[r], cuts[r + 1], []
        for c, n in enumerate(lengths):
            lo, hi = max(a, int(offs[c])), min(b, int(offs[c + 1]))
            if hi > lo:
                pieces.append((c, lo - int(offs[c]), hi - int(offs[c])))
        out.append(pieces)
    return out


class DistHotPath:
    def __init__(self, ctx, gen, dist, torch, k=15, lower_count=3, engine=0, device=None, **kw):
        self.ctx, self.gen, self.dist, self.torch = ctx, gen, dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.k, self.lower_count, self.engine = k, lower_count, engine
        self.labels = gen.labels
        self.lengths_bp = [c["length"] for c in gen.chroms]
        self.C = len(self.labels)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.csr = sets_to_csr(gen.sgs, self.labels)
        self.kw = kw
        self.bin_size = kw.get("bin_size", 10000)
        self.chunk_size = kw.get("chunk_size", 10_000_000)
        self.window_size = kw.get("window_size", 1_000_000)
        t = torch
        self.sparse = k > 15
        if self.sparse:
            # k > 15: whole chromosomes per rank (longest-processing-time assignment)
            self.owned = lpt_assign(self.lengths_bp, self.world)
            self.pieces = [[(c, 0, self.lengths_bp[c]) for c in o] for o in self.owned]
        else:
            self.pieces = plan_pieces(self.lengths_bp, self.world, self.chunk_size if self.chunk_size > 0 else self.bin_size)
            self.owned = [sorted({c for c, _, _ in p}) for p in self.pieces]
        self.my_chroms = self.owned[self.rank]
        self.max_local = max(len(p) for p in self.pieces)
        # what the caller hands to count_and_filter, in this order: bases [start, stop) of chromosome `chrom`
        # (stop = end + k - 1 inside a chromosome: the halo that lets the piece see its last k-mer starts)
        self.local_pieces = [dict(chrom=c, start=a, end=b, stop=min(b + k - 1, self.lengths_bp[c]) if b < self.lengths_bp[c] else b)
                             for c, a, b in self.pieces[self.rank]]
        if not self.sparse:
            self.nslots = ctx.nslots(k)
            # slot range of rank r: [r * chunk, min((r + 1) * chunk, nslots)); 64-slot aligned, the last one may be short
            self.chunk = (self.nslots + 64 * self.world - 1) // (64 * self.world) * 64
            self.nview = max(0, min(self.chunk, self.nslots - self.rank * self.chunk))
            # byte count tables of the local pieces live in ONE torch tensor: the table IS the wire format
            # (one byte per slot + a short overflow list for counts >= 255), so RCCL sends slices of it as they are
            nl = max(1, len(self.local_pieces))
            self.tabs = t.zeros((nl, self.world, self.chunk), dtype=t.uint8, device=self.device)
            self.dummy8 = None
            # receive side: round i, source rank s -> byte slice of rank s's i-th piece; the filter reads them in place
            self.recv8 = self.tabs if self.world == 1 else \
                t.zeros((self.max_local, self.world, self.chunk), dtype=t.uint8, device=self.device)
        else:
            self._sbuf = {}     # growth-only exchange buffers of the key-range path
        self.min_fold = kw.get("min_fold", 2.0)
        self.baseline = kw.get("baseline", 1)
        self.min_freq = kw.get("min_freq", 200)
        self.max_freq = kw.get("max_freq", 1e9)
        self.ratio = kw.get("ratio", 1.0)
        self.max_pval = kw.get("max_pval", 0.05)
        self.shared_rows = kw.get("shared_rows", self.device.type == "cuda")
        self.wall = {}
        self._pin = {}
        self._fence()

    def _fence(self):
        """torch allocates and fills on ITS stream, the library launches on the context's own non-blocking
        stream: wait for torch's pending work before a library call reads or writes a fresh torch tensor."""
        if self.device.type == "cuda":
            self.torch.cuda.current_stream(self.device).synchronize()

    def _t(self, name, t0):
        t1 = time.perf_counter()
        self.wall[name] = self.wall.get(name, 0.0) + (t1 - t0)
        return t1

    # ------------------------------------------------------------------ helpers
    def _ptr(self, tensor):
        return tensor.data_ptr()

    def _all_gather_rows(self, arr, dtype, to_host=True):
        """all_gather of a [m, w] host array with rank-dependent m -> concatenated host array
        (None when to_host is False: the rank took part in the collective but needs no copy)."""
        t, dist = self.torch, self.dist
        arr = np.ascontiguousarray(arr)
        w = arr.shape[1] if arr.ndim == 2 else 1
        m = t.tensor([arr.shape[0]], dtype=t.int64, device=self.device)
        ms = [t.zeros(1, dtype=t.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(ms, m)
        ms = [int(x.item()) for x in ms]
        mmax = max(max(ms), 1)
        pad = t.zeros((mmax, w), dtype=dtype, device=self.device)
        if arr.shape[0]:
            pad[: arr.shape[0]] = t.from_numpy(arr.reshape(arr.shape[0], w)).to(self.device)
        outs = [t.zeros((mmax, w), dtype=dtype, device=self.device) for _ in range(self.world)]
        dist.all_gather(outs, pad)
        if not to_host:
            return None
        return np.concatenate([o[:n].cpu().numpy() for o, n in zip(outs, ms)], axis=0)

    def _all_gather_dev(self, ten, m):
        """all_gather of device tensors [m_r, w] with rank-dependent m_r -> one device tensor."""
        t, dist = self.torch, self.dist
        if self.world == 1:
            return ten
        ms = [t.zeros(1, dtype=t.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(ms, t.tensor([m], dtype=t.int64, device=self.device))
        ms = [int(x.item()) for x in ms]
        mmax = max(max(ms), 1)
        pad = t.zeros((mmax, ten.shape[1]), dtype=ten.dtype, device=self.device)
        pad[:m] = ten
        outs = [t.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(outs, pad)
        return t.cat([o[:n] for o, n in zip(outs, ms)], dim=0)

    def _to_host(self, ten):
        t = self.torch
        if ten.device.type != "cuda":
            return ten.contiguous().numpy()
        key = (tuple(ten.shape), ten.dtype)
        buf = self._pin.get(key)
        if buf is None:
            buf = t.empty(ten.shape, dtype=ten.dtype, pin_memory=True)
            if len(self._pin) > 4:
                self._pin.clear()
            self._pin[key] = buf
        buf.copy_(ten)
        t.cuda.synchronize()
        return buf.numpy()

    # ------------------------------------------------------------------ first half
    def count_and_filter(self, d_pieces, host_rows_on_all_ranks=True):
        """d_pieces: device pointers (or arrays, for a CPU context) of the ASCII bases of `self.local_pieces`,
        in that order.  host_rows_on_all_ranks=False: only rank 0 copies the gathered matrix to the host."""
        if self.sparse:
            return self._count_and_filter_sparse(d_pieces, host_rows_on_all_ranks)
        ctx, t, dist = self.ctx, self.torch, self.dist
        mine = self.local_pieces
        tt = time.perf_counter()
        self._fence()
        ctx.genome_reset(len(mine))
        for li, pc in enumerate(mine):
            ctx.tables_bind(li, self._ptr(self.tabs[li]))
            ctx.genome_add_device(li, d_pieces[li], pc["stop"] - pc["start"])
        if self.world > 1:
            ctx.sync()
            tt = self._t("pack", tt)
        # count piece i and put its byte table on the wire (slot-range slice r -> rank r) while piece i+1 is
        # being counted: the exchange hides behind the counting kernels.  One rank: nothing goes on the wire, so the
        # pieces are counted in one call (packing and the chains of several pieces side by side, as sp_count does)
        works, n_ovf = [], np.zeros(self.max_local, np.int64)
        if self.world == 1 and mine:
            ctx.count_range(self.k, self.lower_count, self.engine, 0, len(mine))
        for i in range(self.max_local):
            if i < len(mine):
                if self.world > 1:
                    ctx.count_range(self.k, self.lower_count, self.engine, i, i + 1)   # synchronises
                n_ovf[i] = ctx.table_overflow(i)
                send = self.tabs[i]
            else:
                if self.dummy8 is None:
                    self.dummy8 = t.zeros((self.world, self.chunk), dtype=t.uint8, device=self.device)
                send = self.dummy8
            if self.world > 1:
                # the exchange must move every slot of the table exactly once: world equal slices that cover it
                assert send.numel() == self.world * self.chunk == self.recv8[i].numel() and self.world * self.chunk >= self.nslots, \
                    "all_to_all_single: %d send / %d recv bytes for %d ranks x %d slots (table: %d slots)" % (
                        send.numel(), self.recv8[i].numel(), self.world, self.chunk, self.nslots)
                works.append(dist.all_to_all_single(self.recv8[i].view(-1), send.reshape(-1), async_op=True))
        tt = self._t("count(+exchange issue)", tt)
        # overflow pairs of every piece to every rank (small: counts >= 255 are rare)
        mine_ovf = t.zeros((max(int(n_ovf.sum()), 1), 2), dtype=t.int32, device=self.device)
        self._fence()
        off = 0
        for i in range(len(mine)):
            if n_ovf[i]:
                ctx.table_overflow(i, mine_ovf.data_ptr() + off * 8, int(n_ovf[i]))
            off += int(n_ovf[i])
        ctx.sync()
        mine_ovf = mine_ovf[:int(n_ovf.sum())]
        novf_t = t.from_numpy(n_ovf).to(self.device)
        if self.world > 1:
            outs = [t.zeros_like(novf_t) for _ in range(self.world)]
            dist.all_gather(outs, novf_t)
            novf_all = np.stack([o.cpu().numpy() for o in outs])      # [rank][its i-th piece]
        else:
            novf_all = n_ovf[None]
        all_ovf = self._all_gather_dev(mine_ovf, int(n_ovf.sum()))
        for w in works:
            w.wait()
        if hasattr(t, "cuda") and self.device.type == "cuda":
            t.cuda.synchronize()
        tt = self._t("exchange wait", tt)
        # one byte slice + overflow list per CHROMOSOME: pieces of a chromosome counted on different ranks add up
        ptrs, optrs, ons = [0] * self.C, [0] * self.C, np.zeros(self.C, np.int64)
        split = np.zeros(self.C, bool)
        self._merged = []      # keeps the merged overflow lists alive until the filter has run
        base, off = self.rank * self.chunk, 0
        for s_, pieces in enumerate(self.pieces):
            for i, (gi, a, b) in enumerate(pieces):
                n = int(novf_all[s_, i])
                src, src_ovf = self._ptr(self.recv8[i, s_]), (all_ovf.data_ptr() + off * 8 if n else 0)
                off += n
                if not ptrs[gi]:
                    ptrs[gi], optrs[gi], ons[gi] = src, src_ovf, n
                    continue
                split[gi] = True
                cap = int(ons[gi]) + n + self.lengths_bp[gi] // 255 + 16
                merged = t.zeros((cap, 2), dtype=t.int32, device=self.device)
                self._fence()
                m = ctx.table_merge(ptrs[gi], optrs[gi], int(ons[gi]), src, src_ovf, n, base, self.nview,
                                    merged.data_ptr(), cap) if self.nview else 0
                self._merged.append(merged)
                optrs[gi], ons[gi] = (merged.data_ptr() if m else 0), m
            off += int(novf_all[s_, len(pieces):].sum())
        # global `lengths` (sum of the dumped counts per chromosome, Jellyfish.py:97,449): the owner of a whole
        # chromosome knows it from counting; a split chromosome's is summed over the slot ranges after merging
        lens = t.zeros(self.C, dtype=t.int64, device=self.device)
        local = ctx.lengths() if mine else np.zeros(0, np.int64)
        lens_h = np.zeros(self.C, np.int64)
        for li, pc in enumerate(mine):
            if not split[pc["chrom"]]:
                lens_h[pc["chrom"]] = local[li]
        for gi in np.flatnonzero(split):
            if self.nview:
                lens_h[gi] = ctx.table_lengths(ptrs[gi], optrs[gi], int(ons[gi]), base, self.nview, self.lower_count)[0]
        lens += t.from_numpy(lens_h).to(self.device)
        dist.all_reduce(lens)
        lengths = lens.cpu().numpy()
        tt = self._t("merge+lengths", tt)
        n_union = n_rows = n_hist = 0
        if self.nview:
            ctx.filter_view(ptrs, base, self.nview, lengths, self.k, self.lower_count, optrs, ons)
            n_union, n_rows, n_hist = ctx.filter(*self.csr, self.min_fold, self.baseline, self.min_freq,
                                                 self.max_freq, self.ratio)
        if self.shared_rows and not host_rows_on_all_ranks and hasattr(ctx, "filter_fetch_async_ptr"):
            # round 5: every rank's rows travel straight from the filter's buffers to ITS row range of the shared
            # page-locked segment on the copy stream, behind the map stage; `result.wait()` = copy done + barrier.
            # (round 6, advisor: the choice depends on nothing rank-specific -- a rank with an empty slot range takes the
            # same path with zero rows, so every rank meets the same collectives in the same order -- and the merged
            # overflow lists k3_emit still reads on the context's own stream stay referenced until wait() has run)
            r = self._rows_shared_async(n_rows if self.nview else 0, n_union, n_hist, lengths, keep=self._merged)
            if self.nview:
                ctx.filter_view(None, 0, 0, None, 0, 0)
            self._merged = []
            self._t("filter+fetch", tt)
            return r
        # surviving rows stay on the device: gathered over xGMI, copied to the host once, where needed
        keys_t = t.empty((max(n_rows, 1),), dtype=t.int64, device=self.device)
        counts_t = t.empty((max(n_rows, 1), self.C), dtype=t.int32, device=self.device)
        self._fence()
        if self.nview:
            ctx.filter_fetch_device(keys_t.data_ptr(), counts_t.data_ptr(), None, n_rows)
            ctx.filter_view(None, 0, 0, None, 0, 0)
        self._merged = []
        tt = self._t("filter+fetch", tt)
        return self._gather_rows(keys_t, counts_t, n_rows, n_union, n_hist, lengths, host_rows_on_all_ranks, tt)

    def _rows_shared_async(self, n_rows, n_union, n_hist, lengths, keep=None):
        t, dist = self.torch, self.dist
        r = HotPathResult()
        r.kmer_lengths = lengths
        stats = t.tensor([n_union, n_rows, n_hist], dtype=t.int64, device=self.device)
        per_rank = [t.zeros(3, dtype=t.int64, device=self.device) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(per_rank, stats)
        else:
            per_rank = [stats]
        per_rank = np.stack([x.cpu().numpy() for x in per_rank])
        r.n_union, r.n_rows, r.n_hist = (int(x) for x in per_rank.sum(axis=0))
        r.freqs = r.tot = None
        ms = per_rank[:, 1]
        M, first = int(ms.sum()), int(ms[:self.rank].sum())
        kbytes, cbytes = 8 * max(M, 1), 4 * self.C * max(M, 1)
        koff = (kbytes + 4095) & ~4095
        self._shm_ensure(koff + cbytes)
        base = self._shm_addr
        if n_rows:
            self.ctx.filter_fetch_async_ptr(base + 8 * first, base + koff + 4 * self.C * first, n_rows)
        buf = self._shm.buf
        r.keys = np.frombuffer(buf, np.uint64, M, 0)
        r.counts = np.frombuffer(buf, np.uint32, M * self.C, koff).reshape(M, self.C)
        ctx, world = self.ctx, self.world

        held = [keep]

        def wait():
            ctx.filter_fetch_wait()
            held[0] = None          # (the emit kernel is done: its overflow lists may go back to the allocator)
            if world > 1:
                dist.barrier()      # every rank's rows are in place

        r.wait = wait
        self.rows_handover = "shared, asynchronous"
        return r

    def _gather_rows(self, keys_t, counts_t, n_rows, n_union, n_hist, lengths, host_rows_on_all_ranks, tt):
        """Surviving rows of every rank's slot / key range -> one matrix (rank order = ascending range).
        On one node the matrix is assembled in a page-locked POSIX shared-memory segment: every rank copies ITS
        rows over ITS PCIe link into its row range, so the M x C matrix becomes host-visible N times faster than
        an all-gather to rank 0 followed by one device->host copy (223 MB for the wheat-like genome)."""
        t, dist = self.torch, self.dist
        r = HotPathResult()
        r.kmer_lengths = lengths
        stats = t.tensor([n_union, n_rows, n_hist], dtype=t.int64, device=self.device)
        per_rank = [t.zeros(3, dtype=t.int64, device=self.device) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(per_rank, stats)
        else:
            per_rank = [stats]
        per_rank = np.stack([x.cpu().numpy() for x in per_rank])
        r.n_union, r.n_rows, r.n_hist = (int(x) for x in per_rank.sum(axis=0))
        r.freqs = None
        r.tot = None      # row sums: counts.sum(axis=1) on demand
        if self.shared_rows and self.world > 1 and not host_rows_on_all_ranks:
            ms = per_rank[:, 1]
            M, first = int(ms.sum()), int(ms[:self.rank].sum())
            kbytes, cbytes = 8 * max(M, 1), 4 * self.C * max(M, 1)
            self._shm_ensure(((kbytes + 4095) & ~4095) + cbytes)
            base = self._shm_addr
            self.ctx.dev_to_host_ptr(base + 8 * first, keys_t.data_ptr(), 8 * n_rows)
            self.ctx.dev_to_host_ptr(base + ((kbytes + 4095) & ~4095) + 4 * self.C * first, counts_t.data_ptr(),
                                     4 * self.C * n_rows)
            dist.barrier()      # every rank's rows are in place
            buf = self._shm.buf
            r.keys = np.frombuffer(buf, np.uint64, M, 0)
            r.counts = np.frombuffer(buf, np.uint32, M * self.C, (kbytes + 4095) & ~4095).reshape(M, self.C)
            tt = self._t("rows to shared host memory", tt)
            self.rows_handover = "shared"
            return r
        to_host = host_rows_on_all_ranks or self.rank == 0
        gk = self._all_gather_dev(keys_t[:n_rows].reshape(-1, 1), n_rows)
        gc = self._all_gather_dev(counts_t[:n_rows], n_rows)
        if to_host:
            r.keys = self._to_host(gk).ravel().view(np.uint64)
            r.counts = self._to_host(gc).view(np.uint32)
        tt = self._t("gather rows", tt)
        self.rows_handover = "gather"
        return r

    # shared, page-locked host segment for the matrix (growth-only; rank 0 creates, the others attach)
    _shm, _shm_addr, _shm_cap = None, 0, 0

    def _shm_ensure(self, nbytes):
        import ctypes
        from multiprocessing import shared_memory, resource_tracker
        dist = self.dist
        if nbytes <= self._shm_cap:
            return
        self._shm_release()
        cap = int(nbytes * 1.25) + (1 << 20)
        name = [None]
        if self.rank == 0:
            self._shm = shared_memory.SharedMemory(create=True, size=cap)
            name[0] = self._shm.name
        dist.broadcast_object_list(name, src=0)
        if self.rank != 0:
            self._shm = shared_memory.SharedMemory(name=name[0])
            try:    # only the creator unlinks (Python < 3.13 would let every attaching process do it at exit)
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        self._shm_addr = ctypes.addressof(ctypes.c_char.from_buffer(self._shm.buf))
        self._shm_cap = cap
        self.ctx.host_register(self._shm_addr, cap)
        dist.barrier()

    def _shm_release(self):
        if self._shm is None:
            return
        try:
            self.ctx.host_unregister(self._shm_addr)
        except Exception:
            pass
        shm, self._shm, self._shm_addr, self._shm_cap = self._shm, None, 0, 0
        # unlink first and on its own: close() raises BufferError while a HotPathResult still holds np.frombuffer
        # views of the segment, and a skipped unlink leaks hundreds of MB of page-locked /dev/shm until exit
        if self.rank == 0:
            try:
                shm.unlink()
            except Exception:
                pass
        try:
            shm.close()
        except BufferError:
            pass      # views alive: the mapping goes away with them; the name is already gone
        except Exception:
            pass

    def close(self):
        self._shm_release()

    # ------------------------------------------------------------------ first half, k > 15
    def _buf(self, name, n, dtype):
        """growth-only device buffer (hipMalloc/hipFree of GB-sized blocks costs more than the kernels)"""
        t = self.torch
        b = self._sbuf.get(name)
        if b is None or b.numel() < n:
            b = t.empty(max(int(n * 1.1), 1), dtype=dtype, device=self.device)
            self._sbuf[name] = b
        return b

    def _count_and_filter_sparse(self, d_pieces, host_rows_on_all_ranks=True):
        ctx, t, dist = self.ctx, self.torch, self.dist
        mine, W = self.my_chroms, self.world
        tt = time.perf_counter()
        ctx.genome_reset(len(mine))
        for li, gi in enumerate(mine):
            ctx.genome_add_device(li, d_pieces[li], self.lengths_bp[gi])
        ctx.sync()
        tt = self._t("pack", tt)
        if mine:
            ctx.count(self.k, self.lower_count, self.engine)
        tt = self._t("count", tt)
        # common splitters: quantiles of chromosome 0's sorted key list, chosen by its owner
        root = next(r_ for r_, o in enumerate(self.owned) if 0 in o)
        spl = t.zeros(max(W - 1, 1), dtype=t.int64, device=self.device)
        if W > 1:
            if self.rank == root:
                smp = ctx.sparse_sample(self.owned[root].index(0), 4096)
                if smp.size:
                    q = smp[(np.arange(1, W) * smp.size) // W]
                    spl[:W - 1] = t.from_numpy(q.view(np.int64).copy()).to(self.device)
            dist.broadcast(spl, src=root)
        splitters = spl.cpu().numpy()[:W - 1].view(np.uint64) if W > 1 else np.empty(0, np.uint64)
        bounds = [ctx.sparse_split(li, splitters) for li in range(len(mine))]
        sz = np.zeros((self.max_local, W), np.int64)
        for li, b in enumerate(bounds):
            sz[li] = np.diff(b)
        szt = t.from_numpy(sz).to(self.device)
        if W > 1:
            outs = [t.zeros_like(szt) for _ in range(W)]
            dist.all_gather(outs, szt)
            szall = np.stack([o.cpu().numpy() for o in outs])       # [source][its chromosome][destination]
        else:
            szall = sz[None]
        send_counts = sz.sum(axis=0)
        recv_counts = szall[:, :, self.rank].sum(axis=1)
        n_send, n_recv = int(send_counts.sum()), int(recv_counts.sum())
        keys_send, cnts_send = self._buf("ks", n_send, t.int64), self._buf("cs", n_send, t.int32)
        self._fence()
        off = 0
        for d in range(W):
            for li in range(len(mine)):
                n = int(sz[li, d])
                ctx.sparse_export(li, int(bounds[li][d]), n, keys_send.data_ptr() + off * 8, cnts_send.data_ptr() + off * 4)
                off += n
        ctx.sync()
        tt = self._t("split+export", tt)
        if W > 1:
            keys_recv, cnts_recv = self._buf("kr", n_recv, t.int64), self._buf("cr", n_recv, t.int32)
            # uneven splits: what this rank sends / receives must add up to its buffers, and over all ranks to the same total
            assert int(sum(send_counts.tolist())) == n_send and int(sum(recv_counts.tolist())) == n_recv, \
                "all_to_all_single splits: send %s != %d or recv %s != %d" % (send_counts.tolist(), n_send, recv_counts.tolist(), n_recv)
            chk = t.tensor([n_send, -n_recv], dtype=t.int64, device=self.device)
            dist.all_reduce(chk)
            assert int(chk[0].item()) == -int(chk[1].item()), "key-range exchange: %d keys sent, %d expected by the receivers" % (
                int(chk[0].item()), -int(chk[1].item()))
            dist.all_to_all_single(keys_recv[:n_recv], keys_send[:n_send], recv_counts.tolist(), send_counts.tolist())
            dist.all_to_all_single(cnts_recv[:n_recv], cnts_send[:n_send], recv_counts.tolist(), send_counts.tolist())
        else:
            keys_recv, cnts_recv = keys_send, cnts_send
        lens = t.zeros(self.C, dtype=t.int64, device=self.device)
        if mine:
            lens[t.tensor(mine, device=self.device)] = t.from_numpy(ctx.lengths()).to(self.device)
        dist.all_reduce(lens)
        lengths = lens.cpu().numpy()
        if self.device.type == "cuda":
            t.cuda.synchronize()
        tt = self._t("exchange+lengths", tt)
        pk, pc, ng = [0] * self.C, [0] * self.C, np.zeros(self.C, np.int64)
        off = 0
        for s_, owned in enumerate(self.owned):
            for li, gi in enumerate(owned):
                n = int(szall[s_, li, self.rank])
                pk[gi], pc[gi], ng[gi] = keys_recv.data_ptr() + off * 8, cnts_recv.data_ptr() + off * 4, n
                off += n
        ctx.sparse_view(pk, pc, ng, lengths, self.k, self.lower_count)
        try:
            n_union, n_rows, n_hist = ctx.filter(*self.csr, self.min_fold, self.baseline, self.min_freq,
                                                 self.max_freq, self.ratio)
            keys_t = t.empty((max(n_rows, 1),), dtype=t.int64, device=self.device)
            counts_t = t.empty((max(n_rows, 1), self.C), dtype=t.int32, device=self.device)
            self._fence()
            ctx.filter_fetch_device(keys_t.data_ptr(), counts_t.data_ptr(), None, n_rows)
        finally:
            ctx.sparse_view(None, None, None, None, 0, 0)
        tt = self._t("filter+fetch", tt)
        return self._gather_rows(keys_t, counts_t, n_rows, n_union, n_hist, lengths, host_rows_on_all_ranks, tt)

    # ------------------------------------------------------------------ second half
    def map_and_enrich(self, kmer_labels, n_sg, gather_bins=False):
        """K4-K6 over the local pieces.  Every rank ends up with all windows and their tests; `r.bins` holds the
        LOCAL pieces' slot counts (gather_bins=True: rank 0 gets the whole-chromosome slot arrays instead)."""
        ctx, t, dist = self.ctx, self.torch, self.dist
        mine = self.local_pieces
        r = HotPathResult()
        tt = time.perf_counter()
        ws, S = int(self.window_size), n_sg
        woff = np.zeros(self.C + 1, np.int64)      # whole-genome window rows, as the single-GPU path lays them out
        for c, n in enumerate(self.lengths_bp):
            woff[c + 1] = woff[c] + (int(n) + ws - 1) // ws + 1
        win_t = t.zeros((int(woff[-1]), S), dtype=t.int64, device=self.device)
        self._fence()
        r.bins, r.n_mapped = [], 0
        if mine:
            ctx.labels_set_from(kmer_labels, n_sg)
            tt = self._t("labels", tt)      # (the label tables are built on EVERY rank: the part of a pass that does not shrink with N)
            all_slots, n_mapped = ctx.map_bins_all(self.bin_size, self.chunk_size)
            r.bins = all_slots
            r.n_mapped = int(n_mapped.sum())
            ctx.stack_windows_dev(self.bin_size, self.chunk_size, ws, [woff[pc["chrom"]] for pc in mine],
                                  [pc["start"] for pc in mine], win_t.data_ptr())
        tt = self._t("map+stack", tt)
        if self.world > 1:
            dist.all_reduce(win_t)
        nm = t.tensor([r.n_mapped], dtype=t.int64, device=self.device)
        dist.all_reduce(nm)
        r.n_mapped = int(nm.item())
        win = win_t.cpu().numpy()
        with np.errstate(all="ignore"):
            pvals, argmin, sig, ratios = ctx.enrich_dev(win_t.data_ptr(), win.shape[0], S, self.max_pval, 0.5) \
                if win.shape[0] else (np.zeros((0, S)), np.zeros(0, np.int32), np.zeros(0, bool), np.zeros((0, S)))
        nz = np.flatnonzero(win.any(axis=1))
        chrom = np.searchsorted(woff, nz, side="right") - 1
        r.coord_chrom, r.coord_win, r.coord_labels, r.coord_ws = chrom, nz - woff[chrom], self.labels, ws
        r.window_counts = np.ascontiguousarray(win[nz])
        r.pvals, r.argmin, r.sig, r.ratios = pvals[nz], argmin[nz], np.asarray(sig[nz], bool), ratios[nz]
        tt = self._t("windows all-reduce+enrich", tt)
        if gather_bins:
            r.bins = self._gather_bins(r.bins, S)
        return r

    def _gather_bins(self, local_bins, S):
        """Whole-chromosome slot arrays on rank 0 (the `.subgenome.bin.count` lines): a piece that starts at a
        multiple of the chunk size numbers its slots like the chromosome does, shifted by a constant."""
        t = self.torch
        rows = []
        for pc, arr in zip(self.local_pieces, local_bins):
            shift = pc["start"] // self.bin_size + (pc["start"] // self.chunk_size if self.chunk_size > 0 else 0)
            nzr = np.flatnonzero(np.asarray(arr).any(axis=1))
            if nzr.size:
                rows.append(np.concatenate([np.full((nzr.size, 1), pc["chrom"], np.int64), (nzr + shift)[:, None],
                                            np.asarray(arr)[nzr].astype(np.int64)], axis=1))
        rows = np.concatenate(rows, axis=0) if rows else np.zeros((0, 2 + S), np.int64)
        allrows = self._all_gather_rows(rows, t.int64, to_host=self.rank == 0)
        if self.rank != 0:
            return None
        out = []
        for c, n in enumerate(self.lengths_bp):
            L = max(int(n), 1)
            ns = (L + self.bin_size - 1) // self.bin_size + ((L + self.k - 1) // self.chunk_size + 1 if self.chunk_size > 0 else 1)
            a = np.zeros((ns, S), np.int32)
            sel = allrows[allrows[:, 0] == c]
            np.add.at(a, sel[:, 1], sel[:, 2:].astype(np.int32))
            out.append(a)
        return out
