"""One pass of the whole hot path over a device-resident genome.

This is what `bench.py` times and what the pipeline would call when the genome
already lives in HBM: K0 pack -> K1/K2 count -> K3 matrix + differential filter
-> K4 label table -> K5 bin map -> window stack -> K6 enrichment.

Multi-GPU (one process per GPU, torch.distributed over RCCL): chromosomes are
sharded across ranks for K0-K2 and K5; the filter is sharded by k-mer slot
range after an all-to-all of table slices (the one real exchange step of the
path), see `DistPlan` below and DESIGN.md "Multi-GPU".
"""
import time

import numpy as np

from .config import sets_to_csr


class HotPathResult:
    """Plain result record.  `coords` (window row names, Circos.stack_matrix style) are built on first use
    from the chromosome / window index arrays: 14 K tuples are 1.5 ms of Python per pass."""
    _coords = None
    wait = staticmethod(lambda: None)     # overlapped device->host copies: valid after wait()
    coord_chrom = coord_win = None
    coord_labels = None
    coord_ws = 0

    @property
    def coords(self):
        if self._coords is None and self.coord_chrom is not None:
            ws, labels = self.coord_ws, self.coord_labels
            self._coords = [(labels[c], x * ws, x * ws + ws)
                            for c, x in zip(self.coord_chrom.tolist(), self.coord_win.tolist())]
        return self._coords

    @coords.setter
    def coords(self, value):
        self._coords = value


class HotPath:
    def __init__(self, ctx, labels, lengths, sgs, k=15, lower_count=3, min_fold=2.0, baseline=1,
                 min_freq=200, max_freq=1e9, ratio=1.0, bin_size=10000, chunk_size=10_000_000,
                 window_size=1_000_000, max_pval=0.05, engine=0):
        self.ctx, self.labels, self.lengths, self.sgs = ctx, list(labels), list(lengths), sgs
        self.k, self.lower_count, self.engine = k, lower_count, engine
        self.min_fold, self.baseline, self.min_freq, self.max_freq, self.ratio = (
            min_fold, baseline, min_freq, max_freq, ratio)
        self.bin_size, self.chunk_size, self.window_size, self.max_pval = (
            bin_size, chunk_size, window_size, max_pval)
        self.csr = sets_to_csr(sgs, self.labels)
        self.wall = {}     # host wall-clock per phase (seconds, accumulated), for bench.py

    def _t(self, name, t0):
        self.ctx.sync()
        t1 = time.perf_counter()
        self.wall[name] = self.wall.get(name, 0.0) + (t1 - t0)
        return t1

    # ---- first half: K0..K3 ------------------------------------------------
    def count_and_filter(self, d_ascii, want_freqs=False, sort=False, overlap=False):
        """d_ascii: device pointers of the ASCII chromosomes (already in HBM).
        overlap=True: the matrix rows travel to page-locked host memory on a copy stream while the caller goes on
        (e.g. with map_and_enrich); `result.wait()` makes keys / counts / tot valid."""
        ctx = self.ctx
        t = time.perf_counter()
        ctx.genome_reset(len(self.labels))
        t = self._t("genome_reset", t)
        for i, (ptr, n) in enumerate(zip(d_ascii, self.lengths)):
            ctx.genome_add_device(i, ptr, n)
        # (no synchronisation here: the counting lanes start on the chromosomes that are packed already)
        ctx.count(self.k, self.lower_count, self.engine)
        t = self._t("pack+count", t)
        r = HotPathResult()
        r.kmer_lengths = ctx.lengths()
        r.n_union, r.n_rows, r.n_hist = ctx.filter(*self.csr, self.min_fold, self.baseline, self.min_freq,
                                                   self.max_freq, self.ratio)
        t = self._t("filter", t)
        if overlap and not want_freqs and not sort and hasattr(ctx, "filter_fetch_async"):
            r.keys, r.counts, r.tot = ctx.filter_fetch_async(r.n_rows)
            r.freqs = None
            r.wait = ctx.filter_fetch_wait
            self.wall["filter_fetch"] = self.wall.get("filter_fetch", 0.0) + (time.perf_counter() - t)   # issue only
            return r
        r.keys, r.counts, r.freqs, r.tot = ctx.filter_fetch(r.n_rows, want_freqs=want_freqs, sort=sort,
                                                            pinned=not sort)
        t = self._t("filter_fetch", t)
        return r

    # ---- second half: K4..K6 -----------------------------------------------
    def map_and_enrich(self, kmer_labels, n_sg):
        """K4 labels -> K5 bin map -> window stack -> K6 enrichment."""
        ctx = self.ctx
        t = time.perf_counter()
        ctx.labels_set_from(kmer_labels, n_sg)
        t = self._t("labels_set", t)
        r = HotPathResult()
        all_slots, n_mapped = ctx.map_bins_all(self.bin_size, self.chunk_size)
        t = self._t("map_bins", t)
        r.n_mapped = int(n_mapped.sum())
        r.bins = all_slots
        # window stack + Fisher on the device (Circos.stack_matrix -> Stats.enrich semantics) in one call: the window
        # table never leaves HBM between the two; only the non-empty windows are rows (Circos.py:734-742)
        win, woff, pvals, argmin, sig, ratios = ctx.stack_enrich(self.bin_size, self.chunk_size, self.window_size,
                                                                 self.lengths, self.max_pval, 0.5)
        nz = np.flatnonzero(win.any(axis=1))
        chrom = np.searchsorted(woff, nz, side="right") - 1
        r.coord_chrom, r.coord_win, r.coord_labels, r.coord_ws = chrom, nz - woff[chrom], self.labels, self.window_size
        r.window_counts = win[nz].astype(np.int64)
        r.pvals, r.argmin, r.sig, r.ratios = pvals[nz], argmin[nz], sig[nz].astype(bool), ratios[nz]
        t = self._t("enrich", t)
        return r
