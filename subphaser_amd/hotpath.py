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
    pass


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
    def count_and_filter(self, d_ascii, want_freqs=False, sort=False):
        """d_ascii: device pointers of the ASCII chromosomes (already in HBM)."""
        ctx = self.ctx
        t = time.perf_counter()
        ctx.genome_reset(len(self.labels))
        t = self._t("genome_reset", t)
        for i, (ptr, n) in enumerate(zip(d_ascii, self.lengths)):
            ctx.genome_add_device(i, ptr, n)
        t = self._t("pack", t)
        ctx.count(self.k, self.lower_count, self.engine)
        t = self._t("count", t)
        r = HotPathResult()
        r.kmer_lengths = ctx.lengths()
        r.n_union, r.n_rows, r.n_hist = ctx.filter(*self.csr, self.min_fold, self.baseline, self.min_freq,
                                                   self.max_freq, self.ratio)
        t = self._t("filter", t)
        r.keys, r.counts, r.freqs, r.tot = ctx.filter_fetch(r.n_rows, want_freqs=want_freqs, sort=sort)
        t = self._t("filter_fetch", t)
        return r

    # ---- second half: K4..K6 -----------------------------------------------
    def map_and_enrich(self, kmer_labels, n_sg):
        ctx = self.ctx
        t = time.perf_counter()
        ctx.labels_set(kmer_labels.keys, kmer_labels.sg_idx, n_sg)
        t = self._t("labels_set", t)
        r = HotPathResult()
        r.bins, coords, rows, r.n_mapped = [], [], [], 0
        for i, (lab, n) in enumerate(zip(self.labels, self.lengths)):
            slots, nm = ctx.map_bins(i, self.bin_size, self.chunk_size)
            r.n_mapped += nm
            r.bins.append(slots)
            nz = np.flatnonzero(slots.any(axis=1))
            if nz.size == 0:
                continue
            # slot -> bin start (boundary bins split across two slots are summed by the window stack)
            if self.chunk_size:
                nch = (n + (self.k - 1)) // self.chunk_size + 1
                j = np.arange(1, nch + 1, dtype=np.int64)
                first_slot = (j * self.chunk_size - (self.k - 1)) // self.bin_size + j
                bins = nz - np.searchsorted(first_slot, nz, side="right")
            else:
                bins = nz
            # bins ascend, so each window is one contiguous run: segment sums instead of a scatter-add
            win = (bins * self.bin_size) // self.window_size
            seg = np.concatenate(([0], np.flatnonzero(np.diff(win)) + 1))
            summed = np.add.reduceat(slots[nz].astype(np.int64), seg, axis=0)
            for w in win[seg].tolist():
                coords.append((lab, int(w * self.window_size), int(w * self.window_size + self.window_size)))
            rows.append(summed)
        r.coords = coords
        r.window_counts = np.concatenate(rows) if rows else np.zeros((0, n_sg), np.int64)
        t = self._t("map_bins+stack", t)
        if len(r.window_counts):
            with np.errstate(all="ignore"):
                r.pvals, r.argmin, r.sig, r.ratios = ctx.enrich(r.window_counts, self.max_pval, 0.5)
        t = self._t("enrich", t)
        return r
