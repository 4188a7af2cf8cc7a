"""OracleContext: the CPU oracle (oracle/pyoracle.py) behind the same Python
interface as subphaser_amd._native.Context.  TEST INFRASTRUCTURE ONLY -- it lets
the host-side mirror modules run on CPU against the golden vectors, and it is
the reference implementation the `-m gpu` parity tests compare the HIP path with."""
import numpy as np
import pyoracle as po


class OracleContext:
    h = "oracle"

    def __init__(self, nthreads=2):
        self.nthreads = nthreads
        self.seqs = []
        self.k = None
        self.n_chrom = 0

    def close(self):
        pass

    def sync(self):
        pass

    # genome
    def genome_reset(self, n):
        self.seqs = [np.empty(0, np.uint8)] * n
        self.n_chrom = n

    def genome_add(self, i, seq):
        self.seqs[i] = po._ascii(seq).copy()

    def genome_len(self, i):
        return int(self.seqs[i].size)

    def genome_unpack(self, i):
        a = self.seqs[i].copy()
        up = a & 0xDF
        ok = (up == 65) | (up == 67) | (up == 71) | (up == 84)
        return np.where(ok, up, ord("N")).astype(np.uint8)

    # count
    def count(self, k, lower_count=3, engine=0):
        if k < 1 or k > 32:
            raise ValueError("k=%d unsupported (1..32)" % k)
        self.k, self.lower = k, max(1, lower_count)
        self.dumps = [po.count(s, k, self.lower, self.nthreads) for s in self.seqs]

    def lengths(self):
        return np.array([int(d[1].astype(np.int64).sum()) for d in self.dumps], np.int64)

    def dump(self, i, sort=True):
        return self.dumps[i]

    # filter
    def filter(self, set_off, unit_off, unit_chrom, min_fold, baseline, min_freq, max_freq, ratio):
        sgs = []
        for s in range(len(set_off) - 1):
            sgs.append([[int(c) for c in unit_chrom[unit_off[u]:unit_off[u + 1]]]
                        for u in range(set_off[s], set_off[s + 1])])
        self._f = po.filter_dumps(self.dumps, sgs, list(range(self.n_chrom)), min_fold, baseline, min_freq,
                                  max_freq, ratio)
        return self._f.n_union, len(self._f.keys), len(self._f.hist)

    def filter_fetch(self, n_rows, want_freqs=True, sort=True, pinned=False):
        f = self._f
        return f.keys, f.counts, f.freqs, f.tot

    def filter_hist(self, n):
        return self._f.hist

    # labels / map
    def labels_set(self, keys, sg, n_sg):
        self.lab_keys = np.ascontiguousarray(keys, np.uint64)
        self.lab_sg = np.ascontiguousarray(sg, np.uint8)
        self.n_sg = n_sg
        self.hit = np.zeros(len(self.lab_keys), np.uint8)

    def labels_set_from(self, labels, n_sg):       # (the oracle has no device: the host arrays)
        self.labels_set(labels.keys, labels.sg_idx, n_sg)

    def map_nslots(self, i, bin_size, chunk_size):
        return po.n_slots(self.seqs[i].size, bin_size, chunk_size, self.k)

    def map_bins(self, i, bin_size=10000, chunk_size=10_000_000):
        out, hit, n = po.map_bins(self.seqs[i], self.k, self.lab_keys, self.lab_sg, self.n_sg, bin_size,
                                  chunk_size, self.nthreads)
        self.hit |= hit
        return out, n

    def map_bins_all(self, bin_size=10000, chunk_size=10_000_000):
        res = [self.map_bins(i, bin_size, chunk_size) for i in range(self.n_chrom)]
        off = np.concatenate(([0], np.cumsum([len(r[0]) for r in res]))).astype(np.int64)
        self.last_map = (np.concatenate([r[0] for r in res]), off)
        return [r[0] for r in res], np.array([r[1] for r in res], np.int64)

    def stack_windows(self, bin_size, chunk_size, window_size, lengths):
        big, off = self.last_map
        woff = np.zeros(self.n_chrom + 1, np.int64)
        for i, n in enumerate(lengths):
            woff[i + 1] = woff[i] + (int(n) + int(window_size) - 1) // int(window_size) + 1
        win = np.zeros((int(woff[-1]), self.n_sg), np.int64)
        for c in range(self.n_chrom):
            for slot in range(int(off[c]), int(off[c + 1])):
                row = big[slot]
                if not row.any():
                    continue
                local = slot - int(off[c])
                chunk = 0
                if chunk_size:
                    j = 1
                    while (j * chunk_size - (self.k - 1)) // bin_size + j <= local:
                        chunk = j
                        j += 1
                win[woff[c] + (local - chunk) * bin_size // window_size] += row
        return win, woff

    def stack_enrich(self, bin_size, chunk_size, window_size, lengths, max_pval=0.05, min_ratio=0.5):
        win, woff = self.stack_windows(bin_size, chunk_size, window_size, lengths)
        W, S = win.shape
        pvals, ratios = np.ones((W, S)), np.full((W, S), np.nan)
        argmin, sig = np.zeros(W, np.int32), np.zeros(W, bool)
        nz = np.flatnonzero(win.any(axis=1))
        if nz.size:     # empty windows change no column total: testing the non-empty rows alone is the same thing
            with np.errstate(all="ignore"):
                pvals[nz], argmin[nz], sig[nz], ratios[nz] = self.enrich(win[nz])
        return win, woff, pvals, argmin, sig, ratios

    def map_features(self, seqs):
        out = np.zeros((len(seqs), self.n_sg), np.int64)
        for f, s in enumerate(seqs):
            a = po._ascii(s)
            if a.size == 0:
                continue
            sc, hit, _ = po.map_bins(a, self.k, self.lab_keys, self.lab_sg, self.n_sg, max(a.size, 1), 0, 1)
            out[f] = sc.sum(axis=0)
            self.hit |= hit
        return out

    def map_features_cat(self, cat, off):
        cat = np.asarray(cat, np.uint8)
        return self.map_features([cat[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)])

    def map_intervals(self, chrom, start, end):
        """sp_map_intervals: the sub-sequences cut out of the resident chromosomes, mapped like features"""
        return self.map_features([po._ascii(self.seqs[int(c)])[int(a):int(b)] for c, a, b in zip(chrom, start, end)])

    def labels_hit(self):
        return int(self.hit.sum())

    # enrich
    def enrich(self, counts, max_pval=0.05, min_ratio=0.5):
        return po.enrich(counts, max_pval, min_ratio)


class OracleDistContext(OracleContext):
    """OracleContext + the multi-GPU entry points (dense tables in caller-owned memory,
    slot-range filter views), for the gloo tests of subphaser_amd/dist.py.  Pointers are
    addresses of CPU torch tensors."""

    def __init__(self, nthreads=1):
        super().__init__(nthreads)
        self.bound = {}
        self.view = None

    def nslots(self, k):
        from subphaser_amd import kmer
        return kmer.dense_slots(k)

    @staticmethod
    def _view_u8(ptr, n):
        import ctypes
        return np.frombuffer((ctypes.c_uint8 * n).from_address(int(ptr)), dtype=np.uint8)

    def tables_bind(self, i, ptr):
        self.bound[i] = ptr

    # byte tables (raw count saturated at 255 + overflow pairs): what sp_count leaves in bound memory
    def table_overflow(self, i, d_pairs=None, cap=0):
        import ctypes
        ov = self.ovf.get(i, np.zeros((0, 2), np.uint32))
        if d_pairs and len(ov):
            assert cap >= len(ov)
            o = np.frombuffer((ctypes.c_uint32 * (2 * len(ov))).from_address(int(d_pairs)), np.uint32).reshape(-1, 2)
            o[:] = ov
        return int(len(ov))

    @staticmethod
    def _pairs(ptr, n):
        import ctypes
        if not n:
            return np.zeros((0, 2), np.uint32)
        return np.frombuffer((ctypes.c_uint32 * (2 * int(n))).from_address(int(ptr)), np.uint32).reshape(-1, 2)

    def _decode(self, d_tab, d_ovf, n_ovf, slot_base, n):
        """exact raw counts of a byte slice"""
        arr = self._view_u8(d_tab, n).astype(np.uint32)
        o = self._pairs(d_ovf, n_ovf)
        loc = o[:, 0].astype(np.int64) - slot_base
        ok = (loc >= 0) & (loc < n)
        assert (arr[loc[ok]] == 255).all()
        arr[loc[ok]] = o[ok, 1]
        assert int((arr == 255).sum()) == int((o[ok, 1] == 255).sum())    # every saturated byte has its pair
        return arr

    def table_merge(self, d_dst, d_dst_ovf, n_dst_ovf, d_src, d_src_ovf, n_src_ovf, slot_base, n, d_out_ovf, cap):
        tot = self._decode(d_dst, d_dst_ovf, n_dst_ovf, slot_base, n) + self._decode(d_src, d_src_ovf, n_src_ovf, slot_base, n)
        self._view_u8(d_dst, n)[:] = np.minimum(tot, 255).astype(np.uint8)
        big = np.flatnonzero(tot >= 255)
        if big.size > cap:
            raise MemoryError("overflow capacity")
        if big.size:
            o = self._pairs(d_out_ovf, big.size)
            o[:, 0] = big + slot_base
            o[:, 1] = tot[big]
        return int(big.size)

    def table_lengths(self, d_tab, d_ovf, n_ovf, slot_base, n, lower_count):
        arr = self._decode(d_tab, d_ovf, n_ovf, slot_base, n).astype(np.int64)
        keep = arr >= max(1, lower_count)
        return int(arr[keep].sum()), int(keep.sum())

    def stack_windows_dev(self, bin_size, chunk_size, window_size, win_off, seg_start, d_win):
        import ctypes
        big, off = self.last_map
        S = self.n_sg
        tot_rows = None
        for c in range(self.n_chrom):
            for slot in range(int(off[c]), int(off[c + 1])):
                row = big[slot]
                if not row.any():
                    continue
                local = slot - int(off[c])
                chunk = 0
                if chunk_size:
                    j = 1
                    while (j * chunk_size - (self.k - 1)) // bin_size + j <= local:
                        chunk = j
                        j += 1
                pos = (int(seg_start[c]) if seg_start is not None else 0) + (local - chunk) * bin_size
                w = int(win_off[c]) + pos // window_size
                dst = np.frombuffer((ctypes.c_int64 * S).from_address(int(d_win) + 8 * S * w), np.int64)
                dst += row

    def enrich_dev(self, d_counts, W, S, max_pval=0.05, min_ratio=0.5):
        import ctypes
        win = np.frombuffer((ctypes.c_int64 * (W * S)).from_address(int(d_counts)), np.int64).reshape(W, S)
        pvals, ratios = np.ones((W, S)), np.full((W, S), np.nan)
        argmin, sig = np.zeros(W, np.int32), np.zeros(W, bool)
        nz = np.flatnonzero(win.any(axis=1))
        if nz.size:
            with np.errstate(all="ignore"):
                pvals[nz], argmin[nz], sig[nz], ratios[nz] = self.enrich(win[nz].copy())
        return pvals, argmin, sig, ratios

    def host_register(self, h_ptr, nbytes):
        pass

    def host_unregister(self, h_ptr):
        pass

    def dev_to_host_ptr(self, h_ptr, d_ptr, nbytes):
        import ctypes
        if nbytes:
            ctypes.memmove(int(h_ptr), int(d_ptr), int(nbytes))

    def genome_add_device(self, i, arr, n):
        self.genome_add(i, np.asarray(arr[:n], np.uint8))

    def count(self, k, lower_count=3, engine=0):
        from subphaser_amd import kmer
        super().count(k, lower_count, engine)
        n = kmer.dense_slots(k)
        self.ovf = {}
        for i, s in enumerate(self.seqs):
            if i in self.bound:
                keys, cnts = po.count(s, k, 1, self.nthreads)     # raw counts, threshold applied on read
                tab = self._view_u8(self.bound[i], n)
                tab[:] = 0
                slots = kmer.slots_of_keys(keys, k).astype(np.int64)
                tab[slots] = np.minimum(cnts, 255).astype(np.uint8)
                big = np.flatnonzero(cnts >= 255)
                o = np.argsort(slots[big], kind="stable")
                self.ovf[i] = np.stack([slots[big][o].astype(np.uint32), cnts[big][o].astype(np.uint32)], axis=1) \
                    if big.size else np.zeros((0, 2), np.uint32)

    def filter_fetch_device(self, d_keys, d_counts, d_tot, n_rows):
        import ctypes
        f = self._f
        if n_rows == 0:
            return
        C_ = f.counts.shape[1]
        if d_keys:
            np.frombuffer((ctypes.c_uint64 * n_rows).from_address(int(d_keys)), np.uint64)[:] = f.keys
        if d_counts:
            np.frombuffer((ctypes.c_uint32 * (n_rows * C_)).from_address(int(d_counts)), np.uint32)[:] = f.counts.ravel()
        if d_tot:
            np.frombuffer((ctypes.c_uint64 * n_rows).from_address(int(d_tot)), np.uint64)[:] = f.tot

    def count_range(self, k, lower_count, engine, first, last):
        if first == 0:
            self.count(k, lower_count, engine)      # the double counts everything on the first call

    # ---- k > 15: key-range exchange of the sorted (key, count) lists
    def sparse_sizes(self):
        return np.array([len(d[0]) for d in self.dumps], np.int64)

    def sparse_sample(self, chrom, n_samples):
        keys = self.dumps[chrom][0]
        stride = len(keys) // n_samples
        return keys.copy() if stride < 1 else keys[::stride][:n_samples].copy()

    def sparse_split(self, chrom, splitters):
        keys = self.dumps[chrom][0]
        b = np.searchsorted(keys, np.asarray(splitters, np.uint64), side="left")
        return np.concatenate(([0], b, [len(keys)])).astype(np.int64)

    def sparse_export(self, chrom, first, count, d_keys, d_counts):
        import ctypes
        if count == 0:
            return
        keys, cnts = self.dumps[chrom]
        np.frombuffer((ctypes.c_uint64 * count).from_address(int(d_keys)), np.uint64)[:] = keys[first:first + count]
        np.frombuffer((ctypes.c_uint32 * count).from_address(int(d_counts)), np.uint32)[:] = cnts[first:first + count]

    def sparse_view(self, d_keys, d_counts, n, lengths, k, lower_count):
        import ctypes
        if d_keys is None:
            self.sview = None
            return
        dumps = []
        for pk, pc, m in zip(d_keys, d_counts, n):
            m = int(m)
            if m == 0:
                dumps.append((np.empty(0, np.uint64), np.empty(0, np.uint32)))
                continue
            dumps.append((np.frombuffer((ctypes.c_uint64 * m).from_address(int(pk)), np.uint64).copy(),
                          np.frombuffer((ctypes.c_uint32 * m).from_address(int(pc)), np.uint32).copy()))
        self.sview = (dumps, np.array(lengths, np.int64))
        self.k = k

    def filter_view(self, ptrs, slot_base, nview, lengths, k, lower_count, d_ovf=None, n_ovf=None):
        self.view = None if ptrs is None else (list(ptrs), int(slot_base), int(nview), np.array(lengths), k, lower_count,
                                               list(d_ovf) if d_ovf is not None else None,
                                               list(n_ovf) if n_ovf is not None else None)
        if ptrs is not None:
            self.k = k

    def filter(self, set_off, unit_off, unit_chrom, min_fold, baseline, min_freq, max_freq, ratio):
        sgs = []
        for s in range(len(set_off) - 1):
            sgs.append([[int(c) for c in unit_chrom[unit_off[u]:unit_off[u + 1]]]
                        for u in range(set_off[s], set_off[s + 1])])
        if getattr(self, "sview", None) is not None:
            dumps, lengths = self.sview
            self._f = po.filter_dumps(dumps, sgs, list(range(len(dumps))), min_fold, baseline, min_freq, max_freq,
                                      ratio, lengths=lengths)
            return self._f.n_union, len(self._f.keys), len(self._f.hist)
        if self.view is None:
            return super().filter(set_off, unit_off, unit_chrom, min_fold, baseline, min_freq, max_freq, ratio)
        from subphaser_amd import kmer
        import ctypes
        ptrs, base, nview, lengths, k, L, d_ovf, n_ovf = self.view
        dumps = []
        for ci, p in enumerate(ptrs):
            arr = self._view_u8(p, nview).astype(np.uint32)
            if d_ovf is not None and n_ovf[ci]:
                o = np.frombuffer((ctypes.c_uint32 * (2 * int(n_ovf[ci]))).from_address(int(d_ovf[ci])),
                                  np.uint32).reshape(-1, 2)
                loc = o[:, 0].astype(np.int64) - base
                ok = (loc >= 0) & (loc < nview)
                assert (arr[loc[ok]] == 255).all()
                arr[loc[ok]] = o[ok, 1]
            assert not (arr == 255).any() or d_ovf is not None
            idx = np.flatnonzero(arr >= L)
            keys = kmer.keys_of_slots((idx + base).astype(np.uint64), k)
            o = np.argsort(keys, kind="stable")
            dumps.append((keys[o], arr[idx][o].astype(np.uint32)))
        sgs = []
        for s in range(len(set_off) - 1):
            sgs.append([[int(c) for c in unit_chrom[unit_off[u]:unit_off[u + 1]]]
                        for u in range(set_off[s], set_off[s + 1])])
        self._f = po.filter_dumps(dumps, sgs, list(range(len(ptrs))), min_fold, baseline, min_freq, max_freq,
                                  ratio, lengths=lengths)
        return self._f.n_union, len(self._f.keys), len(self._f.hist)
