"""K-mer counting, chromosome x k-mer matrix and differential filter.

Host-side mirror of the live part of the reference's subphaser/Jellyfish.py:
  run_jellyfish_dumps / run_jellyfish_dump   (Jellyfish.py:671-704)
  JellyfishDumps.to_matrix / filter / write_matrix (Jellyfish.py:430-522)
  _filter_kmer                               (Jellyfish.py:611-648)
Same names and argument meaning; the external `jellyfish` binary, the text
dumps and the Python dict are replaced by HIP kernels over dense tables in HBM
(K1-K3, subphaser_amd/csrc/sp_count*.hip, sp_filter.hip).
"""
import os
from collections import OrderedDict

import numpy as np

from . import _native
from . import kmer as kmerlib
from .config import sets_to_csr
from .runtime import get_context, logger
from .seqs import load_chromfile
from ._native import write_chunks


class KmerDump(str):
    """What run_jellyfish_dumps returns per chromosome.  It IS the dump file name
    (`{chromfile}_{k}.fa`, Jellyfish.py:690) so existing callers can treat it as a
    path, and it remembers which GPU table holds the counts."""

    def __new__(cls, path, ctx, index, k, lower_count):
        obj = str.__new__(cls, path)
        obj.ctx, obj.index, obj.k, obj.lower_count = ctx, index, k, lower_count
        return obj

    def fetch(self):
        """(canonical keys ascending, counts) with count >= lower_count."""
        return self.ctx.dump(self.index)

    def write_text(self, path=None):
        """jellyfish `dump -c` text: `KMER COUNT` per line (+ empty .ok checkpoint)."""
        path = path or str(self)
        keys, counts = self.fetch()
        kmers = kmerlib.decode_many(keys, self.k)
        with open(path, "w") as f:
            f.write("".join("%s %d\n" % (s, c) for s, c in zip(kmers, counts.tolist())))
        open(path + ".ok", "w").close()
        return path


def run_jellyfish_dumps(seqfiles, ncpu=4, k=17, lower_count=2, threads=4, overwrite=False,
                        write_dumps=False, engine=0, ctx=None, **kargs):
    """Count canonical k-mers of every chromosome file on the GPU.
    ncpu/threads/overwrite are accepted for call compatibility (__main__.py:403-404)."""
    ctx = ctx or get_context()
    seqfiles = list(seqfiles)
    ctx.genome_reset(len(seqfiles))
    for i, seqfile in enumerate(seqfiles):
        rec = load_chromfile(seqfile)
        ctx.genome_add(i, rec.seq)
        rec.index[id(ctx)] = i
    ctx.count(k, lower_count, engine)
    dumps = [KmerDump("{}_{}.fa".format(f, k), ctx, i, k, lower_count) for i, f in enumerate(seqfiles)]
    if write_dumps:
        for d in dumps:
            d.write_text()
    return dumps


def run_jellyfish_dump(seqfile, threads=4, k=17, prefix=None, lower_count=2, method="jellyfish",
                       overwrite=False, **kargs):
    return run_jellyfish_dumps([seqfile], k=k, lower_count=lower_count, **kargs)[0]


class DeviceMatrix:
    """Handle for the chromosome x k-mer count matrix resident in HBM (the
    reference's d_mat dict, Jellyfish.py:439-460).  len() = number of k-mers
    seen (count >= lower_count) in at least one chromosome."""

    def __init__(self, ctx, labels):
        self.ctx, self.labels = ctx, labels
        self._n = None

    def __len__(self):
        if self._n is None:
            # filter() reports the union size as a by-product; before that, join the dumps on the host
            keys = [self.ctx.dump(i, sort=False)[0] for i in range(len(self.labels))]
            self._n = int(len(np.unique(np.concatenate(keys)))) if keys else 0
        return self._n


class FilteredMatrix:
    """The reference's d_mat2 (k-mer -> [count/length per chromosome]) as arrays."""

    def __init__(self, keys, counts, freqs, tot, k, labels, lengths=None, ctx=None):
        self.keys, self.counts, self.freqs, self.tot = keys, counts, freqs, tot
        self.k, self.labels = k, labels
        self.lengths, self.ctx = lengths, ctx      # count / length is how freqs was formed (Jellyfish.py:648)

    def __len__(self):
        return len(self.keys)

    def kmers(self):
        return kmerlib.decode_many(self.keys, self.k)

    def items(self):
        for s, row in zip(self.kmers(), self.freqs.tolist()):
            yield s, row


class JellyfishDumps:
    def __init__(self, dumpfiles, labels=None, ncpu=4, method="map", chunksize=None, **kargs):
        self.dumpfiles = list(dumpfiles)
        self.labels = labels
        self.ncpu, self.method, self.chunksize = ncpu, method, chunksize
        ctxs = {id(d.ctx) for d in self.dumpfiles if isinstance(d, KmerDump)}
        if len(ctxs) != 1 or not all(isinstance(d, KmerDump) for d in self.dumpfiles):
            raise ValueError("JellyfishDumps needs the KmerDump handles returned by run_jellyfish_dumps")
        self.ctx = self.dumpfiles[0].ctx
        self.k = self.dumpfiles[0].k

    def __len__(self):
        return len(self.dumpfiles)

    def to_matrix(self, array=False):
        """The outer join already exists as C dense tables in HBM; only `lengths`
        (sum of the dumped counts per chromosome, Jellyfish.py:97,449) comes back."""
        self.lengths = [int(x) for x in self.ctx.lengths()]
        for d in self.dumpfiles:
            logger.info("Loading " + str(d))
        return DeviceMatrix(self.ctx, self.labels)

    def filter(self, d_mat, lengths, sgs, outfig=None, by_count=False, min_freq=200, max_freq=10000,
               min_fold=2, baseline=1, min_prop=None, max_prop=None, ratio=1):
        # `lengths` is ignored exactly like the reference does (it uses self.lengths, Jellyfish.py:467,486)
        tot_lens = sum(self.lengths)
        if min_prop is not None:
            min_freq = min_prop * tot_lens
            logger.info("Adjust `min_freq` to {} according to `min_prop`".format(min_freq))
        if max_prop is not None:
            max_freq = max_prop * tot_lens
            logger.info("Adjust `max_freq` to {} according to `max_prop`".format(max_freq))
        if min_freq > max_freq:
            raise ValueError("`min_freq` ({}) should be lower than `max_freq` ({})".format(min_freq, max_freq))
        n_single = 0
        for sg in sgs:
            if len(sg) == 1:
                logger.warning("Singleton `{}` is ignored".format(sg))
                n_single += 1
        if n_single == len(sgs):
            raise ValueError("All singletons are not allowed")
        d_lens = OrderedDict(zip(self.labels, self.lengths))
        lens0 = [lab for lab, _len in d_lens.items() if _len == 0]
        if lens0:
            raise ValueError("Chromosomes `{}` have only 0 kmers".format(lens0))
        set_off, unit_off, unit_chrom = sets_to_csr(sgs, self.labels)
        n_union, n_rows, n_hist = self.ctx.filter(set_off, unit_off, unit_chrom, min_fold, baseline,
                                                  min_freq, max_freq, ratio)
        d_mat._n = n_union
        self.n_union, self.n_hist = n_union, n_hist
        keys, counts, freqs, tot = self.ctx.filter_fetch(n_rows)
        logger.info("After filtering, remained {} ({:.2%}) differential (freq >= {}) and {} ({:.2%}) "
                    "candidate (freq > 0) kmers".format(n_rows, n_rows / max(n_union, 1), min_freq, n_hist,
                                                        n_hist / max(n_union, 1)))
        if outfig is not None:
            if n_hist == 0:
                raise ValueError("0 kmer with fold > {}. Please reset the filter options.".format(min_fold))
            self.tot_freqs = None   # fetched lazily by plot_histogram
        return FilteredMatrix(keys, counts, freqs, tot, self.k, self.labels, lengths=list(self.lengths), ctx=self.ctx)

    def hist_tot(self):
        """tot of every fold-passing k-mer (the reference's tot_freqs, Jellyfish.py:499-502)."""
        return self.ctx.filter_hist(self.n_hist)

    def write_matrix(self, d_mat, fout):
        """`.kmer.mat`: header `kmer <labels>`, rows k-mer + str(count/length)
        (Jellyfish.py:515-520; read back by Data.py:6-21)."""
        fout.write("\t".join(["kmer"] + list(self.labels)) + "\n")
        keys, freqs, k = d_mat.keys, d_mat.freqs, d_mat.k
        if len(keys) and _native.text_kmer_matrix(fout, keys, k, freqs):     # rows formatted by the library's threads
            return

        def fmt(lo, hi):
            kmers = kmerlib.decode_many(keys[lo:hi], k)
            return "".join(km + "\t" + "\t".join(map(repr, row)) + "\n"
                           for km, row in zip(kmers, freqs[lo:hi].tolist()))
        write_chunks(fout, len(keys), fmt)


def plot_histogram(data, outfig, step=25, xlim=99, xlabel="Kmer occurrence", ylabel="Count", vline=None):
    """Same figure as Jellyfish.py:650-666 (visualisation; optional)."""
    try:
        from matplotlib import pyplot as plt
    except ImportError:
        logger.warning("matplotlib missing: skipping " + outfig)
        return
    plt.switch_backend("agg")
    data = np.asarray(data)
    nbins = max(1, int((data.max() - 0) / step))
    plt.figure(figsize=(7, 5), dpi=300, tight_layout=True)
    # plt.hist(data, bins=nbins) as the reference calls it, drawn as ONE filled step outline of the bins left of the
    # x limit instead of one patch per bin: the histogram of millions of row sums has tens of thousands of bins
    # (seconds of patch drawing at wheat scale); the visible figure is the same solid bars
    counts, edges = np.histogram(data, bins=nbins)
    right = np.percentile(data, xlim)
    nvis = int(np.searchsorted(edges[:-1], right, side="right"))
    if hasattr(plt, "stairs"):
        plt.stairs(counts[:nvis], edges[:nvis + 1], fill=True, color="C0")
        plt.ylim(0, max(1, counts.max()) * 1.05)
    else:
        plt.bar(edges[:nvis], counts[:nvis], width=np.diff(edges)[:nvis], align="edge", color="C0")
    plt.xlim(0, right)
    plt.xlabel(xlabel)
    plt.ylabel(ylabel)
    plt.ticklabel_format(style="plain")
    if vline is not None:
        plt.axvline(vline, ls="--", c="grey")
    plt.savefig(outfig, bbox_inches="tight", dpi=300)
    plt.close()
