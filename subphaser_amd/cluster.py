"""Chromosome -> subgenome assignment and subgenome-specific k-mer test ("next" row f-1).

The reference's Cluster.py is the unchanged CONSUMER of the `.kmer.mat` matrix
(BASELINE.json north_star).  This module only re-states the part that
produces the input of the second half of the hot path:
  Cluster.__init__/fit/sort_subgenomes/assign_subgenomes (Cluster.py:17-47, 114-143)
  Cluster.output_kmers / _output_kmers                    (Cluster.py:151-194)
so that an end-to-end run is self-contained.  KMeans is delegated to
scikit-learn exactly like the reference; bootstrap and PCA plots stay with the
reference (visualisation, out of scope).  The per-k-mer Student t-test is
vectorised over the M x C matrix instead of looped through a process pool.
"""
import sys
from collections import OrderedDict

import numpy as np

from . import kmer as kmerlib
from .runtime import logger
from .seqs import KmerLabels


def load_matrix(datafile):
    """`.kmer.mat` reader with the contract of the reference's Data.py:6-21."""
    colnames, rownames, data = [], [], []
    with open(datafile) as fh:
        for i, line in enumerate(fh):
            t = line.strip().split()
            if i == 0:
                colnames = t[1:]
                continue
            rownames.append(t[0])
            data.append(list(map(float, t[1:])))
    return colnames, rownames, np.array(data, np.float64).reshape(len(rownames), len(colnames))


class Cluster:
    def __init__(self, data, n_clusters, sg_prefix="SG", sg_assigned={}, re_assign=True, bootstrap=False,
                 replicates=1000, jackknife=80, **kargs):
        """data: path of a `.kmer.mat` file or a FilteredMatrix (jellyfish.filter result)."""
        if isinstance(data, str):
            self.chrs, kmers, self.raw_data = load_matrix(data)
            self.k = len(kmers[0]) if kmers else 0
            self.keys = kmerlib.encode_many(kmers)
        else:
            self.chrs, self.raw_data, self.keys, self.k = list(data.labels), data.freqs, data.keys, data.k
        self.n_clusters, self.sg_prefix = n_clusters, sg_prefix
        if sg_assigned:
            logger.info("Skip k-means clustering")
            labels = [sg_assigned[c] for c in self.chrs]
            self.n_clusters = len(set(sg_assigned.values()))
            self.d_sg = self.assign_subgenomes(labels=labels) if re_assign else sg_assigned
            if not re_assign:
                self.sg_names = sorted(set(sg_assigned.values()))
        else:
            self.kmean = self.fit(self.normalize_data(self.raw_data.transpose()), n_clusters)
            self.d_sg = self.assign_subgenomes()
        self.d_bs = {c: "NA" for c in self.chrs}

    @staticmethod
    def normalize_data(data, axis=0):
        with np.errstate(all="ignore"):
            return (data - data.mean(axis=axis)) / data.std(axis=axis)

    def fit(self, data, n_clusters, **kargs):
        from sklearn.cluster import KMeans
        kmean = KMeans(n_clusters=n_clusters)
        kmean.fit(data)
        return kmean

    def sort_subgenomes(self, labels):
        assert len(self.chrs) == len(labels)
        d_map = {}
        for label, _ in sorted(zip(labels, self.chrs), key=lambda x: x[1]):
            if label not in d_map:
                d_map[label] = (max(d_map.values()) + 1) if d_map else 0
        return [d_map[label] for label in labels]

    def assign_subgenomes(self, base=1, labels=None):
        if labels is None:
            labels = self.kmean.labels_
        fmt = "{{}}{{:0>{}d}}".format(len(str(self.n_clusters)))
        self.labels = labels = self.sort_subgenomes(list(labels))
        d_sg = OrderedDict((c, fmt.format(self.sg_prefix, lab + base)) for lab, c in zip(labels, self.chrs))
        self.sg_names = sorted(set(d_sg.values()))
        return d_sg

    def output_subgenomes(self, fout=sys.stdout):
        print("\t".join(["#chrom", "subgenome", "bootstrap"]), file=fout)
        for c, sg in sorted(self.d_sg.items(), key=lambda x: x[1]):
            print("\t".join(map(str, [c, sg, self.d_bs[c]])), file=fout)

    def output_kmers(self, fout=sys.stdout, max_pval=0.05, ncpu=4, method="map", test_method="ttest_ind"):
        """Student t-test (pooled variance, two-sided) between the highest-mean and the
        second-highest-mean subgenome groups of every k-mer; keeps p <= max_pval.
        Returns KmerLabels (array form of the reference's d_ksg dict)."""
        if test_method != "ttest_ind":
            raise ValueError("only ttest_ind is implemented in this build")
        from scipy import special
        sgs = sorted(set(self.d_sg.values()))
        groups = [[i for i, c in enumerate(self.chrs) if self.d_sg[c] == sg] for sg in sgs]
        X = self.raw_data
        M = X.shape[0]
        means = np.stack([X[:, g].mean(axis=1) for g in groups], axis=1) if M else np.zeros((0, len(sgs)))
        # the reference orders groups by -sum/len (Cluster.py:182); ties keep SG-name order (stable)
        keyv = np.stack([-(X[:, g].sum(axis=1) / len(g)) for g in groups], axis=1) if M else means
        order = np.argsort(keyv, axis=1, kind="stable")
        top, second = order[:, 0], (order[:, 1] if len(sgs) > 1 else order[:, 0])
        pvals = np.ones(M)
        for a in range(len(sgs)):
            for b in range(len(sgs)):
                if a == b:
                    continue
                sel = np.flatnonzero((top == a) & (second == b))
                if sel.size:
                    pvals[sel] = _ttest_ind(X[np.ix_(sel, groups[a])], X[np.ix_(sel, groups[b])], special)
        print("\t".join(["#kmer", "subgenome", "p_value", "ratios"]), file=fout)
        with np.errstate(invalid="ignore"):
            keep = np.flatnonzero(~(pvals > max_pval))      # `if pvalue > max_pval: continue` keeps NaN
        from .textio import write_chunks
        kkeys, ktop, kp, kmeans, k = self.keys[keep], top[keep], pvals[keep], means[keep], self.k

        def fmt(lo, hi):
            kmers = kmerlib.decode_many(kkeys[lo:hi], k)
            return "".join("\t".join([km, sgs[t], repr(p), ",".join(map(repr, mv))]) + "\n"
                           for km, t, p, mv in zip(kmers, ktop[lo:hi].tolist(), kp[lo:hi].tolist(), kmeans[lo:hi].tolist()))
        fout.flush() if hasattr(fout, "flush") else None
        write_chunks(fout, len(kkeys), fmt)
        canon = kmerlib.canonical(self.keys[keep], self.k)
        return KmerLabels(canon, top[keep].astype(np.uint8), sgs, self.k)


def _ttest_ind(a, b, special):
    """Row-wise scipy.stats.ttest_ind(a, b) (equal_var=True, two-sided) p-values."""
    n1, n2 = a.shape[1], b.shape[1]
    with np.errstate(all="ignore"):
        v1 = a.var(axis=1, ddof=1) if n1 > 1 else np.zeros(a.shape[0])
        v2 = b.var(axis=1, ddof=1) if n2 > 1 else np.zeros(a.shape[0])
        df = n1 + n2 - 2.0
        svar = ((n1 - 1) * v1 + (n2 - 1) * v2) / df if df > 0 else np.full(a.shape[0], np.nan)
        denom = np.sqrt(svar * (1.0 / n1 + 1.0 / n2))
        t = (a.mean(axis=1) - b.mean(axis=1)) / denom
        return 2.0 * special.stdtr(df, -np.abs(t))
