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

from . import _native
from . import kmer as kmerlib
from .runtime import logger
from .seqs import KmerLabels
from ._native import write_chunks


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


def relabel_by_chromosome_order(chrs, raw_labels):
    """Cluster ids renumbered 0, 1, 2 ... in the order in which they first occur when the chromosomes are
    walked in sorted-name order (what the reference's sort_subgenomes produces, Cluster.py:119-127), as
    array operations: the first position of every id in the name-sorted label vector ranks the ids."""
    raw = np.asarray(raw_labels)
    if raw.size != len(chrs):
        raise ValueError("{} labels for {} chromosomes".format(raw.size, len(chrs)))
    by_name = np.argsort(np.asarray(chrs, dtype=object), kind="stable")
    ids, first, inverse = np.unique(raw[by_name], return_index=True, return_inverse=True)
    rank_of_id = np.empty(ids.size, np.int64)
    rank_of_id[np.argsort(first, kind="stable")] = np.arange(ids.size)
    out = np.empty(raw.size, np.int64)
    out[by_name] = rank_of_id[inverse]
    return out


class Cluster:
    """Chromosome -> subgenome assignment (k-means on the Z-normalised matrix, or the `-sg_assigned` table),
    bootstrap support and the subgenome-specific k-mer test."""

    def __init__(self, data, n_clusters, sg_prefix="SG", sg_assigned={}, re_assign=True, bootstrap=False,
                 replicates=1000, jackknife=80, seed=None, **kargs):
        """data: path of a `.kmer.mat` file or a FilteredMatrix (jellyfish.filter result)."""
        if isinstance(data, str):
            self.chrs, kmers, self.raw_data = load_matrix(data)
            self.k = len(kmers[0]) if kmers else 0
            self.keys = kmerlib.encode_many(kmers)
        else:
            self.chrs, self.raw_data, self.keys, self.k = list(data.labels), data.freqs, data.keys, data.k
            # the integer matrix behind the frequencies: lets the k-mer test run on the device (sp_kmer_ttest)
            self._counts, self._lengths = getattr(data, "counts", None), getattr(data, "lengths", None)
            self._ctx = getattr(data, "ctx", None)
            self._counts_dev = getattr(data, "counts_dev", None)
        self.sg_prefix, self.seed = sg_prefix, seed
        self.n_clusters = len(set(sg_assigned.values())) if sg_assigned else n_clusters
        if sg_assigned:
            logger.info("Skip k-means clustering")
            given = [sg_assigned[c] for c in self.chrs]
            if re_assign:
                self._name(relabel_by_chromosome_order(self.chrs, given))
            else:
                self.d_sg, self.sg_names = sg_assigned, sorted(set(sg_assigned.values()))
                self.labels = relabel_by_chromosome_order(self.chrs, given)
        else:
            self._name(relabel_by_chromosome_order(self.chrs, self._kmeans(self.zscores()).labels_))
        self.d_bs = self.bootstrap(replicates, jackknife) if bootstrap and replicates and replicates > 0 \
            else {c: "NA" for c in self.chrs}

    # chromosomes x k-mers, every k-mer column Z-normalised (Cluster.py:24-26, 78-82)
    def zscores(self, freqs=None):
        x = (self.raw_data if freqs is None else freqs).T
        with np.errstate(all="ignore"):
            return (x - x.mean(axis=0)) / x.std(axis=0)

    normalize_data = staticmethod(lambda data, axis=0: (data - data.mean(axis=axis)) / data.std(axis=axis))

    def _kmeans(self, points):
        from sklearn.cluster import KMeans
        return KMeans(n_clusters=self.n_clusters, random_state=self.seed).fit(points)

    def _name(self, labels):
        width = len(str(self.n_clusters))
        self.labels = np.asarray(labels, np.int64)
        names = ["{}{:0>{}d}".format(self.sg_prefix, int(l) + 1, width) for l in self.labels]
        self.d_sg = OrderedDict(zip(self.chrs, names))
        self.sg_names = sorted(set(names))

    def bootstrap(self, replicates=1000, jackknife=80):
        """Support of every chromosome's assignment: share of `replicates` k-means runs, each on `replicates`
        k-mers drawn with replacement (the reference resamples n_samples=replicates and ignores the jackknife
        size it computes, Cluster.py:85-90), that give the chromosome the same renumbered cluster id."""
        logger.info("Performing bootstrap of {} replicates, with each replicate resampling {}% data "
                    "with replacement".format(replicates, jackknife))
        z = self.zscores()                                   # C x M
        rng = np.random.RandomState(self.seed)
        M = z.shape[1]
        agree = np.zeros(len(self.chrs), np.int64)
        for _ in range(int(replicates)):
            cols = rng.randint(0, M, size=int(replicates))
            rep = relabel_by_chromosome_order(self.chrs, self._kmeans(z[:, cols]).labels_)
            agree += rep == self.labels
        return {c: int(100 * a / replicates) for c, a in zip(self.chrs, agree.tolist())}

    def output_subgenomes(self, fout=sys.stdout):
        fout.write("#chrom\tsubgenome\tbootstrap\n")
        for c in sorted(self.d_sg, key=lambda x: self.d_sg[x]):      # stable: by subgenome, input order inside
            fout.write("{}\t{}\t{}\n".format(c, self.d_sg[c], self.d_bs[c]))

    def output_kmers(self, fout=sys.stdout, max_pval=0.05, ncpu=4, method="map", test_method="ttest_ind", defer=False):
        """Student t-test (pooled variance, two-sided) between the highest-mean and the
        second-highest-mean subgenome groups of every k-mer; keeps p <= max_pval.
        Returns KmerLabels (array form of the reference's d_ksg dict).
        defer=True: nothing is written; returns (KmerLabels, write) where write(fout) produces the same text later
        (the CLI runs it on a writer thread while the mapping stage uses the labels)."""
        if test_method not in TEST_METHODS:
            raise ValueError("test_method must be one of {}".format(TEST_METHODS))
        from scipy import special
        sgs = sorted(set(self.d_sg.values()))
        groups = [[i for i, c in enumerate(self.chrs) if self.d_sg[c] == sg] for sg in sgs]
        X = self.raw_data
        M = X.shape[0]
        ctx = getattr(self, "_ctx", None)
        # the kernel keeps a group's values in registers: at most TTEST_MAX_GROUP chromosomes per subgenome
        # (sp_kmer_ttest returns SP_EUNSUP beyond); scaffold-level runs with larger groups take the numpy code below
        if (test_method == "ttest_ind" and M and ctx is not None and hasattr(ctx, "kmer_ttest")
                and getattr(self, "_counts", None) is not None and self._lengths is not None
                and max(len(g) for g in groups) <= TTEST_MAX_GROUP):
            # device path: one thread per k-mer (csrc/sp_enrich.hip k7_ttest); the numpy code below is the same test
            # for matrices that only exist as a `.kmer.mat` file or behind a context without the kernel
            staged = getattr(self, "_counts_dev", None)      # rows already on the device (the CLI stages them early)
            try:
                top, second, pvals, means = ctx.kmer_ttest(staged if staged else self._counts, self._lengths, groups)
            finally:
                if staged and hasattr(ctx, "release_rows"):     # M x C x 4 bytes of HBM nobody reads again
                    ctx.release_rows()
                    self._counts_dev = None
            return self._write_kmers(fout, sgs, top, pvals, means, max_pval, defer)
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
                if sel.size and test_method == "ttest_ind":
                    pvals[sel] = _ttest_ind(X[np.ix_(sel, groups[a])], X[np.ix_(sel, groups[b])], special)
                elif sel.size:      # the other scipy tests the reference accepts (Cluster.py:178-194), row by row
                    pvals[sel] = _scipy_rows(test_method, X[np.ix_(sel, groups[a])], X[np.ix_(sel, groups[b])])
        return self._write_kmers(fout, sgs, top, pvals, means, max_pval, defer)

    def _write_kmers(self, fout, sgs, top, pvals, means, max_pval, defer=False):
        with np.errstate(invalid="ignore"):
            keep = np.flatnonzero(~(pvals > max_pval))      # `if pvalue > max_pval: continue` keeps NaN
        kkeys, ktop, kp, kmeans, k = self.keys[keep], top[keep], pvals[keep], means[keep], self.k

        def fmt(lo, hi):
            kmers = kmerlib.decode_many(kkeys[lo:hi], k)
            return "".join("\t".join([km, sgs[t], repr(p), ",".join(map(repr, mv))]) + "\n"
                           for km, t, p, mv in zip(kmers, ktop[lo:hi].tolist(), kp[lo:hi].tolist(), kmeans[lo:hi].tolist()))
        def write(fout):
            print("\t".join(["#kmer", "subgenome", "p_value", "ratios"]), file=fout)
            fout.flush() if hasattr(fout, "flush") else None
            if len(kkeys) and _native.text_sig_kmers(fout, kkeys, k, ktop, sgs, kp, kmeans):
                return      # formatted by threads of this process (write_chunks below serves non-file objects, in this process)
            write_chunks(fout, len(kkeys), fmt)

        canon = kmerlib.canonical(self.keys[keep], self.k)
        labels = KmerLabels(canon, top[keep].astype(np.uint8), sgs, self.k)
        if defer:
            return labels, write
        write(fout)
        return labels


TEST_METHODS = ("ttest_ind", "kruskal", "wilcoxon", "mannwhitneyu")
TTEST_MAX_GROUP = 64     # SP_TT_MAXG in csrc/sp_enrich.hip


def _kruskal_rows(a, b):
    """scipy.stats.kruskal(a[i], b[i]).pvalue for every row, in array form (same operations in the same order as
    scipy's: average ranks, tie correction 1 - sum(t^3 - t) / (N^3 - N), H = 12 / (N (N + 1)) * sum(R_j^2 / n_j)
    - 3 (N + 1), chi-square survival function with one degree of freedom); rows whose values are all identical --
    where scipy raises "All numbers are identical" -- get p = 1 like the row-wise wrapper below gives them."""
    from scipy import special, stats as st
    n1, n2 = a.shape[1], b.shape[1]
    N = float(n1 + n2)
    x = np.concatenate([a, b], axis=1)
    ranked = st.rankdata(x, axis=1)
    srt = np.sort(ranked, axis=1)
    first = np.concatenate([np.ones((x.shape[0], 1), bool), srt[:, 1:] != srt[:, :-1]], axis=1)
    # size of the tie group every element belongs to: distance between consecutive group starts
    idx = np.where(first, np.arange(x.shape[1])[None, :], 0)
    start = np.maximum.accumulate(idx, axis=1)
    nxt = np.where(first, np.arange(x.shape[1])[None, :], x.shape[1])
    end = np.minimum.accumulate(nxt[:, ::-1], axis=1)[:, ::-1]       # start of the NEXT group at or after this element
    end = np.concatenate([end[:, 1:], np.full((x.shape[0], 1), x.shape[1])], axis=1)
    end = np.where(first, end, 0)        # count every group once, at its first element
    cnt = np.where(first, (end - start).astype(np.float64), 0.0)
    ties = 1.0 - (cnt ** 3 - cnt).sum(axis=1) / (N ** 3 - N) if N >= 2 else np.ones(x.shape[0])
    ssbn = ranked[:, :n1].sum(axis=1) ** 2 / n1 + ranked[:, n1:].sum(axis=1) ** 2 / n2
    with np.errstate(all="ignore"):
        h = (12.0 / (N * (N + 1)) * ssbn - 3 * (N + 1)) / ties
        p = special.chdtrc(1, h)
    return np.where(ties == 0, 1.0, p)


def _wilcoxon_rows(a, b):
    """scipy.stats.wilcoxon(a[i], b[i]).pvalue for every row.  scipy (>= 1.13) picks the method per call: exact when the
    differences hold no ties and no zeros; with ties or zeros and at most 13 pairs an EXACT PERMUTATION TEST over all
    2^n sign assignments (3 ms per call: two hours for the 2.2 M rows of a wheat-like run); the normal approximation
    otherwise.  Here the rows are split the same way: scipy's own `axis` form for the exact and the asymptotic rows,
    and for the permutation rows the null distribution of r_plus = all subset sums of the ranks of |d| (zeros rank 0:
    they only duplicate subsets), one matrix product per block of rows; p = 2 min(#(null <= obs), #(null >= obs)) / 2^n
    with scipy's tolerance, clipped at 1.  Sums of half-integers are exact in floating point, so the p-values are
    scipy's bit for bit (tests/test_abi_and_host.py)."""
    from scipy import stats as st
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    M, n = a.shape
    out = np.ones(M)
    if M == 0 or n == 0:
        return out
    d = a - b
    nz = d != 0
    ad = np.where(nz, np.abs(d), np.inf)              # zeros are dropped ("wilcox"): rank them last, out of the way
    srt = np.sort(ad, axis=1)
    tied = ((srt[:, 1:] == srt[:, :-1]) & np.isfinite(srt[:, 1:])).any(axis=1)
    exact = ~tied & nz.all(axis=1) & (n <= 50)
    perm = ~exact & (n <= 13)
    asym = ~exact & ~perm
    if exact.any():
        out[exact] = st.wilcoxon(a[exact], b[exact], axis=1, method="exact")[1]
    if asym.any():
        out[asym] = st.wilcoxon(a[asym], b[asym], axis=1, method="asymptotic")[1]
    if perm.any():
        rows = np.flatnonzero(perm)
        ranks = np.where(nz[rows], st.rankdata(ad[rows], axis=1), 0.0)
        obs = np.where(d[rows] > 0, ranks, 0.0).sum(axis=1)
        pat = ((np.arange(1 << n)[:, None] >> np.arange(n)[None, :]) & 1).astype(np.float64)      # [2^n, n]
        gamma = np.abs(np.finfo(np.float64).eps * 100 * obs)
        step = max(1, (1 << 22) >> n)                 # ~32 MB of null values per block
        for lo in range(0, rows.size, step):
            sl = slice(lo, lo + step)
            null = ranks[sl] @ pat.T
            le = (null <= (obs[sl] + gamma[sl])[:, None]).sum(axis=1)
            ge = (null >= (obs[sl] - gamma[sl])[:, None]).sum(axis=1)
            out[rows[sl]] = np.clip(np.minimum(le, ge) / float(1 << n) * 2, 0, 1)
    return out


# scipy releases whose per-call method rules the array forms below restate (wilcoxon: exact for n <= 50 without ties
# or zeros, permutation for n <= 13, else asymptotic -- scipy 1.13; mannwhitneyu: exact unless both samples exceed 8
# values or any value is tied) and whose `axis=` / `method=` keywords they rely on.  Checked against the row-wise
# calls of 1.15 (tests/test_abi_and_host.py); any other release takes the row-wise loop (advisor r04).
_SCIPY_ARRAY_RANGE = ((1, 13), (1, 17))


def _scipy_array_ok(version=None):
    if version is None:
        import scipy
        version = scipy.__version__
    try:
        v = tuple(int(x) for x in version.split(".")[:2])
    except ValueError:
        return False
    return _SCIPY_ARRAY_RANGE[0] <= v < _SCIPY_ARRAY_RANGE[1]


def _scipy_rows(name, a, b):
    """p-value of scipy.stats.<name>(a[i], b[i]) for every row (the reference calls the test per k-mer).
    mannwhitneyu: scipy's own axis argument, the rows split by the method a row-wise call picks (millions of rows in seconds); kruskal: the array form
    above (equal to the row-wise call, tests/test_abi_and_host.py); wilcoxon: _wilcoxon_rows; groups of unequal size
    (scipy raises for every row) fall through to the row-wise loop, which reports scipy's error."""
    from scipy import stats as st
    if not _scipy_array_ok():
        name_array = None        # an unvalidated scipy release: its own row-wise calls decide (slow, but its answers)
    else:
        name_array = name
    if a.shape[0] and name_array == "mannwhitneyu":
        # method="auto" decides per CALL: exact unless both samples exceed 8 values or ANY value is tied -- so the rows
        # are split by what a row-wise call would have chosen for each of them
        n1, n2 = a.shape[1], b.shape[1]
        xs = np.sort(np.concatenate([a, b], axis=1), axis=1)
        tied = (xs[:, 1:] == xs[:, :-1]).any(axis=1)
        exact = ~tied if not (n1 > 8 and n2 > 8) else np.zeros(a.shape[0], bool)
        out = np.empty(a.shape[0])
        if exact.any():
            out[exact] = st.mannwhitneyu(a[exact], b[exact], axis=1, method="exact")[1]
        if (~exact).any():
            out[~exact] = st.mannwhitneyu(a[~exact], b[~exact], axis=1, method="asymptotic")[1]
        return out
    if a.shape[0] and name_array == "kruskal":
        return _kruskal_rows(np.asarray(a, np.float64), np.asarray(b, np.float64))
    if a.shape[0] and name_array == "wilcoxon" and a.shape[1] == b.shape[1] and a.shape[1] >= 2:      # (one pair: scipy's permutation branch raises)
        return _wilcoxon_rows(a, b)
    test = getattr(st, name)
    out = np.empty(a.shape[0])
    for i in range(a.shape[0]):
        try:
            out[i] = test(a[i], b[i])[1]
        except ValueError as e:
            if "identical" in str(e) or "zero" in str(e):     # all values equal: no evidence of a difference
                out[i] = 1.0
            else:
                raise
    return out


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
