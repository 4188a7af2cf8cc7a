"""Per-window / per-feature subgenome enrichment.

Host-side mirror of the reference's subphaser/Stats.py (same function names,
arguments and output files); the Fisher right tails and the enrichment decision
run in the HIP kernel k6_enrich (subphaser_amd/csrc/sp_enrich.hip):
  fisher_test (Stats.py:14-31), enrich/_enrich (:140-168), Pvalues (:170-198),
  enrich_bin (:75-118), group_exchanges (:119-132), is_exchange (:133-138),
  enrich_ltr (:33-73), correct_pvals (:11-12).
"""
import re
from itertools import groupby

import numpy as np

from .runtime import get_context, logger

MAX_INT = 2147483647 // 10   # Stats.py:9


def correct_pvals(pvals, method="fdr_bh"):
    """Benjamini-Hochberg step-up, the arithmetic of
    statsmodels.stats.multitest.multipletests(pvals, method='fdr_bh')[1]."""
    if method != "fdr_bh":
        raise ValueError("only fdr_bh is implemented")
    p = np.asarray(pvals, np.float64)
    n = p.size
    if n == 0:
        return p.copy()
    order = np.argsort(p)
    ps = p[order]
    ecdf = np.arange(1, n + 1) / float(n)
    q = ps / ecdf
    q = np.minimum.accumulate(q[::-1])[::-1]
    q[q > 1] = 1
    out = np.empty(n, np.float64)
    out[order] = q
    return out


def fisher_test(each, total, ctx=None):
    """Right-tail Fisher p-value of every column of `each` against `total`,
    with the reference's margins (x22 quirk + MAX_INT clamps, Stats.py:20-25)."""
    assert len(each) == len(total)
    ctx = ctx or get_context()
    S = len(each)
    # a two-row table [each; total-each] has column sums == total, so sp_enrich's own totals match
    rest = [int(t) - int(e) for e, t in zip(each, total)]
    if min(rest) < 0:
        raise ValueError("each > total")
    counts = np.array([list(map(int, each)), rest], np.int64).reshape(2, S)
    with np.errstate(all="ignore"):
        pvals, _, _, _ = ctx.enrich(counts, 0.05, 0.5)
    return [float(x) for x in pvals[0]]


class Pvalue:
    """Result record of one row, same attributes as the reference's Pvalue after _enrich."""
    __slots__ = ("pval", "key", "idx", "sig", "counts", "pvals", "ratios", "ratio", "enrich", "rowname")


def enrich(matrix, colnames=None, rownames=None, ncpu=4, min_ratio=0.5, max_pval=0.05, ctx=None, results=None,
           **kargs):
    """Yield one Pvalue record per row (Stats.py:140-168).  results = (pvals, argmin, sig, ratios) when the
    device has already tested these rows (the fused stack -> enrich call of the pipeline)."""
    arr = np.asarray(matrix, np.int64)
    if arr.ndim != 2:
        arr = arr.reshape(len(matrix), -1)
    if colnames is not None and rownames is not None:
        assert arr.shape == (len(rownames), len(colnames)), "{} != {}".format(
            arr.shape, (len(rownames), len(colnames)))
    assert len(colnames) > 1     # Stats.py:172
    if results is not None:
        pvals, argmin, sig, ratios = results
    else:
        ctx = ctx or get_context()
        pvals, argmin, sig, ratios = ctx.enrich(arr, max_pval, min_ratio)
    S = arr.shape[1]
    for w in range(arr.shape[0]):
        r = Pvalue()
        m = int(argmin[w])
        r.idx, r.key, r.pval, r.sig = m, colnames[m], float(pvals[w, m]), bool(sig[w])
        r.counts = [int(x) for x in arr[w]]
        r.pvals = [float(x) for x in pvals[w]]
        r.ratios = ratios[w].copy()
        r.ratio = r.ratios[m]
        r.enrich = [0] * (S + 1)
        r.enrich[m if r.sig else -1] = 1
        r.rowname = rownames[w]
        yield r


def is_exchange(obs_sg, exp_sg):
    if not exp_sg or not obs_sg:
        return "none"
    return "no" if obs_sg == exp_sg else "yes"


def _fmt(x):
    """str() as the reference's `map(str, line)` renders each column."""
    if isinstance(x, (float, np.floating)):
        return repr(float(x))
    return str(x)


def enrich_bin(fout, fout2, d_sg, *args, **kargs):
    """Enrich by chromosome window; writes `.bin.enrich` (fout) and `.bin.group` (fout2)."""
    total = consistent = exchange = 0
    lines, pvalues = [], []
    for res in enrich(*args, **kargs):
        chrom, start, end = res.rowname
        key = res.key if res.sig else None
        potential_exchange = is_exchange(d_sg.get(chrom), key)
        lines.append([chrom, start, end, key, res.pval,
                      ",".join(map(str, res.counts)), ",".join(map(_fmt, res.ratios)),
                      ",".join(map(str, res.enrich)), ",".join(map(_fmt, res.pvals)), potential_exchange])
        pvalues.append(res.pval)
        total += 1
        exchange += potential_exchange == "yes"
        consistent += potential_exchange == "no"
    if total:
        logger.info("Consistent with subgenome assignment: {} ({:.2%}); potential exchange: {} ({:.2%})".format(
            consistent, consistent / total, exchange, exchange / total))
    qvals = correct_pvals(pvalues)
    fout.write("\t".join(["#chrom", "start", "end", "subgenome", "p_value", "counts", "ratios", "enrich",
                          "pvals", "potential_exchange", "p_corrected"]) + "\n")
    for line, q in zip(lines, qvals):
        line.append(q)
        fout.write("\t".join(map(_fmt, line)) + "\n")
    fout2.write("\t".join(["#chrom", "start", "end", "exchange_from", "exchange_to", "N_bins",
                           "potential_exchange"]) + "\n")
    for line in group_exchanges(lines, d_sg):
        fout2.write("\t".join(map(_fmt, line)) + "\n")
    return lines


def group_exchanges(lines, d_sg):
    """Runs of consecutive significant windows with the same call, per chromosome."""
    for chrom, items in groupby(lines, key=lambda x: x[0]):
        obs_sg = d_sg.get(chrom)
        items = sorted((l for l in items if l[3] is not None), key=lambda x: x[1])
        for sg, xlines in groupby(items, key=lambda x: x[3]):
            xlines = list(xlines)
            yield [chrom, xlines[0][1], xlines[-1][2], sg, obs_sg, len(xlines), is_exchange(obs_sg, sg)]


_FEAT_ID = re.compile(r"(\S+?):\d+\-\d+")


def enrich_ltr(fout, d_sg, matrix, colnames=None, rownames=None, ncpu=4, min_ratio=0.5, max_pval=0.05, ctx=None,
               **kargs):
    """Output LTR / custom-feature enrichments (`.ltr.enrich`, `.custom.enrich`; Stats.py:33-73).
    Array code throughout: feature sets have millions of rows (BASELINE config 5)."""
    from .textio import write_chunks
    arr = np.asarray(matrix, np.int64)
    if arr.ndim != 2:
        arr = arr.reshape(len(matrix), -1)
    if colnames is not None and rownames is not None:
        assert arr.shape == (len(rownames), len(colnames)), "{} != {}".format(arr.shape, (len(rownames), len(colnames)))
    assert len(colnames) > 1     # Stats.py:172
    ctx = ctx or get_context()
    n = arr.shape[0]
    if n:
        pvals, argmin, sig, _ = ctx.enrich(arr, max_pval, min_ratio)
        pmin = pvals[np.arange(n), argmin]
    else:
        pvals, argmin, sig, pmin = np.zeros((0, arr.shape[1])), np.zeros(0, np.int64), np.zeros(0, bool), np.zeros(0)
    sig = np.asarray(sig, bool)
    ids = [r[0] for r in rownames]            # `ltr, *_ = res.rowname`
    # the reference crashes (AttributeError) on ids that are not chrom:start-end
    # (Stats.py:42-43); the evident intent is "unknown chromosome"
    cache = {}
    exch = []
    for i, ltr in enumerate(ids):
        m = _FEAT_ID.match(ltr)
        chrom = m.groups()[0] if m else None
        obs = cache.get(chrom)
        if obs is None and chrom not in cache:
            obs = cache[chrom] = d_sg.get(chrom)
        exch.append(is_exchange(obs, colnames[argmin[i]] if sig[i] else None))
    exch_a = np.array(exch, dtype=object)
    total, exchange, consistent = n, int((exch_a == "yes").sum()), int((exch_a == "no").sum())
    if exchange > 0 and consistent > 0:
        logger.info("Consistent with subgenome assignment: {} ({:.2%}); potential exchange: {} ({:.2%})".format(
            consistent, consistent / total, exchange, exchange / total))
    qvals = correct_pvals(pmin)
    fout.write("\t".join(["#id", "subgenome", "p_value", "counts", "potential_exchange", "p_corrected"]) + "\n")
    sgcol = [colnames[m] if s_ else None for m, s_ in zip(argmin.tolist(), sig.tolist())]

    def fmt(lo, hi):
        return "".join("%s\t%s\t%s\t%s\t%s\t%s\n" % (ids[i], sgcol[i], repr(float(pmin[i])),
                                                   ",".join(map(str, arr[i].tolist())), exch[i], repr(float(qvals[i])))
                       for i in range(lo, hi))
    fout.flush() if hasattr(fout, "flush") else None
    write_chunks(fout, n, fmt)
    d_enriched = {ltr: sg for ltr, sg in zip(ids, sgcol) if sg}
    d_exchange = dict(zip(ids, exch))
    return d_enriched, d_exchange
