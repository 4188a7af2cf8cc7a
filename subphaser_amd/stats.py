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
# one pass of the regex engine over all ids joined by line breaks: group 1 of every line, '' where the id is not
# chrom:start-end (a per-row .match() is a microsecond of interpreter per feature, seconds at 2 x 10^6 features)
_FEAT_LINES = re.compile(r"^(?:(\S+?):\d+\-\d+)?.*$", re.M)
_EXCH = ("none", "no", "yes")


def feature_chroms(ids):
    """chromosome part of every feature id (`ltr.split(':')`-like, Stats.py:42-43), '' where it has none"""
    if not ids:
        return []
    if any("\n" in i for i in ids[:1]):       # ids are FASTA tokens: no blanks; guard the join below anyway
        return [(_FEAT_ID.match(i).groups()[0] if _FEAT_ID.match(i) else "") for i in ids]
    got = _FEAT_LINES.findall("\n".join(ids))
    if len(got) != len(ids):                   # an id containing a line break
        return [(_FEAT_ID.match(i).groups()[0] if _FEAT_ID.match(i) else "") for i in ids]
    return got


def enrich_ltr(fout, d_sg, matrix, colnames=None, rownames=None, ncpu=4, min_ratio=0.5, max_pval=0.05, ctx=None,
               row_chroms=None, as_arrays=False, **kargs):
    """Output LTR / custom-feature enrichments (`.ltr.enrich`, `.custom.enrich`; Stats.py:33-73).
    Array code throughout: feature sets have millions of rows (BASELINE config 5); the rows are formatted by the
    library's threaded writer when fout is a real file.  row_chroms: the chromosome of every row when the caller
    already has it, else it is parsed from the id (`chrom:start-end`).  rownames may be a seqs.IntervalRows (BED
    intervals): the ids are then written straight from its arrays.  as_arrays: return (subgenome index or -1, exchange
    code) arrays instead of the two {id: ...} dicts."""
    from . import _native
    from ._native import write_chunks
    arr = np.ascontiguousarray(matrix, np.int64)
    if arr.ndim != 2:
        arr = arr.reshape(len(matrix), -1)
    if colnames is not None and rownames is not None:
        assert arr.shape == (len(rownames), len(colnames)), "{} != {}".format(arr.shape, (len(rownames), len(colnames)))
    assert len(colnames) > 1     # Stats.py:172
    colnames = list(colnames)
    ctx = ctx or get_context()
    n = arr.shape[0]
    if n:
        pvals, argmin, sig, _ = ctx.enrich(arr, max_pval, min_ratio)
        argmin = np.asarray(argmin, np.int64)
        pmin = pvals[np.arange(n), argmin]
    else:
        pvals, argmin, sig, pmin = np.zeros((0, arr.shape[1])), np.zeros(0, np.int64), np.zeros(0, bool), np.zeros(0)
    sig = np.asarray(sig, bool)
    col_of = {name: j for j, name in enumerate(colnames)}

    def obs_code(c):
        sg = d_sg.get(c) if c else None
        return col_of.get(sg, len(colnames)) if sg else -1    # -1: unknown chromosome; len: a name outside colnames
    ival = rownames if hasattr(rownames, "column") else None      # seqs.IntervalRows: ids stay arrays
    if ival is not None:
        ids = None
        per_name = np.array([obs_code(c) for c in ival.names], np.int64)
        obs = per_name[ival.code] if n else np.zeros(0, np.int64)
    else:
        ids = [r[0] for r in rownames] if n and not isinstance(rownames[0], str) else list(rownames)   # `ltr, *_ = res.rowname`
        # the reference crashes (AttributeError) on ids that are not chrom:start-end
        # (Stats.py:42-43); the evident intent is "unknown chromosome"
        chroms = list(row_chroms) if row_chroms is not None else feature_chroms(ids)
        obs_of = {c: obs_code(c) for c in set(chroms)}
        obs = np.fromiter((obs_of[c] for c in chroms), np.int64, n)
    exp = np.where(sig, argmin, -1)
    # is_exchange(obs, exp): "none" when either side is missing, else "no" / "yes"
    exch_code = np.where((obs < 0) | (exp < 0), 0, np.where(obs == exp, 1, 2)).astype(np.int32)
    total, exchange, consistent = n, int((exch_code == 2).sum()), int((exch_code == 1).sum())
    if exchange > 0 and consistent > 0:
        logger.info("Consistent with subgenome assignment: {} ({:.2%}); potential exchange: {} ({:.2%})".format(
            consistent, consistent / total, exchange, exchange / total))
    qvals = correct_pvals(pmin)
    fout.write("\t".join(["#id", "subgenome", "p_value", "counts", "potential_exchange", "p_corrected"]) + "\n")
    sg_idx = np.where(sig, argmin, len(colnames)).astype(np.int32)     # last name: str(None)
    sg_names = colnames + ["None"]
    done = False
    if n:
        if ival is not None:
            idcol = ival.column()
        else:
            blob, off = _native.str_blob(ids)
            idcol = ("str", blob, off)
        done = _native.text_table(fout, n, [idcol, ("name", sg_idx, sg_names), ("f64", pmin, ","),
                                            ("i64", arr, ","), ("name", exch_code, list(_EXCH)), ("f64", qvals, ",")])
    if not done:
        if ids is None:
            ids = ival.ids()

        def fmt(lo, hi):
            return "".join("%s\t%s\t%s\t%s\t%s\t%s\n" % (ids[i], sg_names[sg_idx[i]], repr(float(pmin[i])),
                                                       ",".join(map(str, arr[i].tolist())), _EXCH[exch_code[i]],
                                                       repr(float(qvals[i])))
                           for i in range(lo, hi))
        write_chunks(fout, n, fmt)
    if as_arrays:      # millions of rows: (subgenome index or -1, exchange code 0 none / 1 no / 2 yes) instead of two dicts
        return np.where(sig, argmin, -1), exch_code
    if ids is None:
        ids = ival.ids()
    hit = np.flatnonzero(sig)
    d_enriched = {ids[i]: colnames[j] for i, j in zip(hit.tolist(), argmin[hit].tolist())}
    d_exchange = dict(zip(ids, map(_EXCH.__getitem__, exch_code.tolist())))
    return d_enriched, d_exchange
