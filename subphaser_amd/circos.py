"""Stack 10-kb bin counts into windows.

Mirror of Circos.stack_matrix / _bed_density(stack=True) of the reference
(subphaser/Circos.py:709-742, 831-842): window = int(START // window_size),
coords = (chrom, w*ws, w*ws+ws) with the end NOT clipped, rows in order of
first appearance, duplicate bin lines (10-Mb chunk boundaries) summed.
"""
from collections import OrderedDict

import numpy as np


def _open(path):
    if str(path).endswith(".gz"):
        import gzip
        return gzip.open(path, "rt")
    return open(path)


def read_bin_counts(inBedCount):
    """Parse a `.bin.count` file -> OrderedDict chrom -> (starts int64 [n], counts int64 [n,S])."""
    per = OrderedDict()
    with _open(inBedCount) as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            t = line.split()
            if len(t) < 3:
                continue
            try:
                start = int(t[1])
                int(t[2])
                vals = [int(x) for x in t[3:]]
            except ValueError:
                continue
            d = per.setdefault(t[0], ([], []))
            d[0].append(start)
            d[1].append(vals)
    return OrderedDict((c, (np.array(s, np.int64), np.array(v, np.int64).reshape(len(s), -1)))
                       for c, (s, v) in per.items())


def read_bin_counts_arrays(inBedCount):
    """Array form of a `.bin.count` file: (names, code[n], starts[n], counts[n, S]); names in
    first-appearance order, code[i] indexes names.  Fast path through the pandas C parser (files with
    millions of feature rows); any irregular file goes through read_bin_counts."""
    try:
        import pandas as pd
        df = pd.read_csv(inBedCount, sep="\t", header=None, dtype=str, keep_default_na=False, quoting=3,
                         engine="c", compression="infer")
        if df.shape[1] < 4:
            raise ValueError("too few columns")
        first = df[0].to_numpy()
        keepm = np.array([not x.startswith("#") for x in first.tolist()], bool) if len(first) else np.zeros(0, bool)
        df = df[keepm]
        if (df[0].str.contains(r"\s").any()):
            raise ValueError("whitespace inside a field")
        starts = df[1].to_numpy().astype(np.int64)
        df[2].to_numpy().astype(np.int64)                     # `end` must parse, like the reference's int()
        counts = df.iloc[:, 3:].to_numpy().astype(np.int64)
        codes, names = pd.factorize(df[0].to_numpy(), sort=False)
        return list(names), codes.astype(np.int64), starts, counts.reshape(len(starts), -1)
    except Exception:
        per = read_bin_counts(inBedCount)
        names = list(per)
        code = np.concatenate([np.full(len(per[c][0]), i, np.int64) for i, c in enumerate(names)]) if names else np.zeros(0, np.int64)
        starts = np.concatenate([per[c][0] for c in names]) if names else np.zeros(0, np.int64)
        counts = np.concatenate([per[c][1] for c in names], axis=0) if names else np.zeros((0, 0), np.int64)
        return names, code, starts, counts


def stack_bins(starts, counts, window_size):
    """One chromosome: (window indices in first-appearance order, summed counts)."""
    window_size = int(window_size) if float(window_size).is_integer() else window_size
    win = (np.asarray(starts) // window_size).astype(np.int64)
    uniq, first, inv = np.unique(win, return_index=True, return_inverse=True)
    summed = np.zeros((uniq.size, counts.shape[1]), np.int64)
    np.add.at(summed, inv, counts)
    order = np.argsort(first, kind="stable")
    return uniq[order], summed[order]


def stack_matrix(inBedCount, window_size=100000):
    """stack short bins: returns (coords, counts) like the reference (Circos.py:831-842).
    Rows: chromosomes in first-appearance order, windows of a chromosome in first-appearance order."""
    return stack_arrays(*read_bin_counts_arrays(inBedCount), window_size=window_size)


def factorize_first(ids):
    """(names in first-appearance order, code[i] into names) of a list of str."""
    try:
        import pandas as pd
        codes, names = pd.factorize(np.asarray(ids, dtype=object), sort=False)
        return list(names), codes.astype(np.int64)
    except ImportError:
        seen = {}
        code = np.fromiter((seen.setdefault(x, len(seen)) for x in ids), np.int64, len(ids))
        return list(seen), code


def stack_arrays(names, code, starts, vals, window_size=100000):
    """stack_matrix on the parsed form of a `.bin.count` file (names, code[n], starts[n], counts[n, S]): what the
    CLI hands over when it already holds the lines as arrays (no text round trip)."""
    starts, vals = np.asarray(starts, np.int64), np.asarray(vals, np.int64)
    if len(starts) == 0:
        return [], []
    ws = int(window_size) if float(window_size).is_integer() else window_size
    win = (starts // ws).astype(np.int64)
    order = np.lexsort((win, code))                           # stable: ties keep file order
    c_s, w_s = code[order], win[order]
    newg = np.ones(len(order), bool)
    newg[1:] = (c_s[1:] != c_s[:-1]) | (w_s[1:] != w_s[:-1])
    gstart = np.flatnonzero(newg)
    summed = np.add.reduceat(vals[order], gstart, axis=0)
    first = np.minimum.reduceat(order, gstart)                # first line of each (chromosome, window)
    g_code, g_win = c_s[gstart], w_s[gstart]
    o2 = np.lexsort((first, g_code))                          # chromosome order, then first appearance
    g_code, g_win, summed = g_code[o2], g_win[o2], summed[o2]
    st = g_win * ws
    name_arr = np.array(names, dtype=object)[g_code]
    if isinstance(ws, int):
        coords = list(zip(name_arr.tolist(), st.tolist(), (st + ws).tolist()))
    else:
        coords = [(n_, int(s_), int(s_ + ws)) for n_, s_ in zip(name_arr.tolist(), st.tolist())]
    return coords, summed.tolist()


def abnormal(data, k=1.5, high_tile=99, low_tile=1):
    """Upper / lower trimming cut-offs (Circos.py:973-980: the 99th / 1st percentiles)."""
    return np.percentile(data, high_tile), np.percentile(data, low_tile)


def stack_bed_density(inBedCount, outpre, colnames, window_size=100000, trim=True):
    """Per-subgenome circos histogram tracks `<outpre>.<SG>.txt` from a `.bin.count` file
    (Circos.py:777-806): windows from stack_matrix, counts capped at the column's 99th percentile,
    lines `chrom start end value` separated by blanks.  Returns {SG: path}."""
    import sys
    coords, counts = stack_matrix(inBedCount, window_size=window_size)
    colnames = list(colnames)
    arr = np.asarray(counts, np.int64).reshape(len(coords), -1) if len(coords) else np.zeros((0, len(colnames)), np.int64)
    assert arr.shape[0] == 0 or arr.shape[1] == len(colnames), "{} != {}".format(len(colnames), arr.shape[1])
    d_outfiles = {key: "{}.{}.txt".format(outpre, key) for key in colnames}
    d_upper = {}
    if trim:
        for j, key in enumerate(colnames):
            upper, _ = abnormal(arr[:, j])            # the reference fails on an empty file as well
            d_upper[key] = upper
            print("using cutoff: upper {} for {}".format(upper, key), file=sys.stderr)
    for j, key in enumerate(colnames):
        with open(d_outfiles[key], "w") as fout:
            col = arr[:, j].tolist()
            if trim:
                up = d_upper[key]
                col = [min(c, up) for c in col]       # min(int, np.float64) keeps the reference's typing
            fout.write("".join("{} {} {} {}\n".format(c[0], c[1], c[2], v) for c, v in zip(coords, col)))
    return d_outfiles


def out_sg_lines(sg_lines, datadir, ratio_col=6, enrich_col=7):
    """`sg_ratio.txt` / `sg_enrich.txt` for circos from the rows enrich_bin returns (Circos.py:619-634)."""
    ratio_file = "{}/sg_ratio.txt".format(datadir)
    enrich_file = "{}/sg_enrich.txt".format(datadir)
    with open(ratio_file, "w") as fr, open(enrich_file, "w") as fe:
        for line in sg_lines:
            chrom, start, end = line[:3]
            for f, dat in zip((fr, fe), (line[ratio_col], line[enrich_col])):
                f.write("\t".join(map(str, [chrom, start, end, dat])) + "\n")
    return ratio_file, enrich_file
