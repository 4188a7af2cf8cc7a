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
    """stack short bins: returns (coords, counts) like the reference."""
    coords, counts = [], []
    for chrom, (starts, vals) in read_bin_counts(inBedCount).items():
        wins, summed = stack_bins(starts, vals, window_size)
        for w, row in zip(wins.tolist(), summed):
            start = int(w * window_size)
            coords.append((chrom, start, int(start + window_size)))
            counts.append(list(row))
    return coords, counts
