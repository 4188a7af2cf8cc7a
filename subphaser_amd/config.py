"""sg.config grammar, id maps and prefix handling.

Mirrors SGConfig / add_prefix / parse_idmap / check_duplicates of the reference
(subphaser/__main__.py:731-789; grammar documented in README.md:88-105):
one homoeologous set per line, whitespace-separated columns = subgenomes,
`,` joins several chromosomes into one unit, `new|old` renames, `#` comments.
"""
from collections import Counter, OrderedDict

from .runtime import logger


def add_prefix(val, prefix=None, sep="|"):
    # with a label prefix every `sep`-separated part gets the prefix and the parts are concatenated
    if not prefix:
        return val
    return "".join("{}{}".format(prefix, v) for v in val.split(sep) if v)


class SGConfig:
    def __init__(self, sgcfg, prefix=None, sep="|"):
        self.sgcfg = sgcfg
        self.prefix, self.sep = prefix, sep
        self.nsgs, self.chrs, self.sgs = [], [], []
        first_n = 0
        with open(sgcfg) as fh:
            for raw in fh:
                cols = raw.split("#")[0].strip().split()
                if not cols:
                    continue
                units = [[add_prefix(x, prefix, sep) for x in col.strip(",").split(",")] for col in cols]
                self.nsgs.append(len(units))
                if first_n == 0:
                    first_n = len(units)
                if len(units) != first_n:
                    logger.warning("Number of column is different in line %s: %d in this line but %d in previous line",
                                   cols, len(units), first_n)
                for unit in units:
                    self.chrs.extend(unit)
                self.sgs.append(units)
        self.nsg = max(self.nsgs) if self.nsgs else 0
        for c, n in Counter(self.chrs).items():
            if n > 1:
                logger.warning("Chromsome id %s repeat %d times", c, n)

    def __iter__(self):
        return iter(self.sgs)


def parse_idmap(mapfile=None):
    """`-target` file: old_id [new_id]; '#' starts a comment."""
    if not mapfile:
        return None
    d = OrderedDict()
    with open(mapfile) as fh:
        for line in fh:
            line = line.strip().split("#")[0]
            if not line:
                continue
            t = line.split()
            d[t[0]] = t[1] if len(t) > 1 else t[0].split("|")[-1]
    return d


def check_duplicates(lst):
    if lst is None:
        return
    dup = {v: c for v, c in Counter(lst).items() if c > 1}
    if dup:
        raise ValueError("Duplicates detected: {}".format(dup))


def sets_to_csr(sgs, labels):
    """sgs (list of sets -> list of units -> list of chromosome ids) -> CSR index arrays."""
    import numpy as np
    idx = {lab: i for i, lab in enumerate(labels)}
    set_off, unit_off, unit_chrom = [0], [0], []
    for sg in sgs:
        for chrs in sg:
            try:
                unit_chrom += [idx[c] for c in chrs]
            except KeyError as e:
                raise KeyError("chromosome {} of the sg config is not among the loaded chromosomes".format(e))
            unit_off.append(len(unit_chrom))
        set_off.append(len(unit_off) - 1)
    return (np.array(set_off, np.int32), np.array(unit_off, np.int32), np.array(unit_chrom, np.int32))
