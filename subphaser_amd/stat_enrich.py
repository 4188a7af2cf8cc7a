"""Summarise an enrichment table by annotation prefix x subgenome.

Same output as the reference's stand-alone subphaser/stat_enrich.py:4-37
(key = (id.split('-')[0], subgenome); per annotation: number of rows per
subgenome, then the element-wise sum of the count vectors).  Accepts both the
4-column legacy table and the 6-column `.ltr.enrich`/`.custom.enrich` files
written by stats.enrich_ltr (the reference script only takes 4 columns).
"""
import sys

import numpy as np


def summarize(in_tsv, out=sys.stdout):
    d_count, ids, sgs, width = {}, set(), set(), 0
    for line in open(in_tsv):
        if line.startswith("#"):
            continue
        t = line.strip().split()
        if len(t) < 4:
            continue
        fid, subgenome, counts = t[0], t[1], np.array(list(map(int, t[3].split(","))))
        width = max(width, len(counts))
        key = (fid.split("-")[0], subgenome)
        if key not in d_count:
            d_count[key] = [1, counts]
        else:
            d_count[key][0] += 1
            d_count[key][1] = d_count[key][1] + counts
        ids.add(key[0])
        sgs.add(key[1])
    for ann in sorted(ids):
        num, count = [], None
        for sg in sorted(sgs):
            # (the reference sizes this zero vector by the number of subgenome labels, which breaks
            #  as soon as a `None` row is present; the count vectors' own width is what is meant)
            n, c = d_count.get((ann, sg), (0, np.zeros(width, dtype=int)))
            num.append(n)
            count = c.copy() if count is None else count + c
        out.write("\t".join(map(str, [ann] + num + list(count))) + "\n")


def main():
    summarize(sys.argv[1])


if __name__ == "__main__":
    main()
