#!/usr/bin/env python3
"""Per-kernel timing of the counting engines on one synthetic chromosome (dev tool).
SUBPHASER_HIP_LIB selects a library variant."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from subphaser_amd import _native
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 667_000_000
engine = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 15
ctx = _native.Context(0)
d = ctx.dev_alloc(n)
ctx.synth_chrom(d, n, 3, 0, 0, 3, 0)
ctx.genome_reset(1)
ctx.genome_add_device(0, d, n)
def _count():       # (bound-experiment variants produce wrong partitions: their kernel times are still what is asked for)
    try:
        ctx.count(K, 3, engine)
    except ValueError as e:
        print("count failed (%s): kernel times only" % str(e)[:60])
_count()
ctx.prof_enable(True)
for _ in range(3):
    _count()
ctx.prof_enable(False)
rep = ctx.prof_report()
tot = sum(v["ms"] / v["calls"] for v in rep.values())
print(os.environ.get("SUBPHASER_HIP_LIB", "default"), {k: round(v["ms"] / v["calls"], 3) for k, v in rep.items()}, "total %.3f ms" % tot,
      "-> %.1f Gbases/s" % (n / tot / 1e6), "lengths", int(ctx.lengths()[0]) if not os.environ.get("K1_NOLEN") else -1)
