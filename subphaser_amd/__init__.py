"""subphaser_amd -- MI355X-native drop-in for SubPhaser's k-mer hot path (modules 1-2).

Host-side mirror of the reference's function-level interface for the path
  k-mer counting -> chromosome x k-mer matrix + differential filter ->
  bin/window mapping of subgenome-specific k-mers -> per-window Fisher enrichment
on top of libsubphaser_hip.so (hand-written HIP for gfx950, C-ABI in
include/subphaser_hip.h).  See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"
REFERENCE_VERSION = "1.2.7"   # reference: subphaser/__version__.py
