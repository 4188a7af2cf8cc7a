// sp_lib.hip -- single translation unit of libsubphaser_hip.so (gfx950 only).
// The .hip files are kept separate for reading; they are compiled as one unit so
// that kernels shared between them (scan, pack) need no relocatable device code.
#include "sp_ctx.hip"
#include "sp_count.hip"
#include "sp_count2.hip"
#include "sp_filter.hip"
#include "sp_map.hip"
#include "sp_sparse.hip"
#include "sp_sparse2.hip"
#include "sp_enrich.hip"
#include "sp_synth.hip"
#include "sp_fasta.hip"
#include "sp_text.hip"
