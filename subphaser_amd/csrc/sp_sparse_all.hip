// sp_sparse_all.hip -- the k = 16..32 engines as one translation unit (sp_sparse2.hip uses the scan
// helpers and the per-chromosome list bookkeeping of sp_sparse.hip).
#include "sp_sparse.hip"
#include "sp_sparse2.hip"
