// sp_sparse.hip -- the k = 16..32 engine ("sparse": 64-bit keys, sorted (key, count) arrays
// instead of dense tables).  BASELINE.json config 5 sweeps k = 17 / 21.
//
//   count : 64-bit scan -> canonical key per start position (sentinel where the window is broken)
//           -> radix sort -> run-length encode -> keep count >= lower_count   (per chromosome)
//   filter: concatenate the C sorted arrays as (key, chrom|count) pairs -> sort by key ->
//           one thread per run of equal keys rebuilds the row and applies the same
//           sp_filter_decide() as the dense engine -> ordered compaction (ascending key)
//   map   : labelled k-mers in an open-addressing hash table (+ the L2-resident pre-filter),
//           same binning code path as the dense engine
//
// The sort and the run-length encode are rocPRIM device primitives (plain library ops on this
// secondary path); everything specific to the problem is hand-written.  k <= 15 never comes here.
#include <algorithm>
#include <cstring>
#include <utility>
#include <rocprim/rocprim.hpp>

#include "sp_device.h"
#include "sp_filter.h"
#include "sp_map.h"

#define SPS_SENTINEL (~0ULL)
#define SPS_MAXC 64

__device__ __forceinline__ uint64_t sps_mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// ------------------------------------------------------------------ count
__global__ void __launch_bounds__(256)
sps_keygen(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, int64_t n_units, sp_kparams kp,
           unsigned long long *__restrict__ keys /* pre-filled with the sentinel */,
           unsigned long long *__restrict__ n_valid) {
    __shared__ unsigned long long red[16];
    unsigned long long nv = 0;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units;
         u += (int64_t)gridDim.x * blockDim.x) {
        sp_scan_unit64<SP_UNIT>(pk, nm, u * SP_UNIT, kp, [&](int64_t start, uint64_t fwd, uint64_t rc) {
            keys[start] = fwd < rc ? fwd : rc;
            nv++;
        });
    }
    unsigned long long t = sp_block_sum_u64(nv, red);
    if (threadIdx.x == 0 && t) atomicAdd(n_valid, t);
}

#define SEL_PER_THREAD 16
#define SEL_BLOCK 256
#define SEL_SPAN (SEL_PER_THREAD * SEL_BLOCK)

// runs with count >= lower: per-block tally (+ sum of those counts = `lengths`)
__global__ void __launch_bounds__(SEL_BLOCK)
sps_sel_count(const uint32_t *__restrict__ counts, int64_t n, uint32_t lower, unsigned long long *__restrict__ blk,
              unsigned long long *__restrict__ sum_out) {
    __shared__ unsigned long long red[16];
    const int64_t base = (int64_t)blockIdx.x * SEL_SPAN;
    unsigned long long c = 0, s = 0;
    for (int j = 0; j < SEL_PER_THREAD; j++) {
        int64_t i = base + (int64_t)j * SEL_BLOCK + threadIdx.x;
        if (i < n && counts[i] >= lower) {
            c++;
            s += counts[i];
        }
    }
    unsigned long long tc = sp_block_sum_u64(c, red);
    unsigned long long ts = sp_block_sum_u64(s, red);
    if (threadIdx.x == 0) {
        blk[blockIdx.x] = tc;
        if (ts) atomicAdd(sum_out, ts);
    }
}

__global__ void __launch_bounds__(SEL_BLOCK)
sps_sel_write(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ counts, int64_t n,
              uint32_t lower, const unsigned long long *__restrict__ blk, unsigned long long *__restrict__ out_keys,
              uint32_t *__restrict__ out_counts) {
    __shared__ uint32_t lds[16];
    const int64_t base = (int64_t)blockIdx.x * SEL_SPAN;
    unsigned long long off = blk[blockIdx.x];
    for (int j = 0; j < SEL_PER_THREAD; j++) {
        int64_t i = base + (int64_t)j * SEL_BLOCK + threadIdx.x;
        uint32_t c = (i < n) ? counts[i] : 0u;
        bool p = (i < n) && c >= lower;
        uint32_t tot;
        uint32_t my = sp_block_excl_count(p, lds, tot);
        if (p) {
            out_keys[off + my] = keys[i];
            out_counts[off + my] = c;
        }
        off += tot;
    }
}

// scan kernel of sp_count.hip
__global__ void scan_excl_u64(unsigned long long *a, int64_t n, unsigned long long *total);

// ------------------------------------------------------------------ filter
__global__ void __launch_bounds__(256)
sps_concat(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ counts, int64_t n, int chrom,
           unsigned long long *__restrict__ out_keys, unsigned long long *__restrict__ out_vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_keys[i] = keys[i];
    out_vals[i] = ((unsigned long long)chrom << 32) | counts[i];
}

struct sps_filter_args {
    int C;
    sp_fsets F;
};

// flags per entry: bit0 = differential row, bit1 = fold-passing (hist), bit2 = head of a run (union)
__global__ void __launch_bounds__(SEL_BLOCK)
sps_eval(const unsigned long long *__restrict__ K, const unsigned long long *__restrict__ V, int64_t n,
         sps_filter_args A, uint8_t *__restrict__ flags, unsigned long long *__restrict__ blk_row,
         unsigned long long *__restrict__ blk_hist, unsigned long long *__restrict__ n_union) {
    __shared__ unsigned long long red[16];
    const int64_t base = (int64_t)blockIdx.x * SEL_SPAN;
    unsigned long long nrow = 0, nhist = 0, nuni = 0;
    for (int j = 0; j < SEL_PER_THREAD; j++) {
        const int64_t i = base + (int64_t)j * SEL_BLOCK + threadIdx.x;
        if (i >= n) continue;
        uint8_t fl = 0;
        const unsigned long long key = K[i];
        if (i == 0 || K[i - 1] != key) {
            uint32_t row[SPS_MAXC];
            for (int c = 0; c < A.C; c++) row[c] = 0;
            unsigned long long tot = 0;
            for (int64_t q = i; q < n && K[q] == key; q++) {
                const unsigned long long v = V[q];
                row[(int)(v >> 32)] = (uint32_t)v;
                tot += (uint32_t)v;
            }
            bool is_row, is_hist;
            sp_filter_decide([&](int c) -> uint32_t { return row[c]; }, tot, A.F, is_row, is_hist);
            fl = 4 | (is_row ? 1 : 0) | (is_hist ? 2 : 0);
            nuni++;
            nrow += is_row;
            nhist += is_hist;
        }
        flags[i] = fl;
    }
    unsigned long long t_row = sp_block_sum_u64(nrow, red);
    unsigned long long t_hist = sp_block_sum_u64(nhist, red);
    unsigned long long t_uni = sp_block_sum_u64(nuni, red);
    if (threadIdx.x == 0) {
        blk_row[blockIdx.x] = t_row;
        blk_hist[blockIdx.x] = t_hist;
        if (t_uni) atomicAdd(n_union, t_uni);
    }
}

__global__ void __launch_bounds__(SEL_BLOCK)
sps_emit(const unsigned long long *__restrict__ K, const unsigned long long *__restrict__ V, int64_t n, int C,
         const uint8_t *__restrict__ flags, uint8_t bit, const unsigned long long *__restrict__ blk,
         unsigned long long *__restrict__ out_keys, uint32_t *__restrict__ out_counts,
         unsigned long long *__restrict__ out_tot) {
    __shared__ uint32_t lds[16];
    const int64_t base = (int64_t)blockIdx.x * SEL_SPAN;
    unsigned long long off = blk[blockIdx.x];
    for (int j = 0; j < SEL_PER_THREAD; j++) {
        const int64_t i = base + (int64_t)j * SEL_BLOCK + threadIdx.x;
        const bool p = (i < n) && (flags[i] & bit);
        uint32_t tot_blk;
        const uint32_t my = sp_block_excl_count(p, lds, tot_blk);
        if (p) {
            const unsigned long long r = off + my, key = K[i];
            unsigned long long tot = 0;
            if (out_counts)
                for (int c = 0; c < C; c++) out_counts[r * C + c] = 0;
            for (int64_t q = i; q < n && K[q] == key; q++) {
                const unsigned long long v = V[q];
                if (out_counts) out_counts[r * C + (int)(v >> 32)] = (uint32_t)v;
                tot += (uint32_t)v;
            }
            if (out_keys) out_keys[r] = key;
            if (out_tot) out_tot[r] = tot;
        }
        off += tot_blk;
    }
}

// ------------------------------------------------------------------ labels + map
// Open-addressing table of 16-byte entries {key, label}: a hit costs ONE cache line (key and label
// used to live in two arrays = two L2 misses per mapped position).  label: 0 none, 1+sg, bit 7 = seen.
__global__ void __launch_bounds__(256)
sps_hash_insert(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ sg, int64_t n,
                unsigned long long *__restrict__ htab, uint64_t mask) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = keys[i];
    uint64_t h = sps_mix(key) & mask;
    for (;;) {
        unsigned long long prev = atomicCAS(&htab[2 * h], SPS_SENTINEL, key);
        if (prev == SPS_SENTINEL || prev == key) break;
        h = (h + 1) & mask;
    }
    htab[2 * h + 1] = (unsigned long long)(1u + sg[i]);
}

__device__ __forceinline__ int sps_lookup(uint64_t key, unsigned long long *__restrict__ htab, uint64_t mask) {
    uint64_t h = sps_mix(key) & mask;
    for (;;) {
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(htab + 2 * h);
        if (e.x == key) {
            const uint32_t l = (uint32_t)e.y;
            if (!(l & 0x80u)) htab[2 * h + 1] = (unsigned long long)(l | 0x80u);   // idempotent "seen" mark
            return (int)(l & 0x7fu) - 1;
        }
        if (e.x == SPS_SENTINEL) return -1;
        h = (h + 1) & mask;
    }
}

// ------------------------------------------------------------------ pair-keyed label table (k > 15, <= 7 subgenomes)
// The k <= 15 pair table, hashed: an entry is keyed by the canonical (k-1)-mer x that two neighbouring starts share
// and its payload holds the same eight 4-bit fields (label of b + x and of x + b for the four bases b, seen flag in
// bit 3), so ONE 16-byte look-up answers BOTH starts of a candidate pair -- the per-k-mer table above costs one L2
// miss per candidate start, and those misses are what k5_map_sparse waits for.
struct sps_pair_loc {
    uint64_t idx;   // canonical (k-1)-mer
    int field;
};
__host__ __device__ __forceinline__ sps_pair_loc sps_loc_prefix(uint64_t o, int k) {     // o = x + b
    sps_pair_loc r;
    const uint32_t b = (uint32_t)(o & 3ULL);
    const uint64_t p = o >> 2, pc = sp_revcomp(p, k - 1);
    if (p <= pc) { r.idx = p; r.field = 4 + (int)b; }
    else { r.idx = pc; r.field = 3 - (int)b; }              // rc: comp(b) + rc(x)
    return r;
}
__host__ __device__ __forceinline__ sps_pair_loc sps_loc_suffix(uint64_t o, int k) {     // o = b + x
    sps_pair_loc r;
    const uint32_t b = (uint32_t)(o >> (2 * (k - 1))) & 3u;
    const uint64_t m1mask = (1ULL << (2 * (k - 1))) - 1ULL;
    const uint64_t x = o & m1mask, xc = sp_revcomp(x, k - 1);
    if (x <= xc) { r.idx = x; r.field = (int)b; }
    else { r.idx = xc; r.field = 7 - (int)b; }              // rc: rc(x) + comp(b)
    return r;
}
__global__ void __launch_bounds__(256)
sps_pair_init(unsigned long long *__restrict__ htab, int64_t cap) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
        htab[2 * i] = SPS_SENTINEL;
        htab[2 * i + 1] = 0ULL;
    }
}
__device__ __forceinline__ void sps_pair_put(unsigned long long *__restrict__ htab, uint64_t mask, sps_pair_loc a, uint32_t l) {
    uint64_t h = sps_mix(a.idx) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&htab[2 * h], SPS_SENTINEL, (unsigned long long)a.idx);
        if (prev == SPS_SENTINEL || prev == a.idx) break;
        h = (h + 1) & mask;
    }
    atomicOr(&htab[2 * h + 1], (unsigned long long)l << (4 * a.field));
}
__global__ void __launch_bounds__(256)
sps_pair_insert(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ sg, int64_t n, int k,
                unsigned long long *__restrict__ htab, uint64_t mask) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i], r = sp_revcomp(key, k);
    const uint32_t l = 1u + sg[i];
    sps_pair_put(htab, mask, sps_loc_prefix(key, k), l);
    sps_pair_put(htab, mask, sps_loc_suffix(key, k), l);
    sps_pair_put(htab, mask, sps_loc_prefix(r, k), l);      // (the same two entries and fields; kept for symmetry
    sps_pair_put(htab, mask, sps_loc_suffix(r, k), l);      //  with k4_pair_table: the OR is idempotent)
}
// entry of the canonical (k-1)-mer `x`: payload (0 when absent) and its slot
__device__ __forceinline__ uint32_t sps_pair_get(uint64_t x, const unsigned long long *__restrict__ htab, uint64_t mask,
                                                 uint64_t &slot) {
    uint64_t h = sps_mix(x) & mask;
    for (;;) {
        const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(htab + 2 * h);
        if (e.x == x) {
            slot = h;
            return (uint32_t)e.y;
        }
        if (e.x == SPS_SENTINEL) return 0u;
        h = (h + 1) & mask;
    }
}
__global__ void __launch_bounds__(256)
sps_pair_seen(const unsigned long long *__restrict__ keys, int64_t n, int k, const unsigned long long *__restrict__ htab,
              uint64_t mask, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[16];
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i], r = sp_revcomp(key, k);
        const sps_pair_loc L[4] = {sps_loc_prefix(key, k), sps_loc_suffix(key, k), sps_loc_prefix(r, k), sps_loc_suffix(r, k)};
        uint32_t seen = 0;
        for (int q = 0; q < 4; q++) {
            uint64_t slot;
            seen |= (sps_pair_get(L[q].idx, htab, mask, slot) >> (4 * L[q].field)) & 8u;
        }
        c += seen ? 1 : 0;
    }
    unsigned long long t = sp_block_sum_u64(c, red);
    if (threadIdx.x == 0 && t) atomicAdd(out, t);
}

// Walk one unit of 64 starts: ONE filter probe and, for candidates, ONE table look-up per PAIR of starts;
// hit(start, sg) says whether the position counts; a counted k-mer is marked seen (first touch only).
template <typename F>
__device__ __forceinline__ void map_pair_scan_h(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm,
                                                int64_t s0, const sp_kparams &kp, const uint32_t *__restrict__ bloom,
                                                int nbits, unsigned long long *__restrict__ htab, uint64_t mask, F &&hit) {
    const uint64_t m1mask = kp.kmask >> 2;
    const int top = 2 * (kp.k - 1);
    uint32_t e = 0, mark = 0;
    uint64_t slot = 0;
    bool fw = true;
    sp_scan_unit_all<SP_UNIT, uint64_t>(pk, nm, s0, kp, [&](int64_t start, uint64_t fwd, uint64_t rc, bool valid_k, bool valid_k1) {
        if (!(start & 1)) {   // first of the pair: b0 + x
            e = 0;
            mark = 0;
            if (valid_k1) {
                const uint64_t xf = fwd & m1mask, xr = rc >> 2;
                const uint64_t canon = xf < xr ? xf : xr;
                fw = xf <= xr;
                if (map_bloom_test(bloom, nbits, canon)) e = sps_pair_get(canon, htab, mask, slot);
            }
            if (e & 0x77777777u) {
                const uint32_t b0 = (uint32_t)(fwd >> top) & 3u;
                const int f0 = fw ? (int)b0 : 7 - (int)b0;
                const uint32_t v0 = valid_k ? (e >> (4 * f0)) & 15u : 0u;
                if ((v0 & 7u) && hit(start, (int)(v0 & 7u) - 1) && !(v0 & 8u)) mark |= 8u << (4 * f0);
            }
        } else if (e & 0x77777777u) {   // second of the pair: x + b1
            const uint32_t b1 = (uint32_t)fwd & 3u;
            const int f1 = fw ? 4 + (int)b1 : 3 - (int)b1;
            const uint32_t v1 = valid_k ? (e >> (4 * f1)) & 15u : 0u;
            if ((v1 & 7u) && hit(start, (int)(v1 & 7u) - 1) && !(v1 & 8u)) mark |= 8u << (4 * f1);
            if (mark) atomicOr(&htab[2 * slot + 1], (unsigned long long)mark);
            mark = 0;
        }
    });
}

__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_sparse(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, sp_kparams kp, sp_map_params P,
              unsigned long long *__restrict__ htab, uint64_t mask,
              const uint32_t *__restrict__ bloom, int bloom_bits, int *__restrict__ slot_counts,
              unsigned long long *__restrict__ n_mapped, int pairs /* htab is the pair-keyed table */) {
    __shared__ int hist[MAP_LDS_ENTRIES];
    __shared__ unsigned long long red[16];
    unsigned long long mapped = 0;
    const int64_t n_ranges = (P.n_units + MAP_BLOCK - 1) / MAP_BLOCK;
    for (int64_t r = blockIdx.x; r < n_ranges; r += gridDim.x) {
        const int64_t u = r * MAP_BLOCK + threadIdx.x;
        const int64_t slot_lo = map_slot(r * MAP_RANGE, P, kp.k);
        if (P.use_lds) {
            for (int i = threadIdx.x; i < MAP_LDS_ENTRIES; i += MAP_BLOCK) hist[i] = 0;
            __syncthreads();
        }
        if (u < P.n_units) {
            auto count = [&](int64_t start, int sg) {
                const int64_t os = map_slot(start, P, kp.k);
                if (P.use_lds)
                    atomicAdd(&hist[(os - slot_lo) * P.S + sg], 1);
                else if (os < P.nslots)
                    atomicAdd(&slot_counts[os * P.S + sg], 1);
                mapped++;
                return true;
            };
            if (pairs)
                map_pair_scan_h(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, htab, mask, count);
            else
                map_pair_scan<uint64_t>(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, [&](int64_t start, uint64_t fwd, uint64_t rc) {
                    const int sg = sps_lookup(fwd < rc ? fwd : rc, htab, mask);
                    if (sg >= 0) count(start, sg);
                });
        }
        if (P.use_lds) {
            __syncthreads();
            for (int i = threadIdx.x; i < MAP_LDS_ENTRIES; i += MAP_BLOCK) {
                int v = hist[i];
                if (v) {
                    int64_t os = slot_lo + i / P.S;
                    if (os < P.nslots) atomicAdd(&slot_counts[os * P.S + (i % P.S)], v);
                }
            }
            __syncthreads();
        }
    }
    unsigned long long t = sp_block_sum_u64(mapped, red);
    if (threadIdx.x == 0 && t) atomicAdd(n_mapped, t);
}

__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_feat_sparse(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, sp_kparams kp, int64_t n_units,
                   const int64_t *__restrict__ foff, int64_t n_feat, int S,
                   unsigned long long *__restrict__ htab, uint64_t mask,
                   const uint32_t *__restrict__ bloom, int bloom_bits, unsigned long long *__restrict__ counts, int pairs) {
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; u < n_units; u += stride) {
        map_feat_cursor cur;
        cur.f = -1;
        cur.next = 0;
        if (pairs)
            map_pair_scan_h(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, htab, mask, [&](int64_t start, int sg) {
                if (!map_feat_locate(cur, start, kp.k, foff, n_feat)) return false;   // runs into the next feature
                atomicAdd(&counts[cur.f * S + sg], 1ULL);
                return true;
            });
        else
            map_pair_scan<uint64_t>(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, [&](int64_t start, uint64_t fwd, uint64_t rc) {
                if (!map_feat_locate(cur, start, kp.k, foff, n_feat)) return;   // runs into the next feature
                const int sg = sps_lookup(fwd < rc ? fwd : rc, htab, mask);
                if (sg < 0) return;
                atomicAdd(&counts[cur.f * S + sg], 1ULL);
            });
    }
}

__global__ void __launch_bounds__(256)
sps_count_seen(const unsigned long long *__restrict__ htab, int64_t n, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[16];
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        c += (htab[2 * i] != SPS_SENTINEL) ? ((htab[2 * i + 1] >> 7) & 1u) : 0u;   // empty entries are all ones
    unsigned long long t = sp_block_sum_u64(c, red);
    if (threadIdx.x == 0 && t) atomicAdd(out, t);
}

// ================================================================== host side
static void sps_free_chrom(sp_sparse_chrom &c) {
    if (c.d_keys) hipFree(c.d_keys);
    if (c.d_cnts) hipFree(c.d_cnts);
    c = sp_sparse_chrom();
}

void sp_sparse_release(sp_ctx *ctx) {
    for (auto &c : ctx->sparse) sps_free_chrom(c);
    ctx->sparse.clear();
    if (ctx->d_hkeys) hipFree(ctx->d_hkeys);
    ctx->d_hkeys = nullptr;
    ctx->hcap = 0;
    sp_buf_free(ctx->b_sp_a);
    sp_buf_free(ctx->b_sp_b);
    sp_buf_free(ctx->b_sp_c);
    sp_buf_free(ctx->b_sp_tmp);
    sp_buf_free(ctx->b_s3_small);
    sp_buf_free(ctx->b_sf_keys);
    sp_buf_free(ctx->b_sf_counts);
    sp_buf_free(ctx->b_sf_tot);
    sp_buf_free(ctx->b_sf_hist);
}

static int sps_ordered_select(sp_ctx *ctx, const unsigned long long *d_keys, const uint32_t *d_counts, int64_t n,
                              uint32_t lower, sp_sparse_chrom &out, unsigned long long *d_small /*>= 4 u64*/) {
    // d_small[0] = sum of kept counts, d_small[1] = kept runs
    const int64_t nblk = (n + SEL_SPAN - 1) / SEL_SPAN;
    int rc = sp_buf_ensure(ctx, ctx->b_sp_tmp, (nblk + 2) * 8);
    if (rc) return rc;
    unsigned long long *d_blk = (unsigned long long *)ctx->b_sp_tmp.p;
    SP_HIP(ctx, hipMemsetAsync(d_small, 0, 32, ctx->stream));
    if (n == 0) {
        out.n = 0;
        return SP_OK;
    }
    SP_LAUNCH(ctx, "sps_sel_count", sps_sel_count, dim3((unsigned)nblk), dim3(SEL_BLOCK), 0, d_counts, n, lower, d_blk,
              d_small);
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, d_blk, nblk, d_small + 1);
    unsigned long long h[2] = {0, 0};
    SP_HIP(ctx, hipMemcpyAsync(h, d_small, 16, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t keep = (int64_t)h[1];
    if (keep > out.cap) {
        if (out.d_keys) hipFree(out.d_keys);
        if (out.d_cnts) hipFree(out.d_cnts);
        out.d_keys = nullptr;
        out.d_cnts = nullptr;
        out.cap = 0;
        SP_HIP(ctx, hipMalloc(&out.d_keys, (size_t)(keep + 1) * 8));
        SP_HIP(ctx, hipMalloc(&out.d_cnts, (size_t)(keep + 1) * 4));
        out.cap = keep;
    }
    out.n = keep;
    out.length_sum = (int64_t)h[0];
    if (keep)
        SP_LAUNCH(ctx, "sps_sel_write", sps_sel_write, dim3((unsigned)nblk), dim3(SEL_BLOCK), 0, d_keys, d_counts, n,
                  lower, d_blk, (unsigned long long *)out.d_keys, out.d_cnts);
    return SP_OK;
}

int sp_sparse_count(sp_ctx *ctx, int k, int lower) {
    const size_t C = ctx->chroms.size();
    if (ctx->sparse.size() != C) {
        for (auto &c : ctx->sparse) sps_free_chrom(c);
        ctx->sparse.assign(C, sp_sparse_chrom());
    }
    const sp_kparams kp = sp_make_kparams(k);
    const unsigned end_bit = (2 * k + 1 > 64) ? 64u : (unsigned)(2 * k + 1);
    void *scr = nullptr;
    int rc = sp_scratch(ctx, 256, &scr);
    if (rc) return rc;
    unsigned long long *d_small = (unsigned long long *)scr;
    for (size_t ci = 0; ci < C; ci++) {
        sp_chrom &c = ctx->chroms[ci];
        sp_sparse_chrom &o = ctx->sparse[ci];
        const int64_t n = c.len;
        o.n = 0;
        o.length_sum = 0;
        c.length_sum = 0;
        c.n_dump = 0;
        if (n <= 0) continue;
        rc = sp_buf_ensure(ctx, ctx->b_sp_a, n * 8);
        if (rc) return rc;
        rc = sp_buf_ensure(ctx, ctx->b_sp_b, n * 8);
        if (rc) return rc;
        rc = sp_buf_ensure(ctx, ctx->b_sp_c, n * 4 + 64);
        if (rc) return rc;
        unsigned long long *A = (unsigned long long *)ctx->b_sp_a.p, *B = (unsigned long long *)ctx->b_sp_b.p;
        uint32_t *Cn = (uint32_t *)ctx->b_sp_c.p;
        SP_HIP(ctx, hipMemsetAsync(A, 0xff, (size_t)n * 8, ctx->stream));
        SP_HIP(ctx, hipMemsetAsync(d_small, 0, 64, ctx->stream));
        const int64_t n_units = (n + SP_UNIT - 1) / SP_UNIT;
        int64_t grid = (n_units + 255) / 256;
        if (grid > (int64_t)ctx->n_cu * 16) grid = (int64_t)ctx->n_cu * 16;
        SP_LAUNCH(ctx, "sps_keygen", sps_keygen, dim3((unsigned)grid), dim3(256), 0, c.d_pk, c.d_nm, n_units, kp, A,
                  d_small + 4);
        unsigned long long nv = 0;
        SP_HIP(ctx, hipMemcpyAsync(&nv, d_small + 4, 8, hipMemcpyDeviceToHost, ctx->stream));
        // sort (sentinels, having bit 2k set, end up last)
        size_t tmp_bytes = 0;
        SP_HIP(ctx, rocprim::radix_sort_keys(nullptr, tmp_bytes, A, B, (size_t)n, 0u, end_bit, ctx->stream));
        rc = sp_buf_ensure(ctx, ctx->b_sp_tmp, (int64_t)tmp_bytes + 256);
        if (rc) return rc;
        SP_HIP(ctx, rocprim::radix_sort_keys(ctx->b_sp_tmp.p, tmp_bytes, A, B, (size_t)n, 0u, end_bit, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (nv == 0) continue;
        if (nv >= (1ULL << 32)) return sp_fail(ctx, SP_EUNSUP, "k > 15: chromosomes of 2^32 or more k-mers are not supported");
        // run-length encode the valid prefix: unique keys -> A, counts -> Cn, number of runs -> d_small[5]
        tmp_bytes = 0;
        SP_HIP(ctx, rocprim::run_length_encode(nullptr, tmp_bytes, B, (unsigned int)nv, A, Cn, d_small + 5, ctx->stream));
        rc = sp_buf_ensure(ctx, ctx->b_sp_tmp, (int64_t)tmp_bytes + 256);
        if (rc) return rc;
        SP_HIP(ctx, rocprim::run_length_encode(ctx->b_sp_tmp.p, tmp_bytes, B, (unsigned int)nv, A, Cn, d_small + 5,
                                               ctx->stream));
        unsigned long long nruns = 0;
        SP_HIP(ctx, hipMemcpyAsync(&nruns, d_small + 5, 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rc = sps_ordered_select(ctx, A, Cn, (int64_t)nruns, (uint32_t)lower, o, d_small);
        if (rc) return rc;
        c.length_sum = o.length_sum;
        c.n_dump = o.n;
    }
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_sparse_dump(sp_ctx *ctx, int chrom, uint64_t *keys, uint32_t *counts) {
    sp_sparse_chrom &o = ctx->sparse[(size_t)chrom];
    if (o.n == 0) return SP_OK;
    SP_HIP(ctx, hipMemcpyAsync(keys, o.d_keys, (size_t)o.n * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(counts, o.d_cnts, (size_t)o.n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->list_mode) {
        // engine 3 keeps dense SLOTS (ascending); a dump is canonical k-mers in ascending order: convert and sort
        // here, on the host -- the dump is the compatibility export (jellyfish-format files, tests), not the hot path
        const sp_kparams kp = sp_make_kparams(ctx->k);
        try {
            std::vector<std::pair<uint64_t, uint32_t>> v((size_t)o.n);
            for (int64_t i = 0; i < o.n; i++) v[(size_t)i] = {sp_key_of_slot(keys[i], kp), counts[i]};
            std::sort(v.begin(), v.end());
            for (int64_t i = 0; i < o.n; i++) {
                keys[i] = v[(size_t)i].first;
                counts[i] = v[(size_t)i].second;
            }
        } catch (const std::bad_alloc &) {
            return sp_fail(ctx, SP_ENOMEM, "sp_dump: out of host memory sorting %lld k-mers", (long long)o.n);
        }
    }
    return SP_OK;
}

// engine 3: the rows the list filter emitted carry dense slots; the API speaks canonical k-mers
__global__ void __launch_bounds__(256)
sps_slots_to_keys(unsigned long long *__restrict__ keys, int64_t n, sp_kparams kp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = sp_key_of_slot(keys[i], kp);
}

// the lists the filter works on: the local chromosomes, or a caller-owned key-range view (sp_sparse_view)
static int sps_C(sp_ctx *ctx) { return ctx->sv_on ? (int)ctx->sv_keys.size() : (int)ctx->chroms.size(); }
static int64_t sps_n(sp_ctx *ctx, int c) { return ctx->sv_on ? ctx->sv_n[(size_t)c] : ctx->sparse[(size_t)c].n; }
static const unsigned long long *sps_keys(sp_ctx *ctx, int c) {
    return (const unsigned long long *)(ctx->sv_on ? ctx->sv_keys[(size_t)c] : ctx->sparse[(size_t)c].d_keys);
}
static const uint32_t *sps_cnts(sp_ctx *ctx, int c) {
    return ctx->sv_on ? ctx->sv_cnts[(size_t)c] : ctx->sparse[(size_t)c].d_cnts;
}
static int64_t sps_len(sp_ctx *ctx, int c) {
    return ctx->sv_on ? ctx->fv_lengths[(size_t)c] : ctx->chroms[(size_t)c].length_sum;
}

int sp_sparse_filter(sp_ctx *ctx, int n_sets, const int32_t *set_off, const int32_t *unit_off,
                     const int32_t *unit_chrom, const std::vector<double> &den, double min_fold, int baseline,
                     double min_freq, double max_freq, double ratio) {
    const int C = sps_C(ctx);
    if (C > SPS_MAXC) return sp_fail(ctx, SP_EUNSUP, "list filter (k > 15, or engine 3): at most %d chromosomes supported (got %d)", SPS_MAXC, C);
    int64_t total = 0;
    for (int c = 0; c < C; c++) total += sps_n(ctx, c);
    ctx->sf_n = total;
    ctx->n_union = ctx->n_rows = ctx->n_hist = 0;
    if (total == 0) {
        ctx->filtered = true;
        return SP_OK;
    }
    int rc = sp_buf_ensure(ctx, ctx->b_sp_a, total * 16);   // keys | vals (unsorted)
    if (rc) return rc;
    rc = sp_buf_ensure(ctx, ctx->b_sp_b, total * 16);       // keys | vals (sorted)
    if (rc) return rc;
    rc = sp_buf_ensure(ctx, ctx->b_sp_c, total + 64);       // flags
    if (rc) return rc;
    unsigned long long *K0 = (unsigned long long *)ctx->b_sp_a.p, *V0 = K0 + total;
    unsigned long long *K1 = (unsigned long long *)ctx->b_sp_b.p, *V1 = K1 + total;
    uint8_t *flags = (uint8_t *)ctx->b_sp_c.p;
    int64_t off = 0;
    for (int c = 0; c < C; c++) {
        const int64_t n_c = sps_n(ctx, c);
        if (n_c)
            SP_LAUNCH(ctx, "sps_concat", sps_concat, dim3((unsigned)((n_c + 255) / 256)), dim3(256), 0,
                      sps_keys(ctx, c), sps_cnts(ctx, c), n_c, c, K0 + off, V0 + off);
        off += n_c;
    }
    const unsigned end_bit = (2 * ctx->k > 64) ? 64u : (unsigned)(2 * ctx->k);
    size_t tmp_bytes = 0;
    SP_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, K0, K1, V0, V1, (size_t)total, 0u, end_bit, ctx->stream));
    // parameter block + block tallies live behind the rocprim temp storage
    const int n_units = set_off[n_sets], n_uc = unit_off[n_units];
    const int64_t nblk = (total + SEL_SPAN - 1) / SEL_SPAN;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_set = al(tmp_bytes), o_uo = o_set + al((size_t)(n_sets + 1) * 4), o_uc = o_uo + al((size_t)(n_units + 1) * 4),
           o_den = o_uc + al((size_t)(n_uc + 1) * 4), o_blk_r = o_den + al((size_t)n_units * 16),
           o_blk_h = o_blk_r + al((size_t)(nblk + 1) * 8), o_small = o_blk_h + al((size_t)(nblk + 1) * 8),
           all_b = o_small + 256;
    rc = sp_buf_ensure(ctx, ctx->b_sp_tmp, (int64_t)all_b);
    if (rc) return rc;
    char *T = (char *)ctx->b_sp_tmp.p;
    SP_HIP(ctx, rocprim::radix_sort_pairs(T, tmp_bytes, K0, K1, V0, V1, (size_t)total, 0u, end_bit, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(T + o_set, set_off, (size_t)(n_sets + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(T + o_uo, unit_off, (size_t)(n_units + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n_uc) SP_HIP(ctx, hipMemcpyAsync(T + o_uc, unit_chrom, (size_t)n_uc * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(T + o_den, den.data(), (size_t)n_units * 16, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemsetAsync(T + o_small, 0, 256, ctx->stream));
    sps_filter_args A;
    A.C = C;
    A.F.n_sets = n_sets;
    A.F.n_multi = 0;
    for (int st = 0; st < n_sets; st++) A.F.n_multi += (set_off[st + 1] - set_off[st]) > 1;
    A.F.baseline = baseline;
    A.F.set_off = (const int32_t *)(T + o_set);
    A.F.unit_off = (const int32_t *)(T + o_uo);
    A.F.unit_chrom = (const int32_t *)(T + o_uc);
    A.F.unit_den = (const double *)(T + o_den);
    A.F.unit_inv = A.F.unit_den + n_units;
    A.F.min_fold = min_fold;
    A.F.min_freq = min_freq;
    A.F.max_freq = max_freq;
    A.F.ratio = ratio;
    unsigned long long *blk_r = (unsigned long long *)(T + o_blk_r), *blk_h = (unsigned long long *)(T + o_blk_h),
                       *small = (unsigned long long *)(T + o_small);
    SP_LAUNCH(ctx, "sps_eval", sps_eval, dim3((unsigned)nblk), dim3(SEL_BLOCK), 0, K1, V1, total, A, flags, blk_r, blk_h,
              small);
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, blk_r, nblk, small + 1);
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, blk_h, nblk, small + 2);
    unsigned long long h[3] = {0, 0, 0};
    SP_HIP(ctx, hipMemcpyAsync(h, small, 24, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_union = (int64_t)h[0];
    ctx->n_rows = (int64_t)h[1];
    ctx->n_hist = (int64_t)h[2];
    // materialise the results now (the sort buffers are reused by the next call)
    const int64_t M = ctx->n_rows, H = ctx->n_hist;
    rc = sp_buf_ensure(ctx, ctx->b_sf_keys, (M + 1) * 8);
    if (rc) return rc;
    rc = sp_buf_ensure(ctx, ctx->b_sf_counts, (M + 1) * (int64_t)C * 4);
    if (rc) return rc;
    rc = sp_buf_ensure(ctx, ctx->b_sf_tot, (M + 1) * 8);
    if (rc) return rc;
    rc = sp_buf_ensure(ctx, ctx->b_sf_hist, (H + 1) * 8);
    if (rc) return rc;
    if (M)
        SP_LAUNCH(ctx, "sps_emit", sps_emit, dim3((unsigned)nblk), dim3(SEL_BLOCK), 0, K1, V1, total, C,
                  (const uint8_t *)flags, (uint8_t)1, (const unsigned long long *)blk_r,
                  (unsigned long long *)ctx->b_sf_keys.p, (uint32_t *)ctx->b_sf_counts.p,
                  (unsigned long long *)ctx->b_sf_tot.p);
    if (H)
        SP_LAUNCH(ctx, "sps_emit_hist", sps_emit, dim3((unsigned)nblk), dim3(SEL_BLOCK), 0, K1, V1, total, C,
                  (const uint8_t *)flags, (uint8_t)2, (const unsigned long long *)blk_h, (unsigned long long *)nullptr,
                  (uint32_t *)nullptr, (unsigned long long *)ctx->b_sf_hist.p);
    if (M && ctx->list_mode)
        SP_LAUNCH(ctx, "sps_slots_to_keys", sps_slots_to_keys, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                  (unsigned long long *)ctx->b_sf_keys.p, M, sp_make_kparams(ctx->k));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->filtered = true;
    return SP_OK;
}

int sp_sparse_fetch(sp_ctx *ctx, bool hist, uint64_t *keys, uint32_t *counts, double *freqs, uint64_t *tot) {
    const int C = sps_C(ctx);
    const int64_t M = hist ? ctx->n_hist : ctx->n_rows;
    if (M == 0) return SP_OK;
    if (hist) {
        SP_HIP(ctx, hipMemcpyAsync(tot, ctx->b_sf_hist.p, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return SP_OK;
    }
    std::vector<uint32_t> tmp;
    uint32_t *cdst = counts;
    if (!counts && freqs) {
        tmp.resize((size_t)M * C);
        cdst = tmp.data();
    }
    if (keys) SP_HIP(ctx, hipMemcpyAsync(keys, ctx->b_sf_keys.p, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (tot) SP_HIP(ctx, hipMemcpyAsync(tot, ctx->b_sf_tot.p, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (cdst) SP_HIP(ctx, hipMemcpyAsync(cdst, ctx->b_sf_counts.p, (size_t)M * C * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (freqs)
        for (int64_t r = 0; r < M; r++)
            for (int c = 0; c < C; c++)   // count/length in fp64 (Jellyfish.py:647); IEEE division, same bits as the device path
                freqs[r * C + c] = (double)cdst[r * C + c] / (double)sps_len(ctx, c);
    return SP_OK;
}

int sp_map_filter_build(sp_ctx *ctx, const unsigned long long *d_keys, int64_t n);   // sp_map.hip

int sp_sparse_labels_set(sp_ctx *ctx, const uint64_t *keys, const uint8_t *sg, int64_t n) {
    // <= 7 subgenomes: the pair-keyed table (one look-up per candidate PAIR of starts); else one entry per k-mer
    const char *eng = getenv("SP_MAP_ENGINE");
    ctx->map_engine = (ctx->n_sg > 7 || (eng && eng[0] == '1')) ? 1 : 0;
    const bool pairs = ctx->map_engine == 0;
    int64_t cap = 1024;
    while (cap < (pairs ? 4 : 2) * n + 16) cap <<= 1;
    if (cap != ctx->hcap) {
        if (ctx->d_hkeys) hipFree(ctx->d_hkeys);
        ctx->d_hkeys = nullptr;
        SP_HIP(ctx, hipMalloc(&ctx->d_hkeys, (size_t)cap * 16));
        ctx->hcap = cap;
    }
    if (pairs) {
        SP_LAUNCH(ctx, "sps_pair_init", sps_pair_init, dim3((unsigned)(ctx->n_cu * 8)), dim3(256), 0,
                  (unsigned long long *)ctx->d_hkeys, cap);
    } else {
        // {key = sentinel (all ones), label = 0}: 0xff everywhere, then the label words are written on insert
        // and read only after a key match, so they need no clearing
        SP_HIP(ctx, hipMemsetAsync(ctx->d_hkeys, 0xff, (size_t)cap * 16, ctx->stream));
    }
    if (n == 0) return sp_map_filter_build(ctx, nullptr, 0);
    // the labelled keys stay on the device (sp_labels_hit walks them in pair mode); the dense engine's pair table, if
    // any, was built from the keys this overwrites: it must be cleared in full next time
    ctx->ptab_k = 0;
    ctx->ptab_n = 0;
    int rcb = sp_buf_ensure(ctx, ctx->b_labkeys, n * 9 + 64);
    if (rcb) return rcb;
    unsigned long long *d_keys = (unsigned long long *)ctx->b_labkeys.p;
    uint8_t *d_sg = (uint8_t *)(d_keys + n);
    SP_HIP(ctx, hipMemcpyAsync(d_keys, keys, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_sg, sg, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (pairs)
        SP_LAUNCH(ctx, "sps_pair_insert", sps_pair_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                  (const unsigned long long *)d_keys, (const uint8_t *)d_sg, n, ctx->k, (unsigned long long *)ctx->d_hkeys,
                  (uint64_t)(cap - 1));
    else
        SP_LAUNCH(ctx, "sps_hash_insert", sps_hash_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                  (const unsigned long long *)d_keys, (const uint8_t *)d_sg, n, (unsigned long long *)ctx->d_hkeys,
                  (uint64_t)(cap - 1));
    const int rcf = sp_map_filter_build(ctx, d_keys, n);
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rcf;
}

int sp_sparse_map_launch(sp_ctx *ctx, sp_chrom &c, const sp_map_params &P, int *d_counts, unsigned long long *d_n) {
    if (P.n_units == 0) return SP_OK;
    const sp_kparams kp = sp_make_kparams(ctx->k);
    int64_t n_ranges = (P.n_units + MAP_BLOCK - 1) / MAP_BLOCK;
    int64_t grid = n_ranges;
    if (grid > (int64_t)ctx->n_cu * 16) grid = (int64_t)ctx->n_cu * 16;
    SP_LAUNCH(ctx, "k5_map_sparse", k5_map_sparse, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, c.d_pk, c.d_nm, kp, P,
              (unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1), ctx->d_bloom, ctx->bloom_bits, d_counts,
              d_n, ctx->map_engine == 0 ? 1 : 0);
    return SP_OK;
}

int sp_sparse_feat_launch(sp_ctx *ctx, const uint32_t *d_pk, const uint32_t *d_nm, int64_t n_units, const int64_t *d_foff,
                          int64_t n_feat, int S, unsigned long long *d_counts) {
    const sp_kparams kp = sp_make_kparams(ctx->k);
    int64_t grid = (n_units + MAP_BLOCK - 1) / MAP_BLOCK;
    if (grid > (int64_t)ctx->n_cu * 16) grid = (int64_t)ctx->n_cu * 16;
    SP_LAUNCH(ctx, "k5_map_feat_sparse", k5_map_feat_sparse, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, d_pk, d_nm, kp,
              n_units, d_foff, n_feat, S, (unsigned long long *)ctx->d_hkeys,
              (uint64_t)(ctx->hcap - 1), ctx->d_bloom, ctx->bloom_bits, d_counts, ctx->map_engine == 0 ? 1 : 0);
    return SP_OK;
}

int sp_sparse_hit(sp_ctx *ctx, unsigned long long *d_n) {
    if (ctx->map_engine == 0) {
        if (ctx->n_labels > 0)
            SP_LAUNCH(ctx, "sps_pair_seen", sps_pair_seen, dim3((unsigned)(ctx->n_cu * 4)), dim3(256), 0,
                      (const unsigned long long *)ctx->b_labkeys.p, ctx->n_labels, ctx->k,
                      (const unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1), d_n);
        return SP_OK;
    }
    SP_LAUNCH(ctx, "sps_count_seen", sps_count_seen, dim3((unsigned)(ctx->n_cu * 4)), dim3(256), 0,
              (const unsigned long long *)ctx->d_hkeys, ctx->hcap, d_n);
    return SP_OK;
}

// ------------------------------------------------------------------ multi-GPU support (k > 15)
// The dense path exchanges slot-range slices of the count tables; with 64-bit keys the same exchange
// is a KEY-RANGE partition of every chromosome's sorted (key, count >= lower) list: the owner cuts its
// lists at common splitters (sp_sparse_split), copies the pieces into the buffers handed to RCCL
// (sp_sparse_export), and the receiver filters its key range of all chromosomes (sp_sparse_view).
__global__ void sps_lower_bound(const unsigned long long *__restrict__ keys, int64_t n,
                                const unsigned long long *__restrict__ q, int nq, long long *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const unsigned long long x = q[i];
    int64_t lo = 0, hi = n;   // first index with keys[idx] >= x
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    out[i] = lo;
}

extern "C" {

int sp_sparse_sizes(sp_ctx *ctx, int64_t *n) {
    if (!ctx || !n) return sp_fail(ctx, SP_EINVAL, "sp_sparse_sizes: bad arguments");
    if (!ctx->sparse_mode || !ctx->counted) return sp_fail(ctx, SP_EINVAL, "sp_sparse_sizes: call sp_count with k > 15 first");
    for (size_t c = 0; c < ctx->sparse.size(); c++) n[c] = ctx->sparse[c].n;
    return SP_OK;
}

int sp_sparse_sample(sp_ctx *ctx, int chrom, int64_t n_samples, uint64_t *keys, int64_t *n_out) {
    if (!ctx || !keys || !n_out || n_samples < 1 || chrom < 0 || chrom >= (int)ctx->sparse.size())
        return sp_fail(ctx, SP_EINVAL, "sp_sparse_sample: bad arguments");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const sp_sparse_chrom &o = ctx->sparse[(size_t)chrom];
    const int64_t stride = o.n / n_samples;
    if (stride < 1) {   // short list: all of it
        if (o.n) SP_HIP(ctx, hipMemcpyAsync(keys, o.d_keys, (size_t)o.n * 8, hipMemcpyDeviceToHost, ctx->stream));
        *n_out = o.n;
    } else {            // every stride-th key of the sorted list
        SP_HIP(ctx, hipMemcpy2DAsync(keys, 8, o.d_keys, (size_t)stride * 8, 8, (size_t)n_samples, hipMemcpyDeviceToHost,
                                     ctx->stream));
        *n_out = n_samples;
    }
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_sparse_split(sp_ctx *ctx, int chrom, const uint64_t *splitters, int n_split, int64_t *bounds) {
    if (!ctx || !bounds || n_split < 0 || (n_split > 0 && !splitters) || n_split > 4096 || chrom < 0 ||
        chrom >= (int)ctx->sparse.size())
        return sp_fail(ctx, SP_EINVAL, "sp_sparse_split: bad arguments");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const sp_sparse_chrom &o = ctx->sparse[(size_t)chrom];
    bounds[0] = 0;
    bounds[n_split + 1] = o.n;
    if (n_split == 0) return SP_OK;
    void *scr = nullptr;
    int rc = sp_scratch(ctx, (int64_t)n_split * 16 + 256, &scr);
    if (rc) return rc;
    unsigned long long *d_q = (unsigned long long *)scr;
    long long *d_out = (long long *)(d_q + n_split);
    SP_HIP(ctx, hipMemcpyAsync(d_q, splitters, (size_t)n_split * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_LAUNCH(ctx, "sps_lower_bound", sps_lower_bound, dim3((unsigned)((n_split + 63) / 64)), dim3(64), 0,
              (const unsigned long long *)o.d_keys, o.n, (const unsigned long long *)d_q, n_split, d_out);
    SP_HIP(ctx, hipMemcpyAsync(bounds + 1, d_out, (size_t)n_split * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_sparse_export(sp_ctx *ctx, int chrom, int64_t first, int64_t count, void *d_keys, void *d_counts) {
    if (!ctx || chrom < 0 || chrom >= (int)ctx->sparse.size() || first < 0 || count < 0)
        return sp_fail(ctx, SP_EINVAL, "sp_sparse_export: bad arguments");
    const sp_sparse_chrom &o = ctx->sparse[(size_t)chrom];
    if (first + count > o.n) return sp_fail(ctx, SP_EINVAL, "sp_sparse_export: range exceeds the list (%lld)", (long long)o.n);
    if (count == 0) return SP_OK;
    if (!d_keys || !d_counts) return sp_fail(ctx, SP_EINVAL, "sp_sparse_export: NULL destination");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipMemcpyAsync(d_keys, o.d_keys + first, (size_t)count * 8, hipMemcpyDeviceToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_counts, o.d_cnts + first, (size_t)count * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return SP_OK;   // asynchronous on the context's stream: sp_sync before another stream reads the buffers
}

int sp_sparse_view(sp_ctx *ctx, int C, const void *const *d_keys, const void *const *d_counts, const int64_t *n,
                   const int64_t *lengths, int k, int lower_count) {
    if (!ctx) return SP_EINVAL;
    if (!d_keys) {   // back to the local chromosomes
        ctx->sv_on = false;
        ctx->sv_keys.clear();
        ctx->sv_cnts.clear();
        ctx->sv_n.clear();
        ctx->fv_lengths.clear();
        ctx->filtered = false;
        return SP_OK;
    }
    if (C <= 0 || !d_counts || !n || !lengths || k < 16 || k > 32)
        return sp_fail(ctx, SP_EINVAL, "sp_sparse_view: bad arguments (k = 16..32)");
    if (ctx->fv_on) return sp_fail(ctx, SP_EINVAL, "sp_sparse_view: a dense filter view is active");
    if (ctx->k != 0 && ctx->k != k) return sp_fail(ctx, SP_EINVAL, "sp_sparse_view: k=%d but the context counted with k=%d", k, ctx->k);
    for (int i = 0; i < C; i++)
        if (n[i] < 0 || (n[i] > 0 && (!d_keys[i] || !d_counts[i])))
            return sp_fail(ctx, SP_EINVAL, "sp_sparse_view: list %d is NULL", i);
    ctx->sv_keys.assign((size_t)C, nullptr);
    ctx->sv_cnts.assign((size_t)C, nullptr);
    ctx->sv_n.assign((size_t)C, 0);
    ctx->fv_lengths.assign((size_t)C, 0);
    for (int i = 0; i < C; i++) {
        ctx->sv_keys[(size_t)i] = (const uint64_t *)d_keys[i];
        ctx->sv_cnts[(size_t)i] = (const uint32_t *)d_counts[i];
        ctx->sv_n[(size_t)i] = n[i];
        ctx->fv_lengths[(size_t)i] = lengths[i];
    }
    if (ctx->k == 0) {   // a rank that owns no chromosome still filters its key range
        ctx->k = k;
        ctx->sparse_mode = true;
    }
    ctx->lower = lower_count < 1 ? 1 : lower_count;
    ctx->sv_on = true;
    ctx->filtered = false;
    return SP_OK;
}

}  // extern "C"
