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

// K5 for k > 15, per-k-mer table (more than 7 subgenomes, or SP_MAP_ENGINE=1): the rolling scan, one filter probe per pair
// of starts, one 16-byte look-up per candidate start.  (The pair-keyed table takes k5_map_sparse2 below; the unrolled
// pair kernel of rounds 1-4 -- 532 KB of machine code -- was removed in round 6.)
__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_sparse_lab(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, sp_kparams kp, sp_map_params P,
                  unsigned long long *__restrict__ htab, uint64_t mask,
                  const uint32_t *__restrict__ bloom, int bloom_bits, int *__restrict__ slot_counts,
                  unsigned long long *__restrict__ n_mapped) {
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
            map_pair_scan<uint64_t>(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, [&](int64_t start, uint64_t fwd, uint64_t rc) {
                const int sg = sps_lookup(fwd < rc ? fwd : rc, htab, mask);
                if (sg < 0) return;
                const int64_t os = map_slot(start, P, kp.k);
                if (P.use_lds)
                    atomicAdd(&hist[(os - slot_lo) * P.S + sg], 1);
                else if (os < P.nslots)
                    atomicAdd(&slot_counts[os * P.S + sg], 1);
                mapped++;
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

// ------------------------------------------------------------------ quad-bucket label table for k > 15 (round 6)
// The k <= 15 compact table (sp_map.h) with 64-bit entries: the two pairs of a QUAD of starts share the (k-3)-mer core t =
// last k-3 bases of x1 = first k-3 bases of x2, the table is addressed by the canonical core -- bucket = top bb bits of a
// bijective mix of t -- and ONE 32-byte bucket of four tagged entries answers all four starts, where the pair-keyed hash table
// above costs one 16-byte look-up per candidate PAIR (1.5 G against 0.96 G per wheat-like pass; the look-ups are what the
// kernel waits for).  An entry belongs to a (k-1)-mer y read in the orientation in which its core at one end is canonical:
// side 0 ("L") y = e + t, side 1 ("R") y = t + e, e = the two bases beyond t; it holds the labels of the eight k-mers b + y,
// y + b in THAT orientation (eight 3-bit fields: label 1..3, "seen" in the third bit).  Entry = tag << 25 | overflow flag << 24 |
// fields; tag = the tb = 2 (k - 3) - bb low bits of mix(t) << 5 | side << 4 | e: bucket + tag determine (t, side, e), a tag
// match is exact; tb + 5 <= 39 bits (else -- k = 31 / 32 with a small label set -- the hash table stays).  0 = empty.  A key
// that finds its bucket full goes to an open-addressing overflow table of {key id + 1, fields} pairs.  S <= 3.
#define SQ_FIELD 3
#define SQ_ANY 0x6DB6DBULL
#define SQ_PAYLOAD 0xFFFFFFULL
#define SQ_OVF (1ULL << 24)
#define SQ_TAG_BITS 39
struct sq_tab {
    unsigned long long *buckets;      // 4 entries (32 bytes) per bucket, or NULL: the pair-keyed hash table is in use
    unsigned long long *ovf;          // overflow table: {key id + 1, fields} pairs, 0 = empty
    uint64_t ovf_mask;                // its pairs - 1
    int sb, tb;                       // bits of the core 2 (k - 3); bits of mix(t) kept in the tag
};
__host__ __device__ __forceinline__ uint64_t sq_mix(uint64_t x, int kb) {      // a bijection of the kb-bit values (kb <= 58)
    const uint64_t m = (1ULL << kb) - 1ULL;
    x = (x * 0x9E3779B97F4A7C15ULL) & m;
    x ^= x >> ((kb + 1) / 2);
    x = (x * 0xD6E8FEB86659FD93ULL) & m;
    x ^= x >> ((kb + 1) / 2);
    return x;
}
struct sq_key {
    uint64_t bucket, tag, kid;
};
__host__ __device__ __forceinline__ sq_key sq_key_of(const sq_tab &T, uint64_t t, uint32_t side, uint32_t e) {
    const uint64_t h = sq_mix(t, T.sb);
    sq_key r;
    r.bucket = h >> T.tb;
    r.tag = ((h & ((1ULL << T.tb) - 1ULL)) << 5) | ((uint64_t)side << 4) | e;
    r.kid = (t << 5) | ((uint64_t)side << 4) | e;
    return r;
}
__device__ __forceinline__ void sq_insert(const sq_tab &T, const sq_key &q, uint64_t bits, unsigned long long *fail) {
    unsigned long long *e = T.buckets + 4 * q.bucket;
    const unsigned long long fresh = (q.tag << 25) | bits;
    for (int j = 0; j < 4; j++) {
        const int i = (int)((q.tag + (uint64_t)j) & 3ULL);      // (every key starts at the entry its own tag names)
        const unsigned long long old = atomicCAS(&e[i], 0ULL, fresh);
        if (old == 0ULL) return;
        if ((old >> 25) == q.tag && (old & SQ_PAYLOAD)) {
            atomicOr(&e[i], (unsigned long long)bits);
            return;
        }
    }
    atomicOr(&e[0], (unsigned long long)SQ_OVF);
    uint64_t i = (sps_mix(q.kid) >> 7) & T.ovf_mask;
    for (uint64_t probes = 0; probes <= T.ovf_mask; probes++) {
        const unsigned long long old = atomicCAS(&T.ovf[2 * i], 0ULL, (unsigned long long)(q.kid + 1ULL));
        if (old == 0ULL) atomicAdd(fail + 1, 1ULL);      // (statistics: keys in the overflow table)
        if (old == 0ULL || old == q.kid + 1ULL) {
            atomicOr(&T.ovf[2 * i + 1], (unsigned long long)bits);
            return;
        }
        i = (i + 1ULL) & T.ovf_mask;
    }
    atomicAdd(fail, 1ULL);      // the overflow table is full: the host falls back to the hash table
}
struct sq_hit {
    uint32_t fields, loc;             // loc: 4 * bucket + entry, or 0x80000000 | overflow pair: where the "seen" bits go
};
__device__ __forceinline__ sq_hit sq_find(const sq_tab &T, const ulonglong2 &B0, const ulonglong2 &B1, uint64_t bucket, uint64_t tag,
                                          uint64_t t, uint32_t side_e /* side << 4 | e */) {
    sq_hit r;
    const unsigned long long w[4] = {B0.x, B0.y, B1.x, B1.y};
    r.fields = 0;
    r.loc = (uint32_t)(4ULL * bucket);
#pragma unroll
    for (int i = 3; i >= 0; i--)
        if ((w[i] >> 25) == tag && (w[i] & SQ_PAYLOAD)) {
            r.fields = (uint32_t)(w[i] & SQ_PAYLOAD);
            r.loc = (uint32_t)(4ULL * bucket) + (uint32_t)i;
        }
    if (!r.fields && (B0.x & SQ_OVF)) {         // rare: the bucket overflowed and the key is in none of its entries
        const uint64_t kid = (t << 5) | side_e;
        uint64_t i = (sps_mix(kid) >> 7) & T.ovf_mask;
        for (uint64_t probes = 0; probes <= T.ovf_mask; probes++) {      // (bounded: a full table has no empty slot to stop at)
            const unsigned long long o = T.ovf[2 * i];
            if (o == 0ULL) break;
            if (o == kid + 1ULL) {
                r.fields = (uint32_t)(T.ovf[2 * i + 1] & SQ_PAYLOAD);
                r.loc = 0x80000000u | (uint32_t)i;
                break;
            }
            i = (i + 1ULL) & T.ovf_mask;
        }
    }
    return r;
}
__device__ __forceinline__ void sq_mark(const sq_tab &T, uint32_t loc, uint32_t bits) {
    if (loc & 0x80000000u) atomicOr(&T.ovf[2 * (size_t)(loc & 0x7fffffffu) + 1], (unsigned long long)bits);
    else atomicOr(&T.buckets[loc], (unsigned long long)bits);
}
// The (up to four) table keys under which the k-mer `o` (ONE orientation, as given) is entered / found: its prefix and its
// suffix (k-1)-mer, each under its first and its last (k-3)-mer -- but only where that core is canonical AS READ in this
// orientation; the other orientation of the k-mer supplies the rest (a palindromic core is entered from both).  k >= 16.
struct sq_site {
    uint64_t t;
    uint32_t side, e;
    int field;
};
__host__ __device__ __forceinline__ int sq_sites(uint64_t o, int k, sq_site out[4]) {
    const int sb = 2 * (k - 3);
    const uint64_t m1mask = (1ULL << (2 * (k - 1))) - 1ULL, smask = (1ULL << sb) - 1ULL;
    const uint64_t x[2] = {o >> 2, o & m1mask};                                          // prefix / suffix (k-1)-mer
    const int field[2] = {4 + (int)(o & 3ULL), (int)((o >> (2 * (k - 1))) & 3ULL)};      // o = x + b  /  o = b + x
    int n = 0;
    for (int i = 0; i < 2; i++) {
        const uint64_t u1 = x[i] >> 4, u2 = x[i] & smask;
        if (u1 <= sp_revcomp(u1, k - 3)) {       // x = u1 + e: side R
            out[n].t = u1; out[n].side = 1u; out[n].e = (uint32_t)(x[i] & 15ULL); out[n].field = field[i];
            n++;
        }
        if (u2 <= sp_revcomp(u2, k - 3)) {       // x = e + u2: side L
            out[n].t = u2; out[n].side = 0u; out[n].e = (uint32_t)(x[i] >> sb); out[n].field = field[i];
            n++;
        }
    }
    return n;
}
__global__ void __launch_bounds__(256)
sq_build(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ sg, int64_t n, int k, sq_tab T,
         unsigned long long *__restrict__ fail) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t o2[2] = {keys[i], sp_revcomp(keys[i], k)};
    const uint64_t l = 1ULL + sg[i];
    for (int r = 0; r < 2; r++) {
        sq_site st[4];
        const int ns = sq_sites(o2[r], k, st);
        for (int j = 0; j < ns; j++) sq_insert(T, sq_key_of(T, st[j].t, st[j].side, st[j].e), l << (SQ_FIELD * st[j].field), fail);
    }
}
__global__ void __launch_bounds__(256)
sq_seen(const unsigned long long *__restrict__ keys, int64_t n, int k, sq_tab T, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[16];
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t o2[2] = {keys[i], sp_revcomp(keys[i], k)};
        uint32_t seen = 0;
        for (int r = 0; r < 2; r++) {
            sq_site st[4];
            const int ns = sq_sites(o2[r], k, st);
            for (int j = 0; j < ns; j++) {
                const sq_key q = sq_key_of(T, st[j].t, st[j].side, st[j].e);
                const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(T.buckets + 4 * q.bucket);
                seen |= (sq_find(T, bp[0], bp[1], q.bucket, q.tag, st[j].t, (st[j].side << 4) | st[j].e).fields >> (SQ_FIELD * st[j].field)) & 4u;
            }
        }
        c += seen ? 1 : 0;
    }
    const unsigned long long t = sp_block_sum_u64(c, red);
    if (threadIdx.x == 0 && t) atomicAdd(out, t);
}

// ------------------------------------------------------------------ k5_map_sparse2 (round 5)
// The pair kernel of rounds 1-4 was 532 KB of machine code -- the rolling scan unrolled over 95 bases with the hit path inlined
// at every start -- against 64 KB of instruction cache, and it kept ONE probe in flight per lane.  This is the k5_map2 walk
// (sp_map.hip) for 64-bit keys: the loop over the pairs of a unit stays rolled (32-base windows by run-time shifts out
// of three rotating registers per stream), two pairs travel together (their filter probes, then their table look-ups),
// a hit is a bit in three label planes and a unit's hits are settled once, by popcounts.  Pair-keyed table only
// (S <= 7); the per-k-mer table takes k5_map_sparse_lab above.  TABLE: 0 = the pair-keyed hash table (one look-up per candidate
// pair, S <= 7), 1 = the quad buckets above (one per candidate quad, S <= 3; `htab` / `hmask` unused).
// Round 6: two phases, as map_unit_scan64 (sp_map.hip): the filter leaves the wave's candidate quads in an LDS queue, the
// look-ups are dealt out to all lanes (a lane rebuilds the windows of the quad it was dealt from the owner's six packed words
// in LDS and ORs the labels it finds into the owner's planes).
struct map_pair_win {
    uint64_t xf, xr;       // the shared (k-1)-mer of a pair of starts, forward / reverse complement
    uint32_t b0, b1;       // the base in front of it / behind it
};
// the pair whose first start is base rr (0 .. 14) of the word la: the 32-base window lies in la, lb, lc
__device__ __forceinline__ map_pair_win map_pair_window(uint32_t la, uint32_t lb, uint32_t lc, uint32_t ma, uint32_t mb, uint32_t mc,
                                                        int rr, int k, int sh, uint64_t m1mask) {
    map_pair_win q;
    const uint64_t W = (uint64_t)__builtin_amdgcn_alignbit(lb, la, 2 * rr) | ((uint64_t)__builtin_amdgcn_alignbit(lc, lb, 2 * rr) << 32);
    const unsigned long long mA = ((unsigned long long)ma << 32) | mb, mB = ((unsigned long long)mb << 32) | mc;
    const uint64_t V = (((mA << (2 * rr)) >> 32) << 32) | ((mB << (2 * rr)) >> 32);
    q.b0 = (uint32_t)(V >> 62);            // the base in front of x (first base of the pair's first k-mer)
    q.xf = (V >> sh) & m1mask;             // x forward (the k-mer at the pair's first start without its first base)
    q.xr = (~W >> 2) & m1mask;             // its reverse complement
    // the base behind x (last base of the pair's second k-mer): position rr + k of the words, 16 <= k <= 32: in lb or lc
    const int pb = rr + k;
    q.b1 = (((pb >> 4) == 1 ? lb : lc) >> (2 * (pb & 15))) & 3u;
    return q;
}
template <int TABLE>
__device__ __forceinline__ void map_unit_scan64_h(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm,
                                                  const uint32_t *__restrict__ nm, int64_t s0, const sp_kparams &kp,
                                                  const uint32_t *__restrict__ bloom, int nbits,
                                                  unsigned long long *__restrict__ htab, uint64_t hmask, const sq_tab &T,
                                                  unsigned long long lab[3], const map_unit_lds &U,
                                                  unsigned long long cm = ~0ULL /* starts that count */) {
    constexpr int FW = TABLE ? SQ_FIELD : 4;
    constexpr uint32_t LBL = TABLE ? 3u : 7u, SEEN = TABLE ? 4u : 8u, FMASK = TABLE ? 7u : 15u;
    constexpr uint32_t ANY = TABLE ? (uint32_t)SQ_ANY : 0x77777777u;
    constexpr int NP = TABLE ? 2 : 3, ROUND_W = TABLE ? MAP2_ROUND_W : 2;
    unsigned long long ok_k, ok_x;      // k-mer at s0+j valid; shared (k-1)-mer at s0+j+1 valid
    {
        const uint64_t badA = sp_bad_starts64(nm, s0, kp.k - 1), badB = sp_bad_starts64(nm, s0 + 32, kp.k - 1);
        const uint64_t invA = (uint64_t)nm[s0 >> 5] | ((uint64_t)nm[(s0 >> 5) + 1] << 32);
        const uint64_t invB = (uint64_t)nm[(s0 >> 5) + 1] | ((uint64_t)nm[(s0 >> 5) + 2] << 32);
        const uint32_t kA = ~(uint32_t)(badA | (invA >> (kp.k - 1))), kB = ~(uint32_t)(badB | (invB >> (kp.k - 1)));
        const uint32_t xA = ~(uint32_t)(badA >> 1), xB = ~(uint32_t)(badB >> 1);
        ok_k = ((unsigned long long)kA | ((unsigned long long)kB << 32)) & cm;      // (not covered: neither counted nor marked seen)
        ok_x = (unsigned long long)xA | ((unsigned long long)xB << 32);
    }
    if (__all((ok_x & 0x5555555555555555ULL) == 0)) return;
    // the lanes of the wave that are here (a range's last wave, a grid-stride loop's last round): they share the candidates
    const unsigned long long here = __ballot(1);
    const int n_here = __popcll(here);
    const int me = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
    const int tid = (int)threadIdx.x, lane = tid & 63, wave0 = tid & ~63;
    uint32_t *sw = U.words + tid;
    uint16_t *queue = U.queue + (tid >> 6) * MAP_QCAP(TABLE);
    const int64_t w0 = s0 >> 4;   // a multiple of 4: 16-byte aligned
    const uint4 la = *reinterpret_cast<const uint4 *>(pk + w0);
    uint32_t l0 = la.x, l1 = la.y, l2 = la.z, l3 = la.w, l4 = pk[w0 + 4], l5 = pk[w0 + 5];
#if SP_DERIVE_PM
    (void)pm;
    uint32_t m0 = sp_msb_of_lsb(l0), m1 = sp_msb_of_lsb(l1), m2 = sp_msb_of_lsb(l2), m3 = sp_msb_of_lsb(l3), m4 = sp_msb_of_lsb(l4),
             m5 = sp_msb_of_lsb(l5);
#else
    const uint4 ma = *reinterpret_cast<const uint4 *>(pm + w0);
    uint32_t m0 = ma.x, m1 = ma.y, m2 = ma.z, m3 = ma.w, m4 = pm[w0 + 4], m5 = pm[w0 + 5];
#endif
    sw[0 * MAP_BLOCK] = l0; sw[1 * MAP_BLOCK] = l1; sw[2 * MAP_BLOCK] = l2; sw[3 * MAP_BLOCK] = l3; sw[4 * MAP_BLOCK] = l4; sw[5 * MAP_BLOCK] = l5;
    sw[6 * MAP_BLOCK] = (uint32_t)ok_k; sw[7 * MAP_BLOCK] = (uint32_t)(ok_k >> 32);
#pragma unroll
    for (int bit = 0; bit < NP; bit++) U.planes[bit * MAP_BLOCK + tid] = 0ULL;
    const int sh = 64 - 2 * kp.k;
    const uint64_t m1mask = kp.kmask >> 2;
    const bool core = (nbits & MAP_BLOOM_CORE) != 0;                   // (uniform; k >= 16 here)
    const uint64_t cmask = (1ULL << (2 * (kp.k - 3))) - 1ULL;
    const int wsh = 32 - (MAP_BLOOM_NBITS(nbits) - 5);
    uint32_t last_wi = 0xFFFFFFFFu, last_w = 0u;                       // the word this lane fetched last (index, content)
    uint32_t h_carry = 0u;                                              // hash of the core the previous two pairs ended with
#pragma unroll 1
    for (int wr = 0; wr < 4; wr += ROUND_W) {
        // ---- phase 1: the filter over ROUND_W words of sixteen starts
        uint32_t qn = 0;                                                // (uniform)
#pragma unroll 1
        for (int w = wr; w < wr + ROUND_W; w++) {
#pragma unroll 1
            for (int r = 0; r < 16; r += 4) {
                const int j = 16 * w + r;           // two pairs: x1 at j + 1 (starts j, j + 1), x2 at j + 3 (starts j + 2, j + 3)
                const map_pair_win A = map_pair_window(l0, l1, l2, m0, m1, m2, r, kp.k, sh, m1mask);
                const map_pair_win B = map_pair_window(l0, l1, l2, m0, m1, m2, r + 2, kp.k, sh, m1mask);
                uint32_t hx1, hx2;
                const uint32_t bt1 = map_bloom_bits3(A.xf < A.xr ? A.xf : A.xr, hx1), bt2 = map_bloom_bits3(B.xf < B.xr ? B.xf : B.xr, hx2);
                // the filter words of x1 and x2 (sp_map.h): by the smaller-hashed core of the chain a - x1 - b - x2 - c, or by
                // the (k-1)-mer itself; a word the lane fetched for the previous pair is not fetched again
                uint32_t wi1, wi2;
                if (core) {
                    const uint64_t tb_f = A.xf & cmask, tb_r = A.xr >> 4, tc_f = B.xf & cmask, tc_r = B.xr >> 4;
                    uint32_t ha = h_carry;              // (this iteration's a IS the previous one's c: the same bases)
                    if (j == 0) {                       // (uniform: the unit's first two pairs)
                        const uint64_t ta_f = A.xf >> 4, ta_r = A.xr & cmask;
                        ha = map_core_hash(ta_f < ta_r ? ta_f : ta_r);
                    }
                    const uint32_t hb = map_core_hash(tb_f < tb_r ? tb_f : tb_r), hc = map_core_hash(tc_f < tc_r ? tc_f : tc_r);
                    h_carry = hc;
                    wi1 = map_core_word(ha < hb ? ha : hb, nbits);
                    wi2 = map_core_word(hb < hc ? hb : hc, nbits);
                } else {
                    wi1 = hx1 >> wsh;
                    wi2 = hx2 >> wsh;
                }
                const bool v1 = (ok_x >> j) & 1ULL, v2 = (ok_x >> (j + 2)) & 1ULL;
                const bool need1 = v1 && wi1 != last_wi;
                const bool need2 = v2 && wi2 != (v1 ? wi1 : last_wi);
                uint32_t f1 = 0, f2 = 0;
                if (need1) f1 = bloom[wi1];
                if (need2) f2 = bloom[wi2];
                const uint32_t wd1 = v1 ? (need1 ? f1 : last_w) : 0u;
                const uint32_t wd2 = v2 ? (need2 ? f2 : (v1 ? wd1 : last_w)) : 0u;
                if (v2) { last_wi = wi2; last_w = wd2; }
                else if (v1) { last_wi = wi1; last_w = wd1; }
                const bool cand1 = (wd1 & bt1) == bt1, cand2 = (wd2 & bt2) == bt2;      // (an invalid pair's word is 0)
                const unsigned long long bal = __ballot(cand1 || cand2);
                if (cand1 || cand2)
                    queue[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] =
                        (uint16_t)((uint32_t)lane | ((uint32_t)(j >> 2) << 6) | (cand1 ? 0x400u : 0u) | (cand2 ? 0x800u : 0u));
                qn += (uint32_t)__popcll(bal);
            }
            l0 = l1; l1 = l2; l2 = l3; l3 = l4; l4 = l5;
            m0 = m1; m1 = m2; m2 = m3; m3 = m4; m4 = m5;
        }
        // ---- phase 2: the queue, dealt out to the lanes that are here (LDS operations of a wave complete in program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t e0 = (uint32_t)me; e0 < qn; e0 += (uint32_t)n_here) {
            const uint32_t ent = queue[e0];
            const int owner = wave0 + (int)(ent & 63u), qi = (int)((ent >> 6) & 15u), j = 4 * qi, w = qi >> 2, r = j & 15;
            const bool cand[2] = {(ent & 0x400u) != 0u, (ent & 0x800u) != 0u};
            const uint32_t *ow = U.words + owner;
            const uint32_t oa = ow[w * MAP_BLOCK], ob = ow[(w + 1) * MAP_BLOCK], oc = ow[(w + 2) * MAP_BLOCK];
            const uint32_t okq = ow[(6 + (w >> 1)) * MAP_BLOCK] >> (j & 31);       // countable starts j .. j + 3 of the owner's unit
            const uint32_t pa = sp_msb_of_lsb(oa), pb_ = sp_msb_of_lsb(ob), pc = sp_msb_of_lsb(oc);
            const map_pair_win Pw[2] = {map_pair_window(oa, ob, oc, pa, pb_, pc, r, kp.k, sh, m1mask),
                                        map_pair_window(oa, ob, oc, pa, pb_, pc, r + 2, kp.k, sh, m1mask)};
            uint32_t e[2] = {0u, 0u};
            uint32_t loc[2] = {0u, 0u};                               // TABLE: where a hit's "seen" bits go (sq_mark)
            uint64_t slot[2] = {0, 0};                                // hash table: the pair's slot
            bool fwd_[2] = {Pw[0].xf <= Pw[0].xr, Pw[1].xf <= Pw[1].xr};          // orientation the fields are laid out in
            if (TABLE) {
                // the core the two pairs share: last k-3 bases of x1 = first k-3 bases of x2 (a candidate's (k-1)-mer is valid, so it is)
                const uint64_t s_f = cand[0] ? (Pw[0].xf & cmask) : (Pw[1].xf >> 4), s_r = cand[0] ? (Pw[0].xr >> 4) : (Pw[1].xr & cmask);
                const bool sfw = s_f <= s_r;
                const uint64_t t = sfw ? s_f : s_r;
                const uint64_t hm = sq_mix(t, T.sb);
                const uint64_t bucket = hm >> T.tb, tagb = (hm & ((1ULL << T.tb) - 1ULL)) << 5;
                // x1 = e + s: side L read forward, side R (e reverse-complemented) read backward; x2 = s + e: the mirror image
                const uint32_t se1 = (sfw ? 0u : 16u) | (uint32_t)(sfw ? (Pw[0].xf >> T.sb) : (Pw[0].xr & 15ULL));
                const uint32_t se2 = (sfw ? 16u : 0u) | (uint32_t)(sfw ? (Pw[1].xf & 15ULL) : (Pw[1].xr >> T.sb));
                const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(T.buckets + 4 * bucket);
                const ulonglong2 B0 = bp[0], B1 = bp[1];
                if (cand[0]) {
                    const sq_hit hh = sq_find(T, B0, B1, bucket, tagb | se1, t, se1);
                    e[0] = hh.fields;
                    loc[0] = hh.loc;
                }
                if (cand[1]) {
                    const sq_hit hh = sq_find(T, B0, B1, bucket, tagb | se2, t, se2);
                    e[1] = hh.fields;
                    loc[1] = hh.loc;
                }
                fwd_[0] = fwd_[1] = sfw;      // the fields are laid out in the orientation in which t is canonical
            } else {
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (cand[h]) e[h] = sps_pair_get(Pw[h].xf < Pw[h].xr ? Pw[h].xf : Pw[h].xr, htab, hmask, slot[h]);
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (!(e[h] & ANY)) continue;
                const bool fw = fwd_[h];
                const int f0 = fw ? (int)Pw[h].b0 : 7 - (int)Pw[h].b0, f1 = fw ? 4 + (int)Pw[h].b1 : 3 - (int)Pw[h].b1;
                const uint32_t okk = okq >> (2 * h);
                const uint32_t v0 = (okk & 1u) ? (e[h] >> (FW * f0)) & FMASK : 0u;
                const uint32_t v1 = (okk & 2u) ? (e[h] >> (FW * f1)) & FMASK : 0u;
                const uint32_t two = (v0 & LBL) | ((v1 & LBL) << 8);
                if (!two) continue;
#pragma unroll
                for (int bit = 0; bit < NP; bit++) {
                    const unsigned long long b2 = (unsigned long long)(((two >> bit) & 1u) | (((two >> (8 + bit)) & 1u) << 1));
                    if (b2) atomicOr(&U.planes[bit * MAP_BLOCK + owner], b2 << (j + 2 * h));
                }
                uint32_t mark = 0;       // "seen": first touch only
                if ((v0 & LBL) && !(v0 & SEEN)) mark |= SEEN << (FW * f0);
                if ((v1 & LBL) && !(v1 & SEEN)) mark |= SEEN << (FW * f1);
                if (mark) {
                    if (TABLE) sq_mark(T, loc[h], mark);
                    else atomicOr(&htab[2 * slot[h] + 1], (unsigned long long)mark);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int bit = 0; bit < NP; bit++) lab[bit] = U.planes[bit * MAP_BLOCK + tid];
}

// (six waves per SIMD = TWO 768-thread workgroups per CU: at 82 VGPRs -- one more than that allows -- the kernel ran one: 52.6 -> 66.6 ms)
template <int TABLE>
__global__ void __launch_bounds__(MAP_BLOCK, 6)
k5_map_sparse2(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm, sp_kparams kp,
               sp_map_params P, unsigned long long *__restrict__ htab, uint64_t hmask, sq_tab T, const uint32_t *__restrict__ bloom,
               int bloom_bits, int *__restrict__ slot_counts, unsigned long long *__restrict__ n_mapped) {
    __shared__ int hist[MAP_LDS_ENTRIES];
    __shared__ unsigned long long red[16];
    MAP_UNIT_LDS_DECL(TABLE, 8);
    unsigned long long mapped = 0;
    const int64_t n_ranges = (P.n_units + MAP_BLOCK - 1) / MAP_BLOCK;
    for (int64_t r = blockIdx.x; r < n_ranges; r += gridDim.x) {
        const int64_t u = r * MAP_BLOCK + threadIdx.x;
        // the range's first output slot and where it ends (uniform; once per range, not per hit)
        const int64_t slot_lo = map_slot(r * MAP_RANGE, P, kp.k), end_lo = map_slot_end(r * MAP_RANGE, P, kp.k);
        const bool one_slot = end_lo >= (r + 1) * MAP_RANGE;
        if (P.use_lds) {
            for (int i = threadIdx.x; i < MAP_LDS_ENTRIES; i += MAP_BLOCK) hist[i] = 0;
            __syncthreads();
        }
        if (u < P.n_units) {
            unsigned long long lab[3] = {0ULL, 0ULL, 0ULL};
            map_unit_scan64_h<TABLE>(pk, pm, nm, u * SP_UNIT, kp, bloom, bloom_bits, htab, hmask, T, lab, ulds);
            if (lab[0] | lab[1] | lab[2]) {
                auto add = [&](int64_t os, unsigned long long within) {
                    for (int sg = 0; sg < P.S; sg++) {
                        const int l = sg + 1;
                        const unsigned long long m = ((l & 1) ? lab[0] : ~lab[0]) & ((l & 2) ? lab[1] : ~lab[1]) &
                                                     ((l & 4) ? lab[2] : ~lab[2]) & within;
                        const int v = __popcll(m);
                        if (!v) continue;
                        if (P.use_lds) atomicAdd(&hist[(int)(os - slot_lo) * P.S + sg], v);
                        else if (os < P.nslots) atomicAdd(&slot_counts[os * P.S + sg], v);
                        mapped += v;
                    }
                };
                const int64_t s0 = u * SP_UNIT;
                if (one_slot) {
                    add(slot_lo, ~0ULL);
                } else {
                    int64_t p = s0;
                    while (p < s0 + SP_UNIT) {
                        int64_t e = map_slot_end(p, P, kp.k);
                        if (e > s0 + SP_UNIT) e = s0 + SP_UNIT;
                        const int a = (int)(p - s0), b = (int)(e - s0);
                        const unsigned long long within = (b >= 64 ? ~0ULL : ((1ULL << b) - 1ULL)) & ~((1ULL << a) - 1ULL);
                        if ((lab[0] | lab[1] | lab[2]) & within) add(map_slot(p, P, kp.k), within);
                        p = e;
                    }
                }
            }
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

// feature mode (sp_map.hip: k5_map_feat2), pair-keyed table: the rolled walk over the starts whose k-mer crosses no feature
// boundary, then popcounts of the label planes per overlapping feature
template <int TABLE>
__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_feat_sparse2(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm, sp_kparams kp,
                    int64_t n_units, const int64_t *__restrict__ foff, int64_t n_feat, int S,
                    unsigned long long *__restrict__ htab, uint64_t hmask, sq_tab T,
                    const uint32_t *__restrict__ bloom, int bloom_bits, unsigned long long *__restrict__ counts) {
    MAP_UNIT_LDS_DECL(TABLE, 8);
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int k = kp.k;
    auto span = [](int64_t a, int64_t b) -> unsigned long long {      // bits [a, b) of a unit, 0 <= a, b <= 64
        if (b <= a) return 0ULL;
        return (b >= 64 ? ~0ULL : ((1ULL << b) - 1ULL)) & ~((1ULL << a) - 1ULL);
    };
    for (; u < n_units; u += stride) {
        const int64_t s0 = u * SP_UNIT;
        int64_t lo = 0, hi = n_feat;           // the feature the unit starts in: last f with foff[f] <= s0
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (foff[mid] <= s0) lo = mid;
            else hi = mid;
        }
        unsigned long long fit = ~0ULL;        // a boundary at e = foff[f + 1] disqualifies the starts (e - k, e)
        for (int64_t f = lo; f < n_feat; f++) {
            const int64_t e = foff[f + 1];
            if (e - k + 1 >= s0 + SP_UNIT) break;
            fit &= ~span((e - k + 1 > s0 ? e - k + 1 : s0) - s0, (e < s0 + SP_UNIT ? e : s0 + SP_UNIT) - s0);
            if (e >= s0 + SP_UNIT) break;
        }
        unsigned long long lab[3] = {0ULL, 0ULL, 0ULL};
        map_unit_scan64_h<TABLE>(pk, pm, nm, s0, kp, bloom, bloom_bits, htab, hmask, T, lab, ulds, fit);
        if (!(lab[0] | lab[1] | lab[2])) continue;
        for (int64_t f = lo; f < n_feat; f++) {
            const int64_t a = foff[f], e = foff[f + 1];
            if (a >= s0 + SP_UNIT) break;
            const unsigned long long within = span((a > s0 ? a : s0) - s0, (e < s0 + SP_UNIT ? e : s0 + SP_UNIT) - s0) & fit;
            if (!((lab[0] | lab[1] | lab[2]) & within)) continue;
            for (int sg = 0; sg < S; sg++) {
                const int l = sg + 1;
                const unsigned long long m = ((l & 1) ? lab[0] : ~lab[0]) & ((l & 2) ? lab[1] : ~lab[1]) &
                                             ((l & 4) ? lab[2] : ~lab[2]) & within;
                if (m) atomicAdd(&counts[f * S + sg], (unsigned long long)__popcll(m));
            }
        }
    }
}
// the same for the per-k-mer table (more than 7 subgenomes): the rolling scan, one look-up per candidate start
__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_feat_sparse_lab(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, sp_kparams kp, int64_t n_units,
                       const int64_t *__restrict__ foff, int64_t n_feat, int S,
                       unsigned long long *__restrict__ htab, uint64_t mask,
                       const uint32_t *__restrict__ bloom, int bloom_bits, unsigned long long *__restrict__ counts) {
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; u < n_units; u += stride) {
        map_feat_cursor cur;
        cur.f = -1;
        cur.next = 0;
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
    for (auto &ln : ctx->lanes) {       // the k > 15 counting lanes' workspaces (sp_sparse2.hip)
        sp_buf_free(ln.b_sp_a);
        sp_buf_free(ln.b_sp_b);
        sp_buf_free(ln.b_sp_c);
        sp_buf_free(ln.b_sp_tmp);
        sp_buf_free(ln.b_s3_small);
    }
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

// ------------------------------------------------------------------ list filter: C-way hash join (round 3)
// The first list filter concatenated the C sorted lists, sorted the concatenation by key with a device-wide library
// radix sort and evaluated runs of equal keys: ~220 bytes of HBM traffic per list entry (27 ms per wheat-like pass at
// k = 17, the worst stage of that line).  The lists are sorted already, so the join needs no global sort:
//   sps_bounds   cuts the key space into 2^rb equal ranges and records where every list crosses every range edge;
//   sps_join     one workgroup per range: the C segments (a few hundred entries together) go to LDS, an LDS hash
//                table groups the entries of equal keys (chain per key, owner = the entry of the lowest chromosome),
//                the owner rebuilds the row and takes the decision (the shared sp_filter_decide).  Fold-passing
//                totals go to a staging array at the position the range has in the virtual concatenation (closed
//                form, no atomics); differential rows -- rare -- go to a row staging area handed out in chunks, each
//                with its rank inside the range (ascending key).  A range with more than JOIN_T entries is worked
//                off in rounds of key sub-ranges (pivot = the smallest of the lists' (JOIN_T / C)-th pending keys).
//   sps_place_*  scan of the per-range tallies, then rows / totals move to their final, key-ordered places.
// The lists are read once (12 B per entry) plus once for the range edges (8 B).
#define JOIN_T 1024          // entries per round
#define JOIN_H 2048          // hash slots
#define JOIN_BLOCK 256
#define JOIN_CHUNK 256       // rows handed out per grab of the global row cursor
struct sps_list {
    const unsigned long long *keys;
    const uint32_t *cnts;
    long long n;
};

__global__ void __launch_bounds__(256)
sps_bounds(const sps_list *__restrict__ lists, int shift, long long R, uint32_t *__restrict__ bnd /* C x (R + 1) */) {
    const sps_list L = lists[blockIdx.y];
    uint32_t *b = bnd + (size_t)blockIdx.y * (size_t)(R + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < L.n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = (long long)(L.keys[i] >> shift);
        const long long rp = i ? (long long)(L.keys[i - 1] >> shift) : -1;
        for (long long x = rp + 1; x <= r; x++) b[x] = (uint32_t)i;      // first entry at or beyond the edge of range x
        if (i == L.n - 1)
            for (long long x = r + 1; x <= R; x++) b[x] = (uint32_t)L.n;
    }
}

struct sps_join_args {
    int C, shift;
    long long R;
    sp_fsets F;
    const sps_list *lists;
    const uint32_t *bnd;
    uint32_t *n_rows, *n_hist;            // per range
    unsigned long long *hist_stage;       // [total]: fold-passing totals of range r from the range's first entry on
    unsigned long long *row_cursor;       // rows handed out so far (may exceed row_cap: the surplus is not written)
    unsigned long long row_cap;
    unsigned long long *row_keys, *row_tot;   // row staging: key (all ones = unused), tot, rank inside the range, counts
    uint32_t *row_rank, *row_counts;
    unsigned long long *n_union;
    const unsigned long long *chrom_sets;     // per chromosome: bit s set if it belongs to non-singleton set number s
    int screen;                               // the bit masks are usable (<= 64 non-singleton sets)
    int fast;                                 // every non-singleton set uses baseline 1 or -1 and has no empty unit: the
                                              // uniform fp32 walk of sps_join_blk applies (k3_eval's P.fast)
    const int32_t *rd;                        // its row descriptors: chromosome | JD_UNIT_END | JD_SET_END | JD_BI1, the
    const float *rinv;                        // non-singleton sets in config order; 1 / (unit length) at unit ends, fp32
    int n_rd;
};

// wave-level helpers of the join (wave 0 of a workgroup runs the cursor logic: lane c owns list c)
template <typename T>
__device__ __forceinline__ T jw_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long jw_min(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long x = __shfl_xor(v, o, 64);
        v = x < v ? x : v;
    }
    return v;
}

// ------------------------------------------------------------------ sps_join_blk (round 5): the join, one WORKGROUP per range
// (Rounds 3-4 ran one WAVE per ~100-entry range; that kernel was the `SP_LIST_FILTER=wave` cross-check until round 6 and is gone:
// the independent check of the list filter is `SP_LIST_FILTER=sort`, the device-wide sort + run evaluation.)
// The wave-per-range join read ~90 bytes per list entry for 12 useful ones (PMC, round 4): a range of ~100 entries is
// 21 list segments of ~5 entries -- one or two 64-byte lines of keys and one of counts per segment, used to a tenth --
// plus two strided range-edge words per list and range.  With ranges of ~BJ_T / 1.5 entries (a segment is ~35 entries:
// four lines of keys, used in full) the fixed costs -- edges, cursors, hash clear, the tallies -- are paid once per ~700
// entries instead of once per ~100.  A workgroup of BJ_THREADS takes a range: wave 0 runs the cursor logic of the
// wave kernel (lane c owns list c: pivot, share, offsets by shuffles) and publishes it through LDS, every thread loads
// and hash-inserts BJ_T / BJ_THREADS entries.  Same outputs, same staging protocol (hist totals at the range's
// closed-form position, rows in chunks with their rank inside the range), so sps_tally_* / sps_place_* are unchanged.
//
// Nothing after the hash build walks a chain (second half of round 5).  The first version kept the wave kernel's
// per-key chains: the owner of a key walked them for the screen (set mask, total), again to rebuild its row for the
// decision, again to write a kept row -- dependent LDS reads, ~20 deep for exactly the k-mers that pass the screen, one
// lane busy while 63 wait.  Bound experiments on the peanut-like genome (5.2 ms): no decisions 2.3 ms, nothing after the
// hash build 1.2 ms.  Now every ENTRY works for its owner, all in parallel: it adds its set bit and count to the owner's
// tallies (two LDS atomics), writes its count into the owner's row when the owner is up for a decision or kept, and the
// decision itself is k3_eval's uniform row walk over a descriptor list (chromosome | unit end | set end, reciprocal
// lengths) -- the same instructions in every lane, independent LDS reads.
#ifndef BJ_T
#define BJ_T 1024         // entries per round
#endif
#define BJ_H (2 * BJ_T)   // hash slots
#ifndef BJ_THREADS
#define BJ_THREADS 256
#endif
#define BJ_WAVES (BJ_THREADS / 64)
#define BJ_Q (BJ_T / BJ_THREADS)
#define BJ_ROWS 16        // rows a wave decides at a time
#define BJ_NR (BJ_WAVES * BJ_ROWS)
#define JD_CHROM_MASK 0xfffff
#define JD_UNIT_END (1 << 20)
#define JD_SET_END (1 << 21)
#define JD_BI1 (1 << 22)      // the set's baseline is the second largest frequency (else the smallest)
// LDS of a workgroup: 31.6 KB + the rows (32-bit residuals) -- FOUR workgroups per CU.  The kernel's time follows its occupancy
// (two / three workgroups per CU: 3.61 / 2.70 ms on the peanut-like genome, 15.3 / 10.9 at wheat-like k = 21), so two tables share
// their space with the two that are dead by the time they are needed: the per-owner totals live where the hash keys were (no key
// is compared after the last insert), the per-owner set masks / queue places / ranks where the slot owners were (every entry
// copies its owner's index into Sl[] first).
#define BJ_FC 128         // row descriptors of the uniform walk held in LDS (more: generic decisions)
template <typename RT>
struct bj_lds {
    alignas(8) RT Hk[BJ_H];               // hash keys; after the inserts: Et[BJ_T], per owner entry the sum of the key's counts
    uint32_t Hmin[BJ_H];                  // per slot: (chromosome << 16 | entry) of the owner; after the owner pass: Es[BJ_T], per
                                          // owner entry the set mask -> place in the decision queue -> rank among the kept rows
    RT Kk[BJ_T];
    uint32_t Vv[BJ_T];
    uint16_t Sl[BJ_T], PQ[BJ_T];          // Sl: hash slot, then the owner's entry; PQ: decision queue, then the list of kept rows
    uint8_t Ch[BJ_T];                     // chromosome (6 bits) | kept row << 6 | fold-passing << 7 (owners, after the decision)
    int32_t rd[BJ_FC];
    float rinv[BJ_FC];
    uint32_t seg_off[SPS_MAXC + 2], cur[SPS_MAXC];
    const unsigned long long *keys[SPS_MAXC];
    const uint32_t *cnts[SPS_MAXC];
    uint32_t T, more, n_hist, n_row, n_pend;
    unsigned long long hist_pos, chunk_pos;
};
static_assert(sizeof(unsigned long long) * BJ_T <= sizeof(uint32_t) * BJ_H, "Et fits where the 32-bit hash keys were");
static_assert(SPS_MAXC <= 64, "six bits of Ch[] hold the chromosome");

template <typename RT>
__global__ void __launch_bounds__(BJ_THREADS)
sps_join_blk(sps_join_args A) {
    __shared__ bj_lds<RT> L;
    unsigned long long *const Et = reinterpret_cast<unsigned long long *>(L.Hk);
    uint32_t *const Es = L.Hmin;
    extern __shared__ uint32_t jw_rows[];      // [BJ_NR][C | 1]: rows being decided / written
    __shared__ uint32_t s_csets[SPS_MAXC];
    const int C = A.C, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int Cs = C | 1;                      // row stride in words (odd: consecutive rows start in different banks)
    const sp_fsets F = A.F;                    // (global memory: only the generic decision reads it)
    {
        for (int i = threadIdx.x; i < A.n_rd; i += blockDim.x) {
            L.rd[i] = A.rd[i];
            L.rinv[i] = A.rinv[i];
        }
        for (int i = threadIdx.x; i < C; i += blockDim.x) {
            s_csets[i] = (uint32_t)A.chrom_sets[i];
            L.keys[i] = A.lists[i].keys;
            L.cnts[i] = A.lists[i].cnts;
        }
        __syncthreads();
    }
    const RT EMPTY = (RT)~(RT)0;
    const unsigned long long rmask = A.shift >= 64 ? ~0ULL : ((1ULL << A.shift) - 1ULL);
    const unsigned long long *my_keys = nullptr;      // wave 0, lane c: list c
    if (w == 0 && lane < C) my_keys = A.lists[lane].keys;
    unsigned long long uni = 0;
    unsigned long long chunk_pos = 0, chunk_end = 0;   // block-uniform (every thread keeps the same copy)
    const uint32_t per = BJ_T / (uint32_t)C;
    const float fold32 = (float)F.min_fold;
    // the edges of the NEXT range of this workgroup travel while the current one is joined (unconditional, clamped)
    uint32_t n_cur = 0, n_endp = 0;
    auto edges = [&](long long r) {
        if (w == 0 && lane < C) {
            const long long rc = r < A.R ? r : A.R - 1;
            n_cur = A.bnd[(size_t)lane * (size_t)(A.R + 1) + (size_t)rc];
            n_endp = A.bnd[(size_t)lane * (size_t)(A.R + 1) + (size_t)rc + 1];
        }
    };
    // rows [0, n) of jw_rows <- the counts of the entries whose owner's Es lies in [first, first + n)
    auto build_rows = [&](uint32_t T, uint32_t first, uint32_t n) {
        for (uint32_t i = threadIdx.x; i < n * (uint32_t)Cs; i += BJ_THREADS) jw_rows[i] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < BJ_Q; q++) {
            const uint32_t e = threadIdx.x + BJ_THREADS * q;
            if (e < T) {
                const uint32_t ri = Es[L.Sl[e]] - first;
                if (ri < n) jw_rows[ri * (uint32_t)Cs + (L.Ch[e] & 63u)] = L.Vv[e];
            }
        }
        __syncthreads();
    };
    edges(blockIdx.x);
    for (long long r = blockIdx.x; r < A.R; r += gridDim.x) {
        uint32_t cur = n_cur, endp = n_endp;           // wave 0 only
        edges(r + gridDim.x);
        if (w == 0) {
            const unsigned long long hp = jw_sum((unsigned long long)cur);   // the range's place in the virtual concatenation
            if (lane == 0) L.hist_pos = hp;
        }
        const unsigned long long hi_bits = A.shift >= 64 ? 0ULL : ((unsigned long long)r << A.shift);
        uint32_t rows_before = 0, hist_before = 0;     // block-uniform
        for (;;) {
            uint32_t take = 0;
            if (w == 0) {
                // ---- the round's share of every list: everything, or everything below the pivot key
                const uint32_t left = endp - cur;
                unsigned long long pivot = SPS_SENTINEL;
                if (jw_sum(left) > BJ_T) pivot = jw_min((lane < C && left > per) ? my_keys[cur + per] : SPS_SENTINEL);
                take = left;
                if (pivot != SPS_SENTINEL && lane < C) {     // entries below the pivot: at most `per` (the per-th is >= pivot)
                    const unsigned long long *kk = my_keys + cur;
                    uint32_t lo = 0, hi = left < per ? left : per;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (kk[mid] < pivot) lo = mid + 1;
                        else hi = mid;
                    }
                    take = lo;
                }
                uint32_t incl = take;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t x = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += x;
                }
                const uint32_t T0 = __shfl(incl, 63, 64);                 // <= BJ_T by construction
                const bool more0 = __any(cur + take < endp);
                if (lane <= C) L.seg_off[lane] = lane < C ? incl - take : T0;
                if (lane < C) L.cur[lane] = cur;
                if (lane == 0) {
                    L.T = T0;
                    L.more = more0 ? 1u : 0u;
                    L.n_hist = 0;
                    L.n_row = 0;
                    L.n_pend = 0;
                }
            }
            for (uint32_t i = threadIdx.x; i < BJ_H; i += BJ_THREADS) {
                L.Hk[i] = EMPTY;
                L.Hmin[i] = 0xFFFFFFFFu;
            }
            __syncthreads();
            const uint32_t T = L.T;
            const bool more = L.more != 0;
            uint32_t Hn = 64;
            while (Hn < 2 * T) Hn <<= 1;
            // ---- load + hash-insert (owner of a key = its entry of the lowest chromosome)
#pragma unroll
            for (int q = 0; q < BJ_Q; q++) {
                const uint32_t e = threadIdx.x + BJ_THREADS * q;
                if (e < T) {
                    int lo = 0, hi = C;       // list of entry e: last c with seg_off[c] <= e
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (L.seg_off[mid] <= e) lo = mid;
                        else hi = mid;
                    }
                    const int c = lo;
                    const size_t i = (size_t)L.cur[c] + (e - L.seg_off[c]);
                    const RT res = (RT)(L.keys[c][i] & rmask);
                    L.Kk[e] = res;
                    L.Vv[e] = L.cnts[c][i];
                    L.Ch[e] = (uint8_t)c;
                    uint32_t h = (uint32_t)sps_mix((uint64_t)res) & (Hn - 1);
                    for (;;) {
                        const RT prev = atomicCAS(&L.Hk[h], EMPTY, res);
                        if (prev == EMPTY || prev == res) break;
                        h = (h + 1) & (Hn - 1);
                    }
                    L.Sl[e] = (uint16_t)h;
                    atomicMin(&L.Hmin[h], ((uint32_t)c << 16) | e);
                }
            }
            __syncthreads();
            // ---- every entry learns its owner (Sl: slot -> owner's entry); the totals' space is dead hash keys by now
#pragma unroll
            for (int q = 0; q < BJ_Q; q++) {
                const uint32_t e = threadIdx.x + BJ_THREADS * q;
                if (e < T) L.Sl[e] = (uint16_t)(L.Hmin[L.Sl[e]] & 0xFFFFu);
                Et[e] = 0;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < BJ_Q; q++) Es[threadIdx.x + BJ_THREADS * q] = 0;      // (where the slot owners were)
            __syncthreads();
            // ---- every entry adds itself to its owner's tallies
#pragma unroll
            for (int q = 0; q < BJ_Q; q++) {
                const uint32_t e = threadIdx.x + BJ_THREADS * q;
                if (e < T) {
                    const uint32_t o = L.Sl[e];
                    if (A.screen) atomicOr(&Es[o], s_csets[L.Ch[e]]);
                    atomicAdd(&Et[o], (unsigned long long)L.Vv[e]);
                }
            }
            __syncthreads();
            // ---- owners: the union tally, the screen; the ones that pass queue up for the full decision
#pragma unroll
            for (int q = 0; q < BJ_Q; q++) {
                const uint32_t e = threadIdx.x + BJ_THREADS * q;
                bool pending = false;
                if (e < T && L.Sl[e] == e) {
                    uni++;
                    pending = !A.screen || !((double)__popc(Es[e]) / (double)F.n_multi < F.ratio);
                    Es[e] = 0xFFFFFFFFu;     // (no row)
                }
                const unsigned long long pb = __ballot(pending);
                if (pb) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&L.n_pend, (uint32_t)__popcll(pb));
                    base = __shfl(base, 0, 64);
                    if (pending) {
                        const uint32_t p = base + __popcll(pb & ((1ULL << lane) - 1ULL));
                        L.PQ[p] = (uint16_t)e;
                        Es[e] = p;
                    }
                }
            }
            __syncthreads();
            // ---- decisions, BJ_NR rows at a time
            const uint32_t n_p = L.n_pend;
            for (uint32_t p0 = 0; p0 < n_p; p0 += BJ_NR) {
                const uint32_t n = n_p - p0 < BJ_NR ? n_p - p0 : BJ_NR;
                build_rows(T, p0, n);
                const uint32_t ri = (uint32_t)lane * BJ_WAVES + (uint32_t)w;     // the rows spread evenly over the waves
                if (lane < BJ_ROWS && ri < n) {
                    const uint32_t e = L.PQ[p0 + ri];
                    const uint32_t *row = jw_rows + ri * (uint32_t)Cs;
                    const unsigned long long tot = Et[e];
                    bool r_ = false, h_ = false, generic = !A.fast;
                    if (A.fast) {
                        // _filter_kmer (Jellyfish.py:611-648) for baseline 1 / -1, as in k3_eval: running max, second max
                        // and min of the unit frequencies in fp32 on reciprocal products; a k-mer with a set inside the
                        // 1e-5 band around the threshold takes the generic code (fp64 quotients, the reference's order)
                        int include = 0;
                        unsigned long long num = 0;
                        float m1 = -1.0f, m2 = -1.0f, mn = 3e38f;
                        for (int j = 0; j < A.n_rd; j++) {     // (A.n_rd <= BJ_FC: host)
                            const int d = L.rd[j];                   // (uniform)
                            num += row[d & JD_CHROM_MASK];
                            if (d & JD_UNIT_END) {
                                const float x = (float)num * L.rinv[j];
                                m2 = fmaxf(m2, fminf(m1, x));
                                m1 = fmaxf(m1, x);
                                mn = fminf(mn, x);
                                num = 0;
                            }
                            if (d & JD_SET_END) {
                                const float thr = fold32 * (((d & JD_BI1) ? m2 : mn) + 1e-20f);
                                const bool pass = m1 > thr * (1.0f + 1e-5f);
                                include += pass ? 1 : 0;
                                generic = generic || (!pass && !(m1 < thr * (1.0f - 1e-5f)));
                                m1 = -1.0f; m2 = -1.0f; mn = 3e38f;
                            }
                        }
                        if (!generic && !((double)include / (double)F.n_multi < F.ratio)) {   // :642-644
                            h_ = true;
                            const double t = (double)tot;
                            r_ = !(t < F.min_freq || t > F.max_freq);                          // :645-646
                        }
                    }
                    if (generic) sp_filter_decide([&](int c) -> uint32_t { return row[c]; }, tot, F, r_, h_);
                    L.Ch[e] = (uint8_t)(L.Ch[e] | (r_ ? 0x40 : 0) | (h_ ? 0x80 : 0));
                }
                __syncthreads();
            }
            // ---- fold-passing totals out; kept rows listed
            bool is_row[BJ_Q];
#pragma unroll
            for (int q = 0; q < BJ_Q; q++) {
                const uint32_t e = threadIdx.x + BJ_THREADS * q;
                const uint32_t fl = e < T ? (uint32_t)L.Ch[e] >> 6 : 0u;      // (set for owners only)
                is_row[q] = (fl & 1u) != 0;
                const bool is_hist = (fl & 2u) != 0;
                // fold-passing totals: range start + tally so far + a place of the wave's in this round (any order)
                const unsigned long long bh = __ballot(is_hist);
                if (bh) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&L.n_hist, (uint32_t)__popcll(bh));
                    base = __shfl(base, 0, 64);
                    if (is_hist)
                        A.hist_stage[L.hist_pos + hist_before + base + __popcll(bh & ((1ULL << lane) - 1ULL))] = Et[e];
                }
            }
            __syncthreads();       // (PQ is the decision queue no longer)
#pragma unroll
            for (int q = 0; q < BJ_Q; q++) {
                const uint32_t e = threadIdx.x + BJ_THREADS * q;
                const unsigned long long br = __ballot(is_row[q]);
                if (br) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&L.n_row, (uint32_t)__popcll(br));
                    base = __shfl(base, 0, 64);
                    if (is_row[q]) L.PQ[base + __popcll(br & ((1ULL << lane) - 1ULL))] = (uint16_t)e;
                }
            }
            __syncthreads();
            const uint32_t nrow = L.n_row, nh = L.n_hist;
            if (nrow) {       // rare: rows to the staging area, ranked by key inside the round
                if (chunk_pos + nrow > chunk_end) {      // block-uniform
                    const unsigned long long grab = nrow > JOIN_CHUNK ? nrow : JOIN_CHUNK;
                    if (threadIdx.x == 0) L.chunk_pos = atomicAdd(A.row_cursor, grab);
                    __syncthreads();
                    chunk_pos = L.chunk_pos;
                    chunk_end = chunk_pos + grab;
                }
                // Es: rank among the round's kept rows; every other owner out of the way (the ones that were decided still
                // hold their queue place)
#pragma unroll
                for (int q = 0; q < BJ_Q; q++) {
                    const uint32_t e = threadIdx.x + BJ_THREADS * q;
                    if (e < T && !is_row[q] && L.Sl[e] == e) Es[e] = 0xFFFFFFFFu;
                }
#pragma unroll
                for (int q = 0; q < BJ_Q; q++) {
                    if (!is_row[q]) continue;
                    const uint32_t e = threadIdx.x + BJ_THREADS * q;
                    const RT res = L.Kk[e];
                    uint32_t rank = 0;
                    for (uint32_t j = 0; j < nrow; j++) rank += L.Kk[L.PQ[j]] < res;
                    Es[e] = rank;
                    const unsigned long long pos = chunk_pos + rank;
                    if (pos < A.row_cap) {
                        A.row_keys[pos] = hi_bits | (unsigned long long)res;
                        A.row_tot[pos] = Et[e];
                        A.row_rank[pos] = rows_before + rank;
                    }
                }
                __syncthreads();
                for (uint32_t r0 = 0; r0 < nrow; r0 += BJ_NR) {
                    const uint32_t n = nrow - r0 < BJ_NR ? nrow - r0 : BJ_NR;
                    build_rows(T, r0, n);
                    for (uint32_t i = threadIdx.x; i < n * (uint32_t)C; i += BJ_THREADS) {
                        const uint32_t rr = i / (uint32_t)C, c = i - rr * (uint32_t)C;
                        const unsigned long long pos = chunk_pos + r0 + rr;
                        if (pos < A.row_cap) A.row_counts[pos * (size_t)C + c] = jw_rows[rr * (uint32_t)Cs + c];
                    }
                    __syncthreads();
                }
                chunk_pos += nrow;
                rows_before += nrow;
            }
            hist_before += nh;
            __syncthreads();      // the round's LDS state is rewritten by the next round / range
            if (!more) break;
            cur += take;
        }
        if (threadIdx.x == 0) {
            A.n_rows[r] = rows_before;
            A.n_hist[r] = hist_before;
        }
    }
    uni = jw_sum(uni);
    if (lane == 0 && uni) atomicAdd(A.n_union, uni);
}

// per-range tallies -> offsets: block sums, one-block scan of the sums, offsets inside every block
#define TALLY_CHUNK 4096
__global__ void __launch_bounds__(256)
sps_tally_sums(const uint32_t *__restrict__ a, long long n, unsigned long long *__restrict__ bsum) {
    __shared__ unsigned long long red[16];
    const long long lo = (long long)blockIdx.x * TALLY_CHUNK, hi = lo + TALLY_CHUNK < n ? lo + TALLY_CHUNK : n;
    unsigned long long v = 0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) v += a[i];
    const unsigned long long t = sp_block_sum_u64(v, red);
    if (threadIdx.x == 0) bsum[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256)
sps_tally_apply(const uint32_t *__restrict__ a, long long n, const unsigned long long *__restrict__ boff,
                const unsigned long long *__restrict__ total, unsigned long long *__restrict__ off /* n + 1 */) {
    __shared__ unsigned long long wsum[16];
    const long long lo = (long long)blockIdx.x * TALLY_CHUNK, hi = lo + TALLY_CHUNK < n ? lo + TALLY_CHUNK : n;
    constexpr int PER = TALLY_CHUNK / 256;
    const long long t0 = lo + (long long)threadIdx.x * PER;
    unsigned long long s = 0;
    for (int j = 0; j < PER; j++)
        if (t0 + j < hi) s += a[t0 + j];
    unsigned long long tot;
    unsigned long long run = boff[blockIdx.x] + sp_block_excl_scan(s, wsum, tot);
    for (int j = 0; j < PER; j++)
        if (t0 + j < hi) {
            off[t0 + j] = run;
            run += a[t0 + j];
        }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = *total;
}

// staged rows -> their final places: offset of the row's range + its rank inside the range (ascending key overall)
__global__ void __launch_bounds__(256)
sps_place_rows(const unsigned long long *__restrict__ row_keys, const unsigned long long *__restrict__ row_tot,
               const uint32_t *__restrict__ row_rank, const uint32_t *__restrict__ row_counts, unsigned long long n_staged,
               int C, int shift, const unsigned long long *__restrict__ row_off, unsigned long long *__restrict__ out_keys,
               uint32_t *__restrict__ out_counts, unsigned long long *__restrict__ out_tot) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_staged) return;
    const unsigned long long key = row_keys[i];
    if (key == SPS_SENTINEL) return;      // the unused tail of a chunk
    const unsigned long long pos = row_off[key >> shift] + row_rank[i];
    out_keys[pos] = key;
    out_tot[pos] = row_tot[i];
    for (int c = 0; c < C; c++) out_counts[pos * (size_t)C + c] = row_counts[i * (size_t)C + c];
}

// fold-passing totals of range r: hist_stage[start of the range ..) -> out[hist_off[r] ..)
__global__ void __launch_bounds__(256)
sps_place_hist(const unsigned long long *__restrict__ hist_stage, const uint32_t *__restrict__ bnd, int C, long long R,
               const uint32_t *__restrict__ n_hist, const unsigned long long *__restrict__ hist_off,
               unsigned long long *__restrict__ out) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const uint32_t m = n_hist[r];
    if (!m) return;
    unsigned long long src = 0;
    for (int c = 0; c < C; c++) src += bnd[(size_t)c * (size_t)(R + 1) + (size_t)r];
    const unsigned long long dst = hist_off[r];
    for (uint32_t j = 0; j < m; j++) out[dst + j] = hist_stage[src + j];
}

static int sps_filter_sort(sp_ctx *ctx, int n_sets, const int32_t *set_off, const int32_t *unit_off,
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

static int sps_filter_join(sp_ctx *ctx, int n_sets, const int32_t *set_off, const int32_t *unit_off,
                           const int32_t *unit_chrom, const std::vector<double> &den, double min_fold, int baseline,
                           double min_freq, double max_freq, double ratio) {
    const int C = sps_C(ctx);
    if (C > SPS_MAXC)
        return sp_fail(ctx, SP_EUNSUP, "list filter (k > 15, or engine 3): at most %d chromosomes supported (got %d)", SPS_MAXC, C);
    int64_t total = 0, longest = 0;
    std::vector<sps_list> hl((size_t)C);
    for (int c = 0; c < C; c++) {
        hl[(size_t)c] = sps_list{sps_keys(ctx, c), sps_cnts(ctx, c), (long long)sps_n(ctx, c)};
        total += sps_n(ctx, c);
        longest = sps_n(ctx, c) > longest ? sps_n(ctx, c) : longest;
    }
    ctx->sf_n = total;
    ctx->n_union = ctx->n_rows = ctx->n_hist = 0;
    if (total == 0) {
        ctx->filtered = true;
        return SP_OK;
    }
    if (longest >= (1LL << 32)) return sp_fail(ctx, SP_EUNSUP, "list filter: a list of 2^32 or more k-mers");
    // key ranges: 2^rb of them, ~100 entries each (one round of one wave)
    int bits = 2 * ctx->k;
    if (ctx->list_mode) {
        bits = 0;
        while ((1LL << bits) < ctx->nslots) bits++;
    }
    if (bits > 64) bits = 64;
    // one workgroup per range of ~2/3 of a round
    const int64_t per_range = (int64_t)BJ_T * 2 / 3;
    int rb = 0;
    while (rb < bits && rb < 23 && ((int64_t)1 << rb) * per_range < total) rb++;
    if (bits - rb > 63) rb = bits - 63;      // k = 32 and a handful of k-mers: `key >> 64` is not a shift (fuzz case k32_join)
    const long long R = 1LL << rb;
    const int shift = bits - rb;
    const int n_units = set_off[n_sets], n_uc = unit_off[n_units];
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // b_sp_a: list descriptors | range edges | per-range tallies | offsets      b_sp_b: staging of the totals
    const size_t o_lists = 0, o_bnd = al((size_t)C * sizeof(sps_list)), o_nr = o_bnd + al((size_t)C * (size_t)(R + 1) * 4),
                 o_nh = o_nr + al((size_t)R * 4), o_roff = o_nh + al((size_t)R * 4), o_hoff = o_roff + al((size_t)(R + 1) * 8),
                 o_set = o_hoff + al((size_t)(R + 1) * 8), o_uo = o_set + al((size_t)(n_sets + 1) * 4),
                 o_uc = o_uo + al((size_t)(n_units + 1) * 4), o_den = o_uc + al((size_t)(n_uc + 1) * 4),
                 o_small = o_den + al((size_t)n_units * 16), o_cs = o_small + 256, o_bs = o_cs + al((size_t)C * 8),
                 o_rd = o_bs + 2 * al((size_t)(R / TALLY_CHUNK + 2) * 8), o_rinv = o_rd + al((size_t)BJ_FC * 4),
                 a_bytes = o_rinv + al((size_t)BJ_FC * 4);
    int rc = sp_buf_ensure(ctx, ctx->b_sp_a, (int64_t)a_bytes);
    if (rc) return rc;
    rc = sp_buf_ensure(ctx, ctx->b_sp_b, total * 8 + 64);
    if (rc) return rc;
    char *A0 = (char *)ctx->b_sp_a.p;
    const sps_list *d_lists = (const sps_list *)(A0 + o_lists);
    uint32_t *bnd = (uint32_t *)(A0 + o_bnd), *n_rows = (uint32_t *)(A0 + o_nr), *n_hist = (uint32_t *)(A0 + o_nh);
    unsigned long long *row_off = (unsigned long long *)(A0 + o_roff), *hist_off = (unsigned long long *)(A0 + o_hoff),
                       *small = (unsigned long long *)(A0 + o_small);     // [0] union [1] row cursor [2] M [3] H
    SP_HIP(ctx, hipMemcpyAsync(A0 + o_lists, hl.data(), (size_t)C * sizeof(sps_list), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(A0 + o_set, set_off, (size_t)(n_sets + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(A0 + o_uo, unit_off, (size_t)(n_units + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n_uc) SP_HIP(ctx, hipMemcpyAsync(A0 + o_uc, unit_chrom, (size_t)n_uc * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(A0 + o_den, den.data(), (size_t)n_units * 16, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemsetAsync(bnd, 0, (size_t)C * (size_t)(R + 1) * 4, ctx->stream));
    {
        int64_t gx = (longest + 255) / 256;
        if (gx > (int64_t)ctx->n_cu * 16) gx = (int64_t)ctx->n_cu * 16;
        SP_LAUNCH(ctx, "sps_bounds", sps_bounds, dim3((unsigned)(gx > 0 ? gx : 1), (unsigned)C), dim3(256), 0, d_lists, shift, R, bnd);
    }
    sps_join_args A;
    A.C = C;
    A.shift = shift;
    A.R = R;
    A.F.n_sets = n_sets;
    A.F.n_multi = 0;
    for (int st = 0; st < n_sets; st++) A.F.n_multi += (set_off[st + 1] - set_off[st]) > 1;
    A.F.baseline = baseline;
    A.F.set_off = (const int32_t *)(A0 + o_set);
    A.F.unit_off = (const int32_t *)(A0 + o_uo);
    A.F.unit_chrom = (const int32_t *)(A0 + o_uc);
    A.F.unit_den = (const double *)(A0 + o_den);
    A.F.unit_inv = A.F.unit_den + n_units;
    A.F.min_fold = min_fold;
    A.F.min_freq = min_freq;
    A.F.max_freq = max_freq;
    A.F.ratio = ratio;
    A.fast = A.F.n_multi > 0 ? 1 : 0;
    for (int st = 0; st < n_sets; st++) {
        const int nu = set_off[st + 1] - set_off[st];
        if (nu == 1) continue;
        const int bi = baseline < 0 ? nu + baseline : baseline;
        if (!(bi == 1 || bi == nu - 1)) A.fast = 0;
        for (int u = set_off[st]; u < set_off[st + 1]; u++)
            if (unit_off[u + 1] == unit_off[u]) A.fast = 0;
    }
    if (getenv("SP_JOIN_GENERIC") && atoi(getenv("SP_JOIN_GENERIC"))) A.fast = 0;     // cross-check switch
    std::vector<int32_t> h_rd;
    std::vector<float> h_rinv;
    if (A.fast) {
        for (int st = 0; st < n_sets; st++) {
            const int nu = set_off[st + 1] - set_off[st];
            if (nu == 1) continue;
            const int bi = baseline < 0 ? nu + baseline : baseline;
            for (int u = set_off[st]; u < set_off[st + 1]; u++)
                for (int j = unit_off[u]; j < unit_off[u + 1]; j++) {
                    int d = unit_chrom[j];
                    if (j == unit_off[u + 1] - 1) {
                        d |= JD_UNIT_END;
                        if (u == set_off[st + 1] - 1) d |= JD_SET_END | (bi == 1 ? JD_BI1 : 0);
                    }
                    h_rd.push_back(d);
                    h_rinv.push_back(j == unit_off[u + 1] - 1 ? (float)den[(size_t)n_units + (size_t)u] : 0.0f);
                }
        }
        if (h_rd.size() > BJ_FC) A.fast = 0;
    }
    if (!A.fast) {
        h_rd.clear();
        h_rinv.clear();
    }
    A.n_rd = (int)h_rd.size();
    A.rd = (const int32_t *)(A0 + o_rd);
    A.rinv = (const float *)(A0 + o_rinv);
    if (A.n_rd) {      // (the stream is synchronized below, before the vectors go out of scope)
        SP_HIP(ctx, hipMemcpyAsync(A0 + o_rd, h_rd.data(), (size_t)A.n_rd * 4, hipMemcpyHostToDevice, ctx->stream));
        SP_HIP(ctx, hipMemcpyAsync(A0 + o_rinv, h_rinv.data(), (size_t)A.n_rd * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    {   // per chromosome: which non-singleton sets it belongs to (screen of sps_join)
        std::vector<unsigned long long> cs((size_t)C, 0ULL);
        int ms = 0;
        for (int st = 0; st < n_sets; st++) {
            if (set_off[st + 1] - set_off[st] <= 1) continue;
            if (ms < 32)
                for (int u = set_off[st]; u < set_off[st + 1]; u++)
                    for (int j = unit_off[u]; j < unit_off[u + 1]; j++) cs[(size_t)unit_chrom[j]] |= 1ULL << ms;
            ms++;
        }
        A.screen = ms <= 32 ? 1 : 0;      // (sps_join_blk keeps 32-bit masks)
        SP_HIP(ctx, hipMemcpyAsync(A0 + o_cs, cs.data(), (size_t)C * 8, hipMemcpyHostToDevice, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));     // cs goes out of scope
        A.chrom_sets = (const unsigned long long *)(A0 + o_cs);
    }
    A.lists = d_lists;
    A.bnd = bnd;
    A.n_rows = n_rows;
    A.n_hist = n_hist;
    A.hist_stage = (unsigned long long *)ctx->b_sp_b.p;
    A.row_cursor = small + 1;
    A.n_union = small;
    // differential rows are rare (a fraction of a percent of the union on the BASELINE genomes): the staging area holds
    // total / 16 rows (at least 2^20); if a filter configuration keeps more, the pass is repeated with what it asked for
    unsigned long long row_cap = (unsigned long long)(total / 16);
    if (row_cap < (1ULL << 20)) row_cap = (unsigned long long)(total < (1LL << 20) ? total : (1LL << 20));
    row_cap += (unsigned long long)ctx->n_cu * 16 * JOIN_CHUNK;       // every resident workgroup may strand one chunk
    unsigned long long h[4] = {0, 0, 0, 0};
    for (int attempt = 0;; attempt++) {
        const size_t row_bytes = 8 + 8 + 4 + (size_t)C * 4;
        rc = sp_buf_ensure(ctx, ctx->b_sp_c, (int64_t)(al(row_cap * 8) * 2 + al(row_cap * 4) + al(row_cap * (size_t)C * 4) + 64));
        if (rc) return rc;
        (void)row_bytes;
        char *S0 = (char *)ctx->b_sp_c.p;
        A.row_cap = row_cap;
        A.row_keys = (unsigned long long *)S0;
        A.row_tot = (unsigned long long *)(S0 + al(row_cap * 8));
        A.row_rank = (uint32_t *)(S0 + 2 * al(row_cap * 8));
        A.row_counts = (uint32_t *)(S0 + 2 * al(row_cap * 8) + al(row_cap * 4));
        SP_HIP(ctx, hipMemsetAsync(A.row_keys, 0xff, row_cap * 8, ctx->stream));
        SP_HIP(ctx, hipMemsetAsync(small, 0, 64, ctx->stream));
        {
            int64_t grid = R;
            if (grid > (int64_t)ctx->n_cu * 16) grid = (int64_t)ctx->n_cu * 16;
            const size_t row_lds = (size_t)BJ_NR * (size_t)(C | 1) * 4;     // <= 16.3 KiB
            // (static + dynamic LDS passes 64 KiB with many chromosomes and 64-bit residuals)
            if (shift <= 31)
                SP_HIP(ctx, hipFuncSetAttribute((const void *)sps_join_blk<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds));
            else
                SP_HIP(ctx, hipFuncSetAttribute((const void *)sps_join_blk<unsigned long long>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds));
            if (shift <= 31)
                SP_LAUNCH(ctx, "sps_join", sps_join_blk<uint32_t>, dim3((unsigned)grid), dim3(BJ_THREADS), row_lds, A);
            else
                SP_LAUNCH(ctx, "sps_join", sps_join_blk<unsigned long long>, dim3((unsigned)grid), dim3(BJ_THREADS), row_lds, A);
        }
        const long long nb = (R + TALLY_CHUNK - 1) / TALLY_CHUNK;
        unsigned long long *bs_r = (unsigned long long *)(A0 + o_bs), *bs_h = bs_r + (R / TALLY_CHUNK + 2);
        SP_LAUNCH(ctx, "sps_tally_sums", sps_tally_sums, dim3((unsigned)nb), dim3(256), 0, (const uint32_t *)n_rows, R, bs_r);
        SP_LAUNCH(ctx, "sps_tally_sums", sps_tally_sums, dim3((unsigned)nb), dim3(256), 0, (const uint32_t *)n_hist, R, bs_h);
        SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, bs_r, (int64_t)nb, small + 2);
        SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, bs_h, (int64_t)nb, small + 3);
        SP_LAUNCH(ctx, "sps_tally_apply", sps_tally_apply, dim3((unsigned)nb), dim3(256), 0, (const uint32_t *)n_rows, R,
                  (const unsigned long long *)bs_r, (const unsigned long long *)(small + 2), row_off);
        SP_LAUNCH(ctx, "sps_tally_apply", sps_tally_apply, dim3((unsigned)nb), dim3(256), 0, (const uint32_t *)n_hist, R,
                  (const unsigned long long *)bs_h, (const unsigned long long *)(small + 3), hist_off);
        SP_HIP(ctx, hipMemcpyAsync(h, small, 32, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h[1] <= row_cap) break;
        if (attempt) return sp_fail(ctx, SP_ESTATE, "list filter: row staging overran twice (%llu > %llu)", h[1], row_cap);
        row_cap = h[1] + (unsigned long long)ctx->n_cu * 16 * JOIN_CHUNK;
    }
    ctx->n_union = (int64_t)h[0];
    ctx->n_rows = (int64_t)h[2];
    ctx->n_hist = (int64_t)h[3];
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
        SP_LAUNCH(ctx, "sps_place_rows", sps_place_rows, dim3((unsigned)((h[1] + 255) / 256)), dim3(256), 0,
                  (const unsigned long long *)A.row_keys, (const unsigned long long *)A.row_tot, (const uint32_t *)A.row_rank,
                  (const uint32_t *)A.row_counts, h[1], C, shift, (const unsigned long long *)row_off,
                  (unsigned long long *)ctx->b_sf_keys.p, (uint32_t *)ctx->b_sf_counts.p, (unsigned long long *)ctx->b_sf_tot.p);
    if (H)
        SP_LAUNCH(ctx, "sps_place_hist", sps_place_hist, dim3((unsigned)((R + 255) / 256)), dim3(256), 0,
                  (const unsigned long long *)A.hist_stage, (const uint32_t *)bnd, C, R, (const uint32_t *)n_hist,
                  (const unsigned long long *)hist_off, (unsigned long long *)ctx->b_sf_hist.p);
    if (M && ctx->list_mode)
        SP_LAUNCH(ctx, "sps_slots_to_keys", sps_slots_to_keys, dim3((unsigned)((M + 255) / 256)), dim3(256), 0,
                  (unsigned long long *)ctx->b_sf_keys.p, M, sp_make_kparams(ctx->k));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->filtered = true;
    return SP_OK;
}

// SP_LIST_FILTER=sort selects the first implementation (concatenate + library radix sort), kept as a cross-check
int sp_sparse_filter(sp_ctx *ctx, int n_sets, const int32_t *set_off, const int32_t *unit_off,
                     const int32_t *unit_chrom, const std::vector<double> &den, double min_fold, int baseline,
                     double min_freq, double max_freq, double ratio) {
    const char *e = getenv("SP_LIST_FILTER");
    if (e && !strcmp(e, "sort"))
        return sps_filter_sort(ctx, n_sets, set_off, unit_off, unit_chrom, den, min_fold, baseline, min_freq, max_freq, ratio);
    return sps_filter_join(ctx, n_sets, set_off, unit_off, unit_chrom, den, min_fold, baseline, min_freq, max_freq, ratio);
}

int sp_sparse_fetch(sp_ctx *ctx, bool hist, uint64_t *keys, uint32_t *counts, double *freqs, uint64_t *tot, bool async) {
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
    // async (sp_filter_fetch_async: no frequencies): the join left the rows in device buffers that nothing writes before the
    // next filter call -- a copy stream takes them to the (page-locked) host buffers while the compute stream goes on with the
    // map stage; sp_filter_fetch_wait joins the two (as the table engines do since round 4)
    hipStream_t cs = ctx->stream;
    async = async && !freqs;
    if (async) {
        if (!ctx->copy_stream) {
            SP_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            SP_HIP(ctx, hipEventCreateWithFlags(&ctx->copy_event, hipEventDisableTiming));
        }
        SP_HIP(ctx, hipEventRecord(ctx->copy_event, ctx->stream));
        SP_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->copy_event, 0));
        cs = ctx->copy_stream;
    }
    if (keys) SP_HIP(ctx, hipMemcpyAsync(keys, ctx->b_sf_keys.p, (size_t)M * 8, hipMemcpyDeviceToHost, cs));
    if (tot) SP_HIP(ctx, hipMemcpyAsync(tot, ctx->b_sf_tot.p, (size_t)M * 8, hipMemcpyDeviceToHost, cs));
    if (cdst) SP_HIP(ctx, hipMemcpyAsync(cdst, ctx->b_sf_counts.p, (size_t)M * C * 4, hipMemcpyDeviceToHost, cs));
    if (async) return SP_OK;
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (freqs)
        for (int64_t r = 0; r < M; r++)
            for (int c = 0; c < C; c++)   // count/length in fp64 (Jellyfish.py:647); IEEE division, same bits as the device path
                freqs[r * C + c] = (double)cdst[r * C + c] / (double)sps_len(ctx, c);
    return SP_OK;
}

int sp_map_filter_build(sp_ctx *ctx, const unsigned long long *d_keys, int64_t n);   // sp_map.hip

// the quad-bucket table the current k > 15 label set lives in (buckets = NULL: the pair-keyed / per-k-mer hash table)
static sq_tab sq_tab_of(const sp_ctx *ctx) {
    sq_tab T;
    T.buckets = ctx->sq_bb ? (unsigned long long *)ctx->b_ctab.p : nullptr;
    T.ovf = (unsigned long long *)ctx->b_covf.p;
    T.ovf_mask = ctx->sq_ovf_mask;
    T.sb = 2 * (ctx->k - 3);
    T.tb = ctx->sq_bb ? T.sb - ctx->sq_bb : 0;
    return T;
}

int sp_sparse_labels_set(sp_ctx *ctx, const uint64_t *keys, const uint8_t *sg, int64_t n, bool on_device) {
    // <= 7 subgenomes: the pair-keyed table (one look-up per candidate PAIR of starts); else one entry per k-mer
    const char *eng = getenv("SP_MAP_ENGINE");
    ctx->map_engine = (ctx->n_sg > 7 || (eng && eng[0] == '1')) ? 1 : 0;
    const bool pairs = ctx->map_engine == 0;
    // round 6: <= 3 subgenomes -> the quad-bucket table (one 32-byte look-up per candidate QUAD of starts) when its tag fits;
    // SP_CTAB=0 keeps the pair-keyed hash table (cross-check), SP_CTAB_FACTOR sizes the buckets (default: >= 2 n of them)
    ctx->sq_bb = 0;
    int bb = 8;
    bool quad = false;
    if (pairs && ctx->n_sg <= 3) {
        const char *env_ct = getenv("SP_CTAB"), *env_f = getenv("SP_CTAB_FACTOR");
        const int64_t factor = env_f && atoll(env_f) > 0 ? atoll(env_f) : 2;
        const int sb = 2 * (ctx->k - 3);
        while (bb < 30 && ((int64_t)1 << bb) < factor * (n > 0 ? n : 1)) bb++;
        if (bb < sb - (SQ_TAG_BITS - 5)) bb = sb - (SQ_TAG_BITS - 5);      // the tag holds sb - bb bits of the mixed core + side + e
        quad = bb <= 27 && bb <= sb && !(env_ct && env_ct[0] == '0');
    }
    int64_t cap = 1024;
    if (!quad)
        while (cap < (pairs ? 4 : 2) * n + 16) cap <<= 1;
    if (cap != ctx->hcap) {
        if (ctx->d_hkeys) hipFree(ctx->d_hkeys);
        ctx->d_hkeys = nullptr;
        SP_HIP(ctx, hipMalloc(&ctx->d_hkeys, (size_t)cap * 16));
        ctx->hcap = cap;
    }
    if (quad) {
        // (d_hkeys stays allocated -- it is what "labels are set" is tested by -- but unused)
        const int64_t nb = (int64_t)1 << bb;
        int64_t ovf_n = 4096;
        while (ovf_n < n / 2) ovf_n <<= 1;
        int rcq = sp_buf_ensure(ctx, ctx->b_ctab, nb * 32);
        if (rcq) return rcq;
        rcq = sp_buf_ensure(ctx, ctx->b_covf, ovf_n * 16);
        if (rcq) return rcq;
        SP_HIP(ctx, hipMemsetAsync(ctx->b_ctab.p, 0, (size_t)nb * 32, ctx->stream));
        SP_HIP(ctx, hipMemsetAsync(ctx->b_covf.p, 0, (size_t)ovf_n * 16, ctx->stream));
        ctx->sq_bb = bb;
        ctx->sq_ovf_mask = (uint64_t)(ovf_n - 1);
    } else if (pairs) {
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
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    SP_HIP(ctx, hipMemcpyAsync(d_keys, keys, (size_t)n * 8, kind, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_sg, sg, (size_t)n, kind, ctx->stream));
    auto build_pairs = [&]() -> int {
        SP_LAUNCH(ctx, "sps_pair_insert", sps_pair_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                  (const unsigned long long *)d_keys, (const uint8_t *)d_sg, n, ctx->k, (unsigned long long *)ctx->d_hkeys,
                  (uint64_t)(ctx->hcap - 1));
        return SP_OK;
    };
    if (quad) {
        // [2] = overflow table full, [3] = keys in the overflow table (the flags of this call: sp_labels_set zeroed them)
        unsigned long long *d_flags = (unsigned long long *)ctx->b_lflags.p;
        SP_LAUNCH(ctx, "sq_build", sq_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const unsigned long long *)d_keys,
                  (const uint8_t *)d_sg, n, ctx->k, sq_tab_of(ctx), d_flags + 2);
        unsigned long long hf[2] = {0, 0};
        SP_HIP(ctx, hipMemcpyAsync(hf, d_flags + 2, 16, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (getenv("SP_DEBUG_FILTER"))
            fprintf(stderr, "[sp] quad-bucket table (k > 15): 2^%d buckets, %llu entries in the overflow table (%lld labelled k-mers)\n",
                    bb, hf[1], (long long)n);
        if (hf[0]) {          // overflow table full (adversarial key sets only): the pair-keyed hash table takes over
            ctx->sq_bb = 0;
            cap = 1024;
            while (cap < 4 * n + 16) cap <<= 1;
            if (cap != ctx->hcap) {
                if (ctx->d_hkeys) hipFree(ctx->d_hkeys);
                ctx->d_hkeys = nullptr;
                SP_HIP(ctx, hipMalloc(&ctx->d_hkeys, (size_t)cap * 16));
                ctx->hcap = cap;
            }
            SP_LAUNCH(ctx, "sps_pair_init", sps_pair_init, dim3((unsigned)(ctx->n_cu * 8)), dim3(256), 0,
                      (unsigned long long *)ctx->d_hkeys, cap);
            const int rcp = build_pairs();
            if (rcp) return rcp;
        }
    } else if (pairs) {
        const int rcp = build_pairs();
        if (rcp) return rcp;
    } else
        SP_LAUNCH(ctx, "sps_hash_insert", sps_hash_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                  (const unsigned long long *)d_keys, (const uint8_t *)d_sg, n, (unsigned long long *)ctx->d_hkeys,
                  (uint64_t)(ctx->hcap - 1));
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
    if (ctx->map_engine == 0 && P.S <= 7) {
        const sq_tab T = sq_tab_of(ctx);
        if (T.buckets)
            SP_LAUNCH(ctx, "k5_map_sparse", k5_map_sparse2<1>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, c.d_pk, c.d_pm, c.d_nm, kp, P,
                      (unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1), T, ctx->d_bloom, ctx->bloom_bits, d_counts, d_n);
        else
            SP_LAUNCH(ctx, "k5_map_sparse", k5_map_sparse2<0>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, c.d_pk, c.d_pm, c.d_nm, kp, P,
                      (unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1), T, ctx->d_bloom, ctx->bloom_bits, d_counts, d_n);
        return SP_OK;
    }
    if (ctx->map_engine == 0) return sp_fail(ctx, SP_ESTATE, "k > 15 map: the pair-keyed table holds at most 7 subgenomes");
    SP_LAUNCH(ctx, "k5_map_sparse_lab", k5_map_sparse_lab, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, c.d_pk, c.d_nm, kp, P,
              (unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1), ctx->d_bloom, ctx->bloom_bits, d_counts, d_n);
    return SP_OK;
}

int sp_sparse_feat_launch(sp_ctx *ctx, const uint32_t *d_pk, const uint32_t *d_pm, const uint32_t *d_nm, int64_t n_units,
                          const int64_t *d_foff, int64_t n_feat, int S, unsigned long long *d_counts) {
    const sp_kparams kp = sp_make_kparams(ctx->k);
    int64_t grid = (n_units + MAP_BLOCK - 1) / MAP_BLOCK;
    if (grid > (int64_t)ctx->n_cu * 16) grid = (int64_t)ctx->n_cu * 16;
    if (ctx->map_engine == 0 && S <= 7) {
        const sq_tab T = sq_tab_of(ctx);
        if (T.buckets)
            SP_LAUNCH(ctx, "k5_map_feat_sparse", k5_map_feat_sparse2<1>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, d_pk, d_pm, d_nm,
                      kp, n_units, d_foff, n_feat, S, (unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1), T, ctx->d_bloom,
                      ctx->bloom_bits, d_counts);
        else
            SP_LAUNCH(ctx, "k5_map_feat_sparse", k5_map_feat_sparse2<0>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, d_pk, d_pm, d_nm,
                      kp, n_units, d_foff, n_feat, S, (unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1), T, ctx->d_bloom,
                      ctx->bloom_bits, d_counts);
    } else if (ctx->map_engine == 0)
        return sp_fail(ctx, SP_ESTATE, "k > 15 map: the pair-keyed table holds at most 7 subgenomes");
    else
        SP_LAUNCH(ctx, "k5_map_feat_sparse_lab", k5_map_feat_sparse_lab, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, d_pk, d_nm, kp,
                  n_units, d_foff, n_feat, S, (unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1), ctx->d_bloom,
                  ctx->bloom_bits, d_counts);
    return SP_OK;
}

// interval mode (sp_map.hip: kv_cover / kv_count): the k > 15 scan that fills the per-unit subgenome masks
__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_mask_sparse_lab(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, sp_kparams kp, int64_t n_units, int S,
                       unsigned long long *__restrict__ htab, uint64_t mask, const uint32_t *__restrict__ bloom, int bloom_bits,
                       const unsigned long long *__restrict__ cov, unsigned long long *__restrict__ masks) {
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; u < n_units; u += stride) {
        const unsigned long long cv = cov[u];
        if (__all(cv == 0ULL)) continue;
        map_pair_scan<uint64_t>(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, [&](int64_t start, uint64_t fwd, uint64_t rc) {
            const unsigned long long bit = 1ULL << (start & 63);
            if (!(cv & bit)) return;
            const int sg = sps_lookup(fwd < rc ? fwd : rc, htab, mask);
            if (sg >= 0) atomicOr(&masks[u * S + sg], bit);
        });
    }
}

template <int TABLE>
__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_mask_sparse2(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm, sp_kparams kp,
                    int64_t n_units, int S, unsigned long long *__restrict__ htab, uint64_t hmask, sq_tab T, const uint32_t *__restrict__ bloom,
                    int bloom_bits, const unsigned long long *__restrict__ cov, unsigned long long *__restrict__ masks) {
    MAP_UNIT_LDS_DECL(TABLE, 8);
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; u < n_units; u += stride) {
        const unsigned long long cv = cov[u];
        if (__all(cv == 0ULL)) continue;
        unsigned long long lab[3] = {0ULL, 0ULL, 0ULL};
        map_unit_scan64_h<TABLE>(pk, pm, nm, u * SP_UNIT, kp, bloom, bloom_bits, htab, hmask, T, lab, ulds, cv);
        for (int sg = 0; sg < S; sg++) {
            const int l = sg + 1;
            masks[u * S + sg] = ((l & 1) ? lab[0] : ~lab[0]) & ((l & 2) ? lab[1] : ~lab[1]) & ((l & 4) ? lab[2] : ~lab[2]) & cv;
        }
    }
}

int sp_sparse_mask_launch(sp_ctx *ctx, sp_chrom &c, int64_t n_units, int S, const unsigned long long *d_cov,
                          unsigned long long *d_masks) {
    const sp_kparams kp = sp_make_kparams(ctx->k);
    int64_t grid = (n_units + MAP_BLOCK - 1) / MAP_BLOCK;
    if (grid > (int64_t)ctx->n_cu * 16) grid = (int64_t)ctx->n_cu * 16;
    if (ctx->map_engine == 0 && S <= 7) {
        const sq_tab T = sq_tab_of(ctx);
        if (T.buckets)
            SP_LAUNCH(ctx, "k5_map_mask_sparse", k5_map_mask_sparse2<1>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, (const uint32_t *)c.d_pk,
                      (const uint32_t *)c.d_pm, (const uint32_t *)c.d_nm, kp, n_units, S, (unsigned long long *)ctx->d_hkeys,
                      (uint64_t)(ctx->hcap - 1), T, (const uint32_t *)ctx->d_bloom, ctx->bloom_bits, d_cov, d_masks);
        else
            SP_LAUNCH(ctx, "k5_map_mask_sparse", k5_map_mask_sparse2<0>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, (const uint32_t *)c.d_pk,
                      (const uint32_t *)c.d_pm, (const uint32_t *)c.d_nm, kp, n_units, S, (unsigned long long *)ctx->d_hkeys,
                      (uint64_t)(ctx->hcap - 1), T, (const uint32_t *)ctx->d_bloom, ctx->bloom_bits, d_cov, d_masks);
        return SP_OK;
    }
    if (ctx->map_engine == 0) return sp_fail(ctx, SP_ESTATE, "k > 15 map: the pair-keyed table holds at most 7 subgenomes");
    SP_LAUNCH(ctx, "k5_map_mask_sparse_lab", k5_map_mask_sparse_lab, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, (const uint32_t *)c.d_pk,
              (const uint32_t *)c.d_nm, kp, n_units, S, (unsigned long long *)ctx->d_hkeys, (uint64_t)(ctx->hcap - 1),
              (const uint32_t *)ctx->d_bloom, ctx->bloom_bits, d_cov, d_masks);
    return SP_OK;
}

int sp_sparse_hit(sp_ctx *ctx, unsigned long long *d_n) {
    if (ctx->map_engine == 0 && ctx->sq_bb) {
        if (ctx->n_labels > 0)
            SP_LAUNCH(ctx, "sq_seen", sq_seen, dim3((unsigned)(ctx->n_cu * 4)), dim3(256), 0,
                      (const unsigned long long *)ctx->b_labkeys.p, ctx->n_labels, ctx->k, sq_tab_of(ctx), d_n);
        return SP_OK;
    }
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
