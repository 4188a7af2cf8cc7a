// sp_count.hip -- K1 (canonical k-mer counting into a dense per-chromosome
// table), K2 (lower_count threshold + lengths), dump compaction.
//
// Replaces the external `jellyfish count --canonical | dump -c -L` step
// (reference: subphaser/Jellyfish.py:671-704).
#include "sp_device.h"
#include "sp_c2batch.h"

// ----------------------------------------------------------------- K1 / engine 1
// Baseline engine: one global atomic per k-mer occurrence into the dense
// table.  Random 4-byte read-modify-writes over a 2-GiB table: bound by the
// memory system's random-access rate, not by streaming bandwidth.  Kept as the
// always-correct fallback and as the parity cross-check for engine 2.
__global__ void __launch_bounds__(256)
k1_count_atomic(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, int64_t n_units,
                sp_kparams32 kp, uint32_t *__restrict__ tab) {
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; u < n_units; u += stride) {
        uint32_t last = 0xFFFFFFFFu;
        uint32_t acc = 0;
        sp_scan_unit32<SP_UNIT>(pk, nm, u * SP_UNIT, kp, [&](int64_t, uint32_t fwd, uint32_t rc) {
            const uint32_t slot = sp_slot_of32(fwd, rc, kp);
            if (slot == last) {
                acc++;  // homopolymer runs collapse into one atomic
            } else {
                if (acc) atomicAdd(&tab[last], acc);
                last = slot;
                acc = 1;
            }
        });
        if (acc) atomicAdd(&tab[last], acc);
    }
}

// ----------------------------------------------------------------- K2 (engine 1) + byte table
// Engine 1 counts into a u32 scratch table; this pass turns it into the byte table every consumer
// reads (raw count saturated at 255 + overflow pairs), fused with lengths[c] = sum of counts >= lower
// (Jellyfish.py:97,449) and the dump size.  One block per bucket of 2^15 slots, same overflow-segment
// protocol as c2_count (sp_count2.hip).
__global__ void __launch_bounds__(256)
k1_narrow(const uint32_t *__restrict__ tab32, int64_t nslots, uint32_t lower, uint8_t *__restrict__ tab,
          unsigned long long *__restrict__ out3 /*[0]=sum,[1]=n,[2]=overflow cursor*/, uint2 *__restrict__ ovf_tmp,
          unsigned long long ovf_cap, uint32_t *__restrict__ seg_base, uint32_t *__restrict__ seg_cnt,
          int64_t n_buckets) {
    __shared__ unsigned long long red[16];
    __shared__ uint32_t s_nov, s_rank;
    __shared__ unsigned long long s_base;
    unsigned long long s = 0, n = 0;
    for (int64_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        if (threadIdx.x == 0) s_nov = s_rank = 0;
        __syncthreads();
        const int64_t lo = b << SP_OVF_SHIFT;
        int64_t hi = lo + (1LL << SP_OVF_SHIFT);
        if (hi > nslots) hi = nslots;
        uint32_t my_ov = 0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const uint32_t v = tab32[i];
            if (v >= lower) { s += v; n++; }
            my_ov += v >= 255u;
            tab[i] = (uint8_t)(v < 255u ? v : 255u);
        }
        if (my_ov) atomicAdd(&s_nov, my_ov);
        __syncthreads();
        const uint32_t nov = s_nov;
        if (nov) {
            if (threadIdx.x == 0) s_base = atomicAdd(&out3[2], (unsigned long long)nov);
            __syncthreads();
            const unsigned long long base = s_base;
            for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
                const uint32_t v = tab32[i];
                if (v >= 255u) {
                    const unsigned long long pos = base + atomicAdd(&s_rank, 1u);
                    if (pos < ovf_cap) ovf_tmp[pos] = make_uint2((uint32_t)i, v);
                }
            }
            if (threadIdx.x == 0) seg_base[b] = (uint32_t)base;
        }
        if (threadIdx.x == 0) seg_cnt[b] = nov;
        __syncthreads();
    }
    unsigned long long ts = sp_block_sum_u64(s, red);
    unsigned long long tn = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0) {
        if (ts) atomicAdd(&out3[0], ts);
        if (tn) atomicAdd(&out3[1], tn);
    }
}

// ----------------------------------------------------------------- overflow list: segments -> sorted list
// exclusive scan of the per-bucket overflow counts (n <= 2^16 buckets; single block)
__device__ __forceinline__ void ovf_scan_body(const uint32_t *__restrict__ seg_cnt, int64_t n, uint32_t *__restrict__ seg_off /*n+1*/,
         unsigned long long *__restrict__ total_out /* number of pairs, or NULL */) {
    __shared__ uint32_t wsum[16];
    const int64_t per = (n + 1023) / 1024;
    int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    uint32_t s = 0;
    for (int64_t i = lo; i < hi; i++) s += seg_cnt[i];
    uint32_t total;
    uint32_t run = sp_block_excl_scan(s, wsum, total);
    if (threadIdx.x == 0) {
        seg_off[n] = total;
        if (total_out) *total_out = total;
    }
    for (int64_t i = lo; i < hi; i++) {
        seg_off[i] = run;
        run += seg_cnt[i];
    }
}
// one wave per bucket: its pairs are ranked by slot and written to their final place.  A few pairs (the usual
// overflow case: ~9 per bucket) are ranked by brute force; a bucket with many (engine 3 stages EVERY kept slot) marks
// its slots in a 2^15-bit LDS bitmap and ranks by prefix popcount -- O(m + 1024) instead of O(m^2).
// SPLIT: the pairs go to separate (u64 slot, u32 count) arrays, the form the list engines work on.
#define OVF_BRUTE 96
template <bool SPLIT>
__device__ __forceinline__ void ovf_place_body(const uint2 *__restrict__ tmp, const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_cnt,
          const uint32_t *__restrict__ seg_off, int64_t n_buckets, uint2 *__restrict__ out,
          unsigned long long *__restrict__ out_keys, uint32_t *__restrict__ out_cnts) {
    __shared__ uint32_t bits[4][1 << (SP_OVF_SHIFT - 5)], pre[4][1 << (SP_OVF_SHIFT - 5)];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + w;
    if (b >= n_buckets) return;
    const uint32_t m = seg_cnt[b];
    if (!m) return;
    const uint2 *src = tmp + seg_base[b];
    const size_t dst0 = seg_off[b];
    auto put = [&](uint32_t rank, uint2 e) {
        if (SPLIT) {
            out_keys[dst0 + rank] = e.x;
            out_cnts[dst0 + rank] = e.y;
        } else {
            out[dst0 + rank] = e;
        }
    };
    if (m <= OVF_BRUTE) {
        for (uint32_t j = lane; j < m; j += 64) {
            const uint2 e = src[j];
            uint32_t rank = 0;
            for (uint32_t i = 0; i < m; i++) rank += src[i].x < e.x;
            put(rank, e);
        }
        return;
    }
    constexpr int NW = 1 << (SP_OVF_SHIFT - 5);     // 1024 words
    for (int i = lane; i < NW; i += 64) bits[w][i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = lane; j < m; j += 64) {
        const uint32_t r = src[j].x & ((1u << SP_OVF_SHIFT) - 1u);
        atomicOr(&bits[w][r >> 5], 1u << (r & 31u));
    }
    __builtin_amdgcn_wave_barrier();
    // exclusive prefix popcount over the words: lane l owns words [16 l, 16 l + 16)
    uint32_t mine = 0;
    for (int i = 0; i < NW / 64; i++) mine += __popc(bits[w][lane * (NW / 64) + i]);
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t nb = __shfl_up(incl, o, 64);
        if (lane >= o) incl += nb;
    }
    uint32_t run = incl - mine;
    for (int i = 0; i < NW / 64; i++) {
        pre[w][lane * (NW / 64) + i] = run;
        run += __popc(bits[w][lane * (NW / 64) + i]);
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = lane; j < m; j += 64) {
        const uint2 e = src[j];
        const uint32_t r = e.x & ((1u << SP_OVF_SHIFT) - 1u);
        put(pre[w][r >> 5] + __popc(bits[w][r >> 5] & ((1u << (r & 31u)) - 1u)), e);
    }
}

// ----------------------------------------------------------------- dump
#define DUMP_PER_THREAD 16
#define DUMP_BLOCK 256
#define DUMP_SLOTS (DUMP_PER_THREAD * DUMP_BLOCK)

__global__ void __launch_bounds__(DUMP_BLOCK)
dump_count(sp_tabref T, int64_t nslots, uint32_t lower, unsigned long long *__restrict__ blk) {
    __shared__ unsigned long long red[16];
    int64_t base = (int64_t)blockIdx.x * DUMP_SLOTS;
    unsigned long long n = 0;
    for (int j = 0; j < DUMP_PER_THREAD; j++) {
        int64_t i = base + (int64_t)j * DUMP_BLOCK + threadIdx.x;
        if (i < nslots && sp_tab_count(T, i, 0) >= lower) n++;
    }
    unsigned long long t = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0) blk[blockIdx.x] = t;
}

// single-block exclusive scan of n u64 values (n up to a few million)
__global__ void __launch_bounds__(1024)
scan_excl_u64(unsigned long long *__restrict__ a, int64_t n, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long wsum[16];
    const int T = 1024;
    int64_t per = (n + T - 1) / T;
    int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    unsigned long long s = 0;
    for (int64_t i = lo; i < hi; i++) s += a[i];
    unsigned long long tot;
    unsigned long long run = sp_block_excl_scan(s, wsum, tot);
    if (threadIdx.x == 0 && total) *total = tot;
    for (int64_t i = lo; i < hi; i++) {
        unsigned long long v = a[i];
        a[i] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(DUMP_BLOCK)
dump_write(sp_tabref T, int64_t nslots, uint32_t lower,
           const unsigned long long *__restrict__ blk, sp_kparams kp,
           unsigned long long *__restrict__ keys, uint32_t *__restrict__ counts) {
    __shared__ uint32_t lds[16];
    int64_t base = (int64_t)blockIdx.x * DUMP_SLOTS;
    unsigned long long off = blk[blockIdx.x];
    for (int j = 0; j < DUMP_PER_THREAD; j++) {
        int64_t i = base + (int64_t)j * DUMP_BLOCK + threadIdx.x;
        uint32_t c = (i < nslots) ? sp_tab_count(T, i, 0) : 0u;
        bool p = (i < nslots) && c >= lower;
        uint32_t tot;
        uint32_t my = sp_block_excl_count(p, lds, tot);
        if (p) {
            keys[off + my] = sp_key_of_slot((uint64_t)i, kp);
            counts[off + my] = c;
        }
        off += tot;
    }
}

static int grid_for(sp_ctx *ctx, int64_t work_items, int per_block, int max_per_cu) {
    int64_t b = (work_items + per_block - 1) / per_block;
    int64_t cap = (int64_t)ctx->n_cu * max_per_cu;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

int sp_count_engine2(sp_ctx *ctx, sp_chrom &c, const sp_kparams &kp, int lower,
                     unsigned long long *d_len4, bool exact, sp_sparse_chrom *list);  // sp_count2.hip
int sp_count_engine3_batch(sp_ctx *ctx, const int *chrom_idx, int n, const sp_kparams &kp, int lower, unsigned long long *d_len);   // sp_count2.hip
void sp_sparse_release(sp_ctx *ctx);                                                  // sp_sparse.hip
bool sp_engine2_supported(int64_t nslots);
int sp_sparse_count(sp_ctx *ctx, int k, int lower);                                   // sp_sparse.hip
int sp_sparse_count3(sp_ctx *ctx, int k, int lower);                                  // sp_sparse2.hip
int sp_sparse_dump(sp_ctx *ctx, int chrom, uint64_t *keys, uint32_t *counts);

// ----------------------------------------------------------------- merging two byte tables (multi-GPU)
// A chromosome whose bases were counted by two ranks arrives at the rank that filters a slot range as two
// byte slices.  dst += src over n slots, exactly: a slot whose sum reaches 255 -- or whose summands were
// already saturated -- is written as 255 and its exact sum (overflow-list look-ups for saturated summands)
// goes to the merged overflow list, through the same per-bucket segment protocol as the counting kernels.
__global__ void __launch_bounds__(256)
kx_merge(uint8_t *dst /* may alias A.tab */, sp_tabref A, sp_tabref B, int64_t slot_base, int64_t n,
         unsigned long long *__restrict__ cursor, uint2 *__restrict__ ovf_tmp, unsigned long long ovf_cap,
         uint32_t *__restrict__ seg_base, uint32_t *__restrict__ seg_cnt, int64_t n_buckets) {
    __shared__ uint32_t s_nov, s_rank;
    __shared__ unsigned long long s_base;
    for (int64_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        if (threadIdx.x == 0) s_nov = s_rank = 0;
        __syncthreads();
        const int64_t lo = b << SP_OVF_SHIFT;
        int64_t hi = lo + (1LL << SP_OVF_SHIFT);
        if (hi > n) hi = n;
        uint32_t my_ov = 0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const uint32_t x = A.tab[i], y = B.tab[i];
            my_ov += (x == 255u || y == 255u || x + y >= 255u);
        }
        if (my_ov) atomicAdd(&s_nov, my_ov);
        __syncthreads();
        const uint32_t nov = s_nov;
        if (nov && threadIdx.x == 0) s_base = atomicAdd(cursor, (unsigned long long)nov);
        __syncthreads();
        const unsigned long long base = s_base;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const uint32_t x = A.tab[i], y = B.tab[i];
            if (x == 255u || y == 255u || x + y >= 255u) {
                const uint32_t slot = (uint32_t)(slot_base + i);
                const uint32_t ex = (x < 255u ? x : sp_ovf_lookup(A.ovf, A.n_ovf, slot)) +
                                    (y < 255u ? y : sp_ovf_lookup(B.ovf, B.n_ovf, slot));
                const unsigned long long pos = base + atomicAdd(&s_rank, 1u);
                if (pos < ovf_cap) ovf_tmp[pos] = make_uint2(slot, ex);
                dst[i] = 255;
            } else {
                dst[i] = (uint8_t)(x + y);
            }
        }
        if (threadIdx.x == 0) {
            seg_cnt[b] = nov;
            if (nov) seg_base[b] = (uint32_t)base;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(1024)
ovf_scan(const uint32_t *__restrict__ seg_cnt, int64_t n, uint32_t *__restrict__ seg_off, unsigned long long *__restrict__ total_out) {
    ovf_scan_body(seg_cnt, n, seg_off, total_out);
}
template <bool SPLIT>
__global__ void __launch_bounds__(256)
ovf_place(const uint2 *__restrict__ tmp, const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_cnt,
          const uint32_t *__restrict__ seg_off, int64_t n_buckets, uint2 *__restrict__ out,
          unsigned long long *__restrict__ out_keys, uint32_t *__restrict__ out_cnts) {
    ovf_place_body<SPLIT>(tmp, seg_base, seg_cnt, seg_off, n_buckets, out, out_keys, out_cnts);
}
// one chromosome per blockIdx.y (sp_c2batch.h)
__global__ void __launch_bounds__(1024)
ovf_scan_b(const c2_bdesc *__restrict__ desc, int64_t n) {
    const c2_bdesc D = desc[blockIdx.y];
    ovf_scan_body(D.seg_cnt, n, D.seg_off, D.d_len4 + 2);
}
__global__ void __launch_bounds__(256)
ovf_place_list_b(const c2_bdesc *__restrict__ desc, int64_t n_buckets) {
    const c2_bdesc D = desc[blockIdx.y];
    ovf_place_body<true>(D.stage, D.seg_base, D.seg_cnt, D.seg_off, n_buckets, (uint2 *)nullptr, D.out_keys, D.out_cnts);
}
int sp_ovf_finalize_split_batch(sp_ctx *ctx, const c2_bdesc *d_desc, int n_chrom, int64_t n_buckets) {
    SP_LAUNCH(ctx, "ovf_scan", ovf_scan_b, dim3(1, (unsigned)n_chrom), dim3(1024), 0, d_desc, n_buckets);
    SP_LAUNCH(ctx, "ovf_place_list", ovf_place_list_b, dim3((unsigned)((n_buckets + 3) / 4), (unsigned)n_chrom), dim3(256), 0, d_desc,
              n_buckets);
    return SP_OK;
}

// sum and number of the counts >= lower of a byte slice (+ the overflow pairs that fall into it)
__global__ void __launch_bounds__(256)
kx_lengths(sp_tabref T, int64_t slot_base, int64_t n, uint32_t lower, unsigned long long *__restrict__ out2) {
    __shared__ unsigned long long red[16];
    unsigned long long s = 0, c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t v = T.tab[i];
        if (v < 255u && v >= lower) { s += v; c++; }
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.n_ovf; i += (int64_t)gridDim.x * blockDim.x) {
        const uint2 e = T.ovf[i];
        if ((int64_t)e.x >= slot_base && (int64_t)e.x < slot_base + n && e.y >= lower) { s += e.y; c++; }
    }
    const unsigned long long ts = sp_block_sum_u64(s, red), tc = sp_block_sum_u64(c, red);
    if (threadIdx.x == 0) {
        if (ts) atomicAdd(&out2[0], ts);
        if (tc) atomicAdd(&out2[1], tc);
    }
}

// Lay the overflow segments a counting kernel left behind out in bucket order (= ascending slot order).
static int ovf_finalize_to(sp_ctx *ctx, uint2 *out, const uint2 *tmp, const uint32_t *seg_base, const uint32_t *seg_cnt,
                           uint32_t *seg_off, int64_t n_buckets, unsigned long long *d_total = nullptr) {
    SP_LAUNCH(ctx, "ovf_scan", ovf_scan, dim3(1), dim3(1024), 0, seg_cnt, n_buckets, seg_off, d_total);
    SP_LAUNCH(ctx, "ovf_place", ovf_place<false>, dim3((unsigned)((n_buckets + 3) / 4)), dim3(256), 0, tmp, seg_base, seg_cnt,
              (const uint32_t *)seg_off, n_buckets, out, (unsigned long long *)nullptr, (uint32_t *)nullptr);
    return SP_OK;
}
int sp_ovf_finalize(sp_ctx *ctx, sp_chrom &c, const uint2 *tmp, const uint32_t *seg_base, const uint32_t *seg_cnt,
                    uint32_t *seg_off, int64_t n_buckets, unsigned long long *d_total) {
    const int rc = ovf_finalize_to(ctx, c.d_ovf, tmp, seg_base, seg_cnt, seg_off, n_buckets, d_total);
    if (rc) return rc;
    // the bucket starts live in a workspace the next chromosome reuses: keep a copy for the filter (sp_tabref::ovf_idx)
    if (c.ovf_idx_cap < n_buckets + 1) {
        if (c.d_ovf_idx) hipFree(c.d_ovf_idx);
        c.d_ovf_idx = nullptr;
        c.ovf_idx_cap = c.ovf_idx_n = 0;
        SP_HIP(ctx, hipMalloc(&c.d_ovf_idx, (size_t)(n_buckets + 1) * sizeof(uint32_t)));
        c.ovf_idx_cap = n_buckets + 1;
    }
    c.ovf_idx_n = n_buckets + 1;
    SP_HIP(ctx, hipMemcpyAsync(c.d_ovf_idx, seg_off, (size_t)(n_buckets + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    return SP_OK;
}
int sp_ovf_finalize_split(sp_ctx *ctx, unsigned long long *keys, uint32_t *cnts, const uint2 *tmp, const uint32_t *seg_base,
                          const uint32_t *seg_cnt, uint32_t *seg_off, int64_t n_buckets, unsigned long long *d_total) {
    SP_LAUNCH(ctx, "ovf_scan", ovf_scan, dim3(1), dim3(1024), 0, seg_cnt, n_buckets, seg_off, d_total);
    SP_LAUNCH(ctx, "ovf_place_list", ovf_place<true>, dim3((unsigned)((n_buckets + 3) / 4)), dim3(256), 0, tmp, seg_base,
              seg_cnt, (const uint32_t *)seg_off, n_buckets, (uint2 *)nullptr, keys, cnts);
    return SP_OK;
}

extern "C" {

static int count_impl(sp_ctx *ctx, int k, int lower_count, int engine, int first, int last) {
    if (!ctx) return SP_EINVAL;
    if (k < 1 || k > 32) return sp_fail(ctx, SP_EUNSUP, "k=%d unsupported (1..32)", k);
    if (lower_count < 1) lower_count = 1;
    if (ctx->chroms.empty()) return sp_fail(ctx, SP_EINVAL, "sp_count: no chromosomes loaded");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    if (first < 0 || last > (int)ctx->chroms.size() || first > last)
        return sp_fail(ctx, SP_EINVAL, "sp_count_range: bad chromosome range [%d, %d)", first, last);
    if (k > 15) {   // 64-bit keys: sparse engines (engine 1 = device-wide radix sort, otherwise the MSD partition)
        if (first != 0 || last != (int)ctx->chroms.size())
            return sp_fail(ctx, SP_EUNSUP, "sp_count_range: k > 15 counts all chromosomes at once");
        for (auto &c : ctx->chroms)
            if (!c.d_pk && c.len > 0) return sp_fail(ctx, SP_EINVAL, "a chromosome is not loaded");
        ctx->counted = false;
        ctx->filtered = false;
        if (ctx->d_label && ctx->k != k) {
            hipFree(ctx->d_label);
            ctx->d_label = nullptr;
        }
        if (ctx->k != k) ctx->labels_ready = false;
        ctx->k = k;
        ctx->lower = lower_count;
        ctx->nslots = 0;
        ctx->sparse_mode = true;
        ctx->list_mode = false;
        int rcs = (engine == 1) ? sp_sparse_count(ctx, k, lower_count) : sp_sparse_count3(ctx, k, lower_count);
        if (rcs) return rcs;
        ctx->counted = true;
        return SP_OK;
    }
    ctx->sparse_mode = false;
    const sp_kparams kp = sp_make_kparams(k);
    const int64_t nslots = sp_dense_slots(k);
    // Engine choice by table occupancy.  A chromosome fills at most len / nslots of its dense table; below 1/3 (the
    // Arabidopsis-like 20-Mb chromosomes at k = 15: 4 %, the peanut-like 128-Mb ones: 24 %) clearing, writing and
    // streaming 512 MiB per chromosome through the filter costs more than it saves, so the counts are kept as LISTS of
    // (slot, count >= lower) pairs in ascending slot order instead (engine 3: the same partition chain ending in
    // c2_count_list), joined by the list filter.  Measured per pass: Arabidopsis-like 13.0 -> 6.3 ms, peanut-like
    // 35.8 -> 31.6 ms; a wheat-like chromosome (670 Mb: more k-mers than slots) stays on byte tables.
    // Whole-genome calls on library-owned tables only (the multi-GPU table exchange needs the byte tables).
    bool list_mode = false;
    {
        const char *env3 = getenv("SP_LIST_ENGINE");      // "0": never, "1": whenever possible (tests)
        bool whole = first == 0 && last == (int)ctx->chroms.size();
        bool possible = whole && sp_engine2_supported(nslots) && ctx->chroms.size() <= 64;
        int64_t longest = 0;
        for (auto &c : ctx->chroms) {
            possible = possible && !c.tab_external;
            longest = c.len > longest ? c.len : longest;
        }
        if (engine == 3) {
            if (!possible)
                return sp_fail(ctx, SP_EUNSUP, "count engine 3 (lists) needs k with 2^17..2^31 dense slots, a whole-genome call, "
                                               "library-owned tables and at most 64 chromosomes");
            list_mode = true;
        } else if (engine == 0 && possible) {
            list_mode = (env3 && env3[0] == '1') || (!(env3 && env3[0] == '0') && longest > 0 && longest * 3 < nslots);
        }
    }
    // a new k invalidates old tables
    if (ctx->k != k || ctx->nslots != nslots) {
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (auto &c : ctx->chroms)
            if (c.d_tab && !c.tab_external) {   // caller-bound tables are sized by the caller for this k
                hipFree(c.d_tab);
                c.d_tab = nullptr;
            }
        if (ctx->d_label) {
            hipFree(ctx->d_label);
            ctx->d_label = nullptr;
        }
        ctx->labels_ready = false;
    }
    ctx->k = k;
    ctx->lower = lower_count;
    ctx->nslots = nslots;
    ctx->counted = false;
    ctx->filtered = false;
    ctx->list_mode = list_mode;
    const size_t C = ctx->chroms.size();
    if (list_mode && ctx->sparse.size() != C) {
        sp_sparse_release(ctx);
        ctx->sparse.assign(C, sp_sparse_chrom());
    }
    // per chromosome: sum, n (counts >= lower), overflow pairs, engine 2's region-overrun flag (a buffer of its own: the
    // lanes below clear their entries on their own streams, next to whatever the main stream still runs)
    int rc = sp_buf_ensure(ctx, ctx->b_cntlen, (int64_t)(4 * C * sizeof(unsigned long long)));
    if (rc) return rc;
    unsigned long long *d_len = (unsigned long long *)ctx->b_cntlen.p;
    const char *env_exact = getenv("SP_C2_EXACT");      // "1": size the buckets from the full histogram (round-2 path)
    const bool sized_by_sample = !(env_exact && env_exact[0] == '1');
    std::vector<char> by_engine2(C, 0);
    // small genomes (engine 3): count chromosomes on up to three more streams side by side (SP_LANES=0: one stream)
    // Byte tables (engine 2, whole-genome calls): the chains run on SP_LANES_DENSE + 1 auxiliary streams (default 4)
    // and NOT on the context's stream.  A chromosome's chain has five single-workgroup kernels (sample histogram,
    // offsets, tile starts, overflow scan / place: ~0.15 ms of a 3-ms chain during which the chip idles) and its big
    // kernels are bound by different things; side by side they fill each other's gaps (130.5 -> 120.5 ms per
    // wheat-like pass).  A lane waits for ITS chromosome's pack kernel only (sp_chrom::ev_packed), so counting starts
    // while the context's stream is still packing the chromosomes behind it.
    int n_lanes = 0;
    const bool dense_lanes = !list_mode && C > 1 && first == 0 && last == (int)C && engine != 1;
    // round 5: list mode counts all its chromosomes with ONE launch per kernel type (sp_c2batch.h) instead of a chain
    // of ten launches per chromosome on four streams; SP_C2_BATCH=0 keeps the lanes (cross-check)
    const char *env_batch = getenv("SP_C2_BATCH");
    // (chromosomes below 2^26 bases: at 128 Mb -- peanut-like -- the per-chromosome chains on four streams are 3 % ahead;
    // SP_C2_BATCH=1 batches whatever the lengths)
    int64_t longest = 0;
    for (size_t ci = (size_t)first; ci < (size_t)last; ci++) longest = ctx->chroms[ci].len > longest ? ctx->chroms[ci].len : longest;
    const bool batch = list_mode && last - first > 1 && sized_by_sample && !(env_batch && env_batch[0] == '0') &&
                       (longest < (1LL << 26) || (env_batch && env_batch[0] == '1'));
    if ((list_mode && C > 1) || dense_lanes) {
        const char *el = getenv(list_mode ? "SP_LANES" : "SP_LANES_DENSE");
        // (batched list counts: two groups since round 6 -- with c2_count_list16's two workgroups per CU a launch fills the chip
        // better, and Arabidopsis-like passes take 4.10 / 4.13 / 4.24 / 4.42 / 4.45 ms with 1 / 2 / 3 / 5 / 7 extra streams)
        n_lanes = el ? atoi(el) : (batch ? 1 : 3);
        // (toy inputs: chains that last microseconds gain nothing from side-by-side streams and pay for their events
        // and joins -- 105 against 41 ms per iteration of the fuzzer; the streams are for genomes)
        int64_t total_len = 0;
        for (size_t ci = (size_t)first; ci < (size_t)last; ci++) total_len += ctx->chroms[ci].len;
        if (!el && total_len < (1LL << 24)) n_lanes = 0;
        if (dense_lanes && n_lanes > 0) n_lanes++;       // (the context's stream takes no chain)
        if (n_lanes > SP_MAX_LANES) n_lanes = SP_MAX_LANES;
        if (n_lanes < 0) n_lanes = 0;
        {   // every lane owns a workspace of ~6 bytes per base of the longest chromosome (sp_count2.hip / sp_sparse2.hip):
            // lanes whose workspace is not allocated yet must fit what the device has free, with a margin (advisor r04)
            int64_t longest_c = 0;
            for (size_t ci = (size_t)first; ci < (size_t)last; ci++) longest_c = ctx->chroms[ci].len > longest_c ? ctx->chroms[ci].len : longest_c;
            // (a batched list count puts a whole GROUP of chromosomes into one lane's workspace: about 1 / (lanes + 1) of the
            // genome, at most one chromosome more -- advisor r05)
            const int64_t per_lane = 6 * (batch ? total_len / (int64_t)(n_lanes + 1) + longest_c : longest_c) + (256LL << 20);
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                int64_t budget = (int64_t)free_b - (int64_t)(total_b / 16);      // keep a sixteenth of the device free
                int fit = 0;
                for (int l = 0; l < n_lanes; l++) {
                    const int64_t need = ctx->lanes[l].ws2_bytes >= per_lane ? 0 : per_lane;
                    if (need > budget) break;
                    budget -= need;
                    fit++;
                }
                if (fit < n_lanes) {
                    if (getenv("SP_DEBUG_COUNT")) fprintf(stderr, "[sp] count: %d of %d lanes fit the free device memory\n", fit, n_lanes);
                    n_lanes = (dense_lanes && fit == 1) ? 0 : fit;      // (one dense lane = the plain single-stream path)
                }
            }
        }
        for (int l = 0; l < n_lanes; l++) {
            if (!ctx->lanes[l].stream) {
                SP_HIP(ctx, hipStreamCreateWithFlags(&ctx->lanes[l].stream, hipStreamNonBlocking));
                SP_HIP(ctx, hipEventCreateWithFlags(&ctx->lanes[l].done, hipEventDisableTiming));
            }
        }
        if (n_lanes && !ctx->lane_go) SP_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_go, hipEventDisableTiming));
        if (n_lanes && list_mode) {      // the lanes start after what is queued on the main stream (packing)
            SP_HIP(ctx, hipEventRecord(ctx->lane_go, ctx->stream));
            for (int l = 0; l < n_lanes; l++) SP_HIP(ctx, hipStreamWaitEvent(ctx->lanes[l].stream, ctx->lane_go, 0));
        }
    }
    // (a lambda: an error return inside must not leave the other lanes' kernels running on the shared scratch)
    auto count_all = [&]() -> int {
        if (batch) {
            for (size_t ci = (size_t)first; ci < (size_t)last; ci++) {
                sp_chrom &c = ctx->chroms[ci];
                if (!c.d_pk && c.len > 0) return sp_fail(ctx, SP_EINVAL, "chromosome %zu not loaded", ci);
                sp_sparse_chrom &o = ctx->sparse[ci];
                o.n = 0;
                o.length_sum = 0;
                int64_t need = c.len / lower_count + 16;     // (as below: every kept slot accounts for >= lower occurrences)
                if (need > nslots) need = nslots;
                if (need > o.cap) {
                    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    if (o.d_keys) hipFree(o.d_keys);
                    if (o.d_cnts) hipFree(o.d_cnts);
                    o.d_keys = nullptr;
                    o.d_cnts = nullptr;
                    o.cap = 0;
                    SP_HIP(ctx, hipMalloc(&o.d_keys, (size_t)(need + 1) * 8));
                    SP_HIP(ctx, hipMalloc(&o.d_cnts, (size_t)(need + 1) * 4));
                    o.cap = need;
                }
                by_engine2[ci] = 1;
            }
            SP_HIP(ctx, hipMemsetAsync(d_len + 4 * (size_t)first, 0, 4 * (size_t)(last - first) * sizeof(unsigned long long), ctx->stream));
            // groups of chromosomes (g, g + G, g + 2 G, ...) on the context's stream and the lanes: one launch per kernel
            // type and GROUP -- the kernels of a chain are bound by different things and fill each other's gaps side by
            // side (one group: 15.4 ms per peanut-like pass against 13.9 for per-chromosome chains on four streams)
            const int G = n_lanes + 1;
            if (n_lanes) {       // (the lanes were told to wait for lane_go above; it must follow the memset)
                SP_HIP(ctx, hipEventRecord(ctx->lane_go, ctx->stream));
                for (int l = 0; l < n_lanes; l++) SP_HIP(ctx, hipStreamWaitEvent(ctx->lanes[l].stream, ctx->lane_go, 0));
            }
            hipStream_t main_stream = ctx->stream;
            int rcb = SP_OK;
            for (int g = 0; g < G && !rcb; g++) {
                std::vector<int> idx;
                for (int ci = first + g; ci < last; ci += G) idx.push_back(ci);
                if (idx.empty()) continue;
                sp_ctx::lane_t *ln = g ? &ctx->lanes[g - 1] : nullptr;
                if (ln) {
                    ctx->lane = ln;
                    ctx->stream = ln->stream;
                }
                rcb = sp_count_engine3_batch(ctx, idx.data(), (int)idx.size(), kp, lower_count, d_len);
                ctx->lane = nullptr;
                ctx->stream = main_stream;
            }
            return rcb;
        }
        for (size_t ci = (size_t)first; ci < (size_t)last; ci++) {
            sp_chrom &c = ctx->chroms[ci];
            if (!c.d_pk && c.len > 0) return sp_fail(ctx, SP_EINVAL, "chromosome %zu not loaded", ci);
            if (list_mode) {
                sp_sparse_chrom &o = ctx->sparse[ci];
                o.n = 0;
                o.length_sum = 0;
                // every kept slot accounts for >= lower k-mer occurrences (and there are at most nslots of them)
                int64_t need = c.len / lower_count + 16;
                if (need > nslots) need = nslots;
                if (need > o.cap) {
                    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    if (o.d_keys) hipFree(o.d_keys);
                    if (o.d_cnts) hipFree(o.d_cnts);
                    o.d_keys = nullptr;
                    o.d_cnts = nullptr;
                    o.cap = 0;
                    SP_HIP(ctx, hipMalloc(&o.d_keys, (size_t)(need + 1) * 8));
                    SP_HIP(ctx, hipMalloc(&o.d_cnts, (size_t)(need + 1) * 4));
                    o.cap = need;
                }
                // lanes: chromosome ci on stream ci % (1 + lanes in use); lane 0 is the context's own stream
                sp_ctx::lane_t *ln = (n_lanes > 0 && ci % (size_t)(n_lanes + 1)) ? &ctx->lanes[ci % (size_t)(n_lanes + 1) - 1] : nullptr;
                hipStream_t main_stream = ctx->stream;
                SP_HIP(ctx, hipMemsetAsync(d_len + 4 * ci, 0, 4 * sizeof(unsigned long long), ln ? ln->stream : main_stream));
                if (ln) {       // (no early return between here and the restore below)
                    ctx->lane = ln;
                    ctx->stream = ln->stream;
                }
                rc = sp_count_engine2(ctx, c, kp, lower_count, d_len + 4 * ci, !sized_by_sample, &o);
                ctx->lane = nullptr;
                ctx->stream = main_stream;
                if (rc) return rc;
                by_engine2[ci] = 1;
                continue;
            }
            if (!c.d_tab) SP_HIP(ctx, hipMalloc(&c.d_tab, (size_t)nslots));
            // every overflow pair accounts for >= 255 k-mer occurrences
            const int64_t need_ovf = c.len / 255 + 16;
            if (need_ovf > c.cap_ovf) {
                if (c.d_ovf) {
                    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    hipFree(c.d_ovf);
                    c.d_ovf = nullptr;
                }
                SP_HIP(ctx, hipMalloc(&c.d_ovf, (size_t)need_ovf * sizeof(uint2)));
                c.cap_ovf = need_ovf;
            }
            int eng = engine;
            if (eng == 0) eng = (sp_engine2_supported(nslots) && c.len >= (1 << 22)) ? 2 : 1;
            if (eng == 2) {
                // (byte tables: the auxiliary streams only; each waits for its chromosome's pack kernel)
                sp_ctx::lane_t *ln = n_lanes > 0 ? &ctx->lanes[ci % (size_t)n_lanes] : nullptr;
                hipStream_t main_stream = ctx->stream;
                if (ln && c.ev_packed) SP_HIP(ctx, hipStreamWaitEvent(ln->stream, c.ev_packed, 0));
                // (round 6 audit: the one reader of the byte tables that returns without a host synchronisation is k3_emit
                // behind sp_filter_fetch_async; copy_event sits right behind it on the context's stream -- a count that
                // follows without sp_filter_fetch_wait must not rewrite the tables under it)
                if (ln && ctx->copy_event) SP_HIP(ctx, hipStreamWaitEvent(ln->stream, ctx->copy_event, 0));
                SP_HIP(ctx, hipMemsetAsync(d_len + 4 * ci, 0, 4 * sizeof(unsigned long long), ln ? ln->stream : main_stream));
                if (ln) {       // (no early return between here and the restore below)
                    ctx->lane = ln;
                    ctx->stream = ln->stream;
                }
                rc = sp_count_engine2(ctx, c, kp, lower_count, d_len + 4 * ci, !sized_by_sample, nullptr);
                ctx->lane = nullptr;
                ctx->stream = main_stream;
                if (rc) return rc;
                by_engine2[ci] = 1;
                continue;
            }
            // engine 1: global atomics into a u32 scratch table, then one pass to the byte table
            SP_HIP(ctx, hipMemsetAsync(d_len + 4 * ci, 0, 4 * sizeof(unsigned long long), ctx->stream));
            const int64_t n_buckets = (nslots + (1LL << SP_OVF_SHIFT) - 1) >> SP_OVF_SHIFT;
            rc = sp_buf_ensure(ctx, ctx->b_tab32, nslots * 4);
            if (rc) return rc;
            rc = sp_buf_ensure(ctx, ctx->b_ovfw, need_ovf * 8 + (3 * n_buckets + 1) * 4 + 256);
            if (rc) return rc;
            uint32_t *tab32 = (uint32_t *)ctx->b_tab32.p;
            uint2 *ovf_tmp = (uint2 *)ctx->b_ovfw.p;
            uint32_t *seg_base = (uint32_t *)(ovf_tmp + need_ovf), *seg_cnt = seg_base + n_buckets, *seg_off = seg_cnt + n_buckets;
            SP_HIP(ctx, hipMemsetAsync(tab32, 0, (size_t)nslots * sizeof(uint32_t), ctx->stream));
            int64_t n_units = (c.len + SP_UNIT - 1) / SP_UNIT;
            if (n_units > 0) {
                int grid = grid_for(ctx, n_units, 256, 16);
                SP_LAUNCH(ctx, "k1_count_atomic", k1_count_atomic, dim3(grid), dim3(256), 0, c.d_pk, c.d_nm,
                          n_units, sp_make_kparams32(k), tab32);
            }
            int grid2 = (int)(n_buckets < (int64_t)ctx->n_cu * 8 ? n_buckets : (int64_t)ctx->n_cu * 8);
            SP_LAUNCH(ctx, "k1_narrow", k1_narrow, dim3(grid2), dim3(256), 0, (const uint32_t *)tab32, nslots,
                      (uint32_t)lower_count, c.d_tab, d_len + 4 * ci, ovf_tmp, (unsigned long long)need_ovf, seg_base, seg_cnt,
                      n_buckets);
            rc = sp_ovf_finalize(ctx, c, ovf_tmp, seg_base, seg_cnt, seg_off, n_buckets, nullptr);   // (k1_narrow keeps an exact cursor in d_len[2])
            if (rc) return rc;
        }
        return SP_OK;
    };
    rc = count_all();
    if (rc) {
        const std::string msg = ctx->err;
        for (int l = 0; l < n_lanes; l++) hipStreamSynchronize(ctx->lanes[l].stream);
        hipStreamSynchronize(ctx->stream);
        ctx->err = msg;
        return rc;
    }
    for (int l = 0; l < n_lanes; l++) {      // the main stream goes on when every lane is done
        SP_HIP(ctx, hipEventRecord(ctx->lanes[l].done, ctx->lanes[l].stream));
        SP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->lanes[l].done, 0));
    }
    std::vector<unsigned long long> h(4 * C);
    SP_HIP(ctx, hipMemcpyAsync(h.data(), d_len, 4 * C * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // engine 2 laid its bucket regions out from a sample: a chromosome whose flag is up had a bucket outgrow its
    // region (keys were dropped) and is counted again with regions from the exact histogram
    int redone = 0;
    for (size_t ci = (size_t)first; ci < (size_t)last; ci++) {
        if (!by_engine2[ci] || !sized_by_sample || h[4 * ci + 3] == 0) continue;
        SP_HIP(ctx, hipMemsetAsync(d_len + 4 * ci, 0, 4 * sizeof(unsigned long long), ctx->stream));
        rc = sp_count_engine2(ctx, ctx->chroms[ci], kp, lower_count, d_len + 4 * ci, true,
                              list_mode ? &ctx->sparse[ci] : nullptr);
        if (rc) return rc;
        redone++;
    }
    if (redone) {
        if (getenv("SP_DEBUG_COUNT")) fprintf(stderr, "[sp] engine 2: %d chromosome(s) recounted with exact bucket sizes\n", redone);
        SP_HIP(ctx, hipMemcpyAsync(h.data(), d_len, 4 * C * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->c2_recounts += redone;
    for (size_t ci = (size_t)first; ci < (size_t)last; ci++) {
        if (by_engine2[ci] && h[4 * ci + 3] != 0)
            return sp_fail(ctx, SP_ESTATE, "count engine 2: chromosome %zu overran its bucket regions with exact sizes", ci);
        ctx->chroms[ci].length_sum = (int64_t)h[4 * ci];
        ctx->chroms[ci].n_dump = (int64_t)h[4 * ci + 1];
        ctx->chroms[ci].n_ovf = (int64_t)h[4 * ci + 2];
        if (list_mode) {     // the pairs ARE the dump: n_ovf of them, in ctx->sparse[ci]
            sp_sparse_chrom &o = ctx->sparse[ci];
            // (h[.. + 2] is the staging cursor: the segments reserved, an upper bound of the pairs written)
            if (ctx->chroms[ci].n_ovf > o.cap || ctx->chroms[ci].n_dump > o.cap)
                return sp_fail(ctx, SP_ESTATE, "count engine 3: chromosome %zu reserved %lld pairs for %lld dump k-mers (capacity %lld)",
                               ci, (long long)ctx->chroms[ci].n_ovf, (long long)ctx->chroms[ci].n_dump, (long long)o.cap);
            o.n = ctx->chroms[ci].n_dump;
            o.length_sum = ctx->chroms[ci].length_sum;
            ctx->chroms[ci].n_ovf = 0;
            continue;
        }
        if (ctx->chroms[ci].n_ovf > ctx->chroms[ci].cap_ovf)
            return sp_fail(ctx, SP_ESTATE, "overflow list of chromosome %zu: %lld pairs exceed the capacity %lld", ci,
                           (long long)ctx->chroms[ci].n_ovf, (long long)ctx->chroms[ci].cap_ovf);
    }
    ctx->counted = true;
    return SP_OK;
}

int sp_count(sp_ctx *ctx, int k, int lower_count, int engine) {
    return count_impl(ctx, k, lower_count, engine, 0, ctx ? (int)ctx->chroms.size() : 0);
}

int sp_count_range(sp_ctx *ctx, int k, int lower_count, int engine, int first, int last) {
    return count_impl(ctx, k, lower_count, engine, first, last);
}

int sp_count_recounts(sp_ctx *ctx, int64_t *recounts) {
    if (!ctx || !recounts) return sp_fail(ctx, SP_EINVAL, "sp_count_recounts: bad arguments");
    *recounts = ctx->c2_recounts;
    return SP_OK;
}

int sp_nslots(sp_ctx *ctx, int k, int64_t *nslots) {
    (void)ctx;
    if (!nslots || k < 1 || k > 15) return sp_fail(ctx, SP_EUNSUP, "sp_nslots: dense tables exist for k = 1..15");
    *nslots = sp_dense_slots(k);
    return SP_OK;
}

int sp_tables_bind(sp_ctx *ctx, int chrom, void *d_table) {
    if (!ctx || chrom < 0 || chrom >= (int)ctx->chroms.size())
        return sp_fail(ctx, SP_EINVAL, "sp_tables_bind: bad arguments");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    sp_chrom &c = ctx->chroms[(size_t)chrom];
    if (c.d_tab && !c.tab_external) {
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(c.d_tab);
    }
    c.d_tab = (uint8_t *)d_table;
    c.tab_external = d_table != nullptr;
    ctx->counted = false;
    return SP_OK;
}

int sp_table_overflow(sp_ctx *ctx, int chrom, void *d_pairs, int64_t cap, int64_t *n_pairs) {
    if (!ctx || !n_pairs || cap < 0 || chrom < 0 || chrom >= (int)ctx->chroms.size())
        return sp_fail(ctx, SP_EINVAL, "sp_table_overflow: bad arguments");
    if (ctx->sparse_mode || ctx->list_mode || !ctx->counted)
        return sp_fail(ctx, SP_EINVAL, "sp_table_overflow: call sp_count (k <= 15, byte-table engines 1 / 2) first");
    sp_chrom &c = ctx->chroms[(size_t)chrom];
    *n_pairs = c.n_ovf;
    if (!d_pairs) return SP_OK;   // size query
    if (cap < c.n_ovf) return sp_fail(ctx, SP_EINVAL, "sp_table_overflow: capacity %lld < %lld", (long long)cap, (long long)c.n_ovf);
    SP_HIP(ctx, hipSetDevice(ctx->device));
    if (c.n_ovf)
        SP_HIP(ctx, hipMemcpyAsync(d_pairs, c.d_ovf, (size_t)c.n_ovf * sizeof(uint2), hipMemcpyDeviceToDevice, ctx->stream));
    return SP_OK;   // asynchronous on the context's stream
}

int sp_table_merge(sp_ctx *ctx, void *d_dst_u8, const void *d_dst_ovf, int64_t n_dst_ovf, const void *d_src_u8,
                   const void *d_src_ovf, int64_t n_src_ovf, int64_t slot_base, int64_t n, void *d_out_ovf, int64_t cap,
                   int64_t *n_out) {
    if (!ctx || !d_dst_u8 || !d_src_u8 || !n_out || n < 0 || slot_base < 0 || cap < 0 || n_dst_ovf < 0 || n_src_ovf < 0 ||
        (n_dst_ovf > 0 && !d_dst_ovf) || (n_src_ovf > 0 && !d_src_ovf) || (cap > 0 && !d_out_ovf) ||
        (d_out_ovf && (d_out_ovf == d_dst_ovf || d_out_ovf == d_src_ovf)))
        return sp_fail(ctx, SP_EINVAL, "sp_table_merge: bad arguments");
    *n_out = 0;
    if (n == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n_buckets = (n + (1LL << SP_OVF_SHIFT) - 1) >> SP_OVF_SHIFT;
    int rc = sp_buf_ensure(ctx, ctx->b_ovfw, (cap + 1) * 8 + (3 * n_buckets + 1) * 4 + 256);
    if (rc) return rc;
    unsigned long long *d_cur = (unsigned long long *)ctx->b_ovfw.p;
    uint2 *tmp = (uint2 *)(d_cur + 1);
    uint32_t *seg_base = (uint32_t *)(tmp + cap), *seg_cnt = seg_base + n_buckets, *seg_off = seg_cnt + n_buckets;
    SP_HIP(ctx, hipMemsetAsync(d_cur, 0, 8, ctx->stream));
    sp_tabref A, B;
    A.tab = (const uint8_t *)d_dst_u8; A.ovf = (const uint2 *)d_dst_ovf; A.n_ovf = n_dst_ovf;
    B.tab = (const uint8_t *)d_src_u8; B.ovf = (const uint2 *)d_src_ovf; B.n_ovf = n_src_ovf;
    int grid = (int)(n_buckets < (int64_t)ctx->n_cu * 8 ? n_buckets : (int64_t)ctx->n_cu * 8);
    SP_LAUNCH(ctx, "kx_merge", kx_merge, dim3(grid), dim3(256), 0, (uint8_t *)d_dst_u8, A, B, slot_base, n, d_cur, tmp,
              (unsigned long long)cap, seg_base, seg_cnt, n_buckets);
    unsigned long long h = 0;
    SP_HIP(ctx, hipMemcpyAsync(&h, d_cur, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = (int64_t)h;
    if ((int64_t)h > cap)
        return sp_fail(ctx, SP_ENOMEM, "sp_table_merge: %lld overflow pairs exceed the capacity %lld (the tables are now "
                       "inconsistent: merge again from fresh inputs)", (long long)h, (long long)cap);
    if (h) return ovf_finalize_to(ctx, (uint2 *)d_out_ovf, tmp, seg_base, seg_cnt, seg_off, n_buckets);
    return SP_OK;
}

int sp_table_lengths(sp_ctx *ctx, const void *d_tab_u8, const void *d_ovf, int64_t n_ovf, int64_t slot_base, int64_t n,
                     int lower_count, int64_t *sum, int64_t *n_dump) {
    if (!ctx || !d_tab_u8 || !sum || !n_dump || n < 0 || n_ovf < 0 || (n_ovf > 0 && !d_ovf))
        return sp_fail(ctx, SP_EINVAL, "sp_table_lengths: bad arguments");
    *sum = *n_dump = 0;
    if (n == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    void *scr = nullptr;
    int rc = sp_scratch(ctx, 256, &scr);
    if (rc) return rc;
    unsigned long long *d2 = (unsigned long long *)scr, h[2] = {0, 0};
    SP_HIP(ctx, hipMemsetAsync(d2, 0, 16, ctx->stream));
    sp_tabref T;
    T.tab = (const uint8_t *)d_tab_u8; T.ovf = (const uint2 *)d_ovf; T.n_ovf = n_ovf;
    SP_LAUNCH(ctx, "kx_lengths", kx_lengths, dim3((unsigned)(ctx->n_cu * 4)), dim3(256), 0, T, slot_base, n,
              (uint32_t)(lower_count < 1 ? 1 : lower_count), d2);
    SP_HIP(ctx, hipMemcpyAsync(h, d2, 16, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *sum = (int64_t)h[0];
    *n_dump = (int64_t)h[1];
    return SP_OK;
}

int sp_lengths(sp_ctx *ctx, int64_t *lengths) {
    if (!ctx || !lengths) return sp_fail(ctx, SP_EINVAL, "sp_lengths: bad arguments");
    if (!ctx->counted) return sp_fail(ctx, SP_EINVAL, "sp_lengths: call sp_count first");
    for (size_t i = 0; i < ctx->chroms.size(); i++) lengths[i] = ctx->chroms[i].length_sum;
    return SP_OK;
}

int sp_dump_size(sp_ctx *ctx, int chrom, int64_t *n) {
    if (!ctx || !n || chrom < 0 || chrom >= (int)ctx->chroms.size())
        return sp_fail(ctx, SP_EINVAL, "sp_dump_size: bad arguments");
    if (!ctx->counted) return sp_fail(ctx, SP_EINVAL, "sp_dump_size: call sp_count first");
    *n = ctx->chroms[(size_t)chrom].n_dump;
    return SP_OK;
}

int sp_dump(sp_ctx *ctx, int chrom, uint64_t *keys, uint32_t *counts, int64_t cap, int64_t *n) {
    if (!ctx || !n || chrom < 0 || chrom >= (int)ctx->chroms.size())
        return sp_fail(ctx, SP_EINVAL, "sp_dump: bad arguments");
    if (!ctx->counted) return sp_fail(ctx, SP_EINVAL, "sp_dump: call sp_count first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    sp_chrom &c = ctx->chroms[(size_t)chrom];
    *n = c.n_dump;
    if (cap < c.n_dump) return sp_fail(ctx, SP_EINVAL, "sp_dump: capacity %lld < %lld", (long long)cap,
                                       (long long)c.n_dump);
    if (c.n_dump == 0) return SP_OK;
    if (ctx->sparse_mode || ctx->list_mode) return sp_sparse_dump(ctx, chrom, keys, counts);
    const sp_kparams kp = sp_make_kparams(ctx->k);
    int64_t nblk = (ctx->nslots + DUMP_SLOTS - 1) / DUMP_SLOTS;
    sp_tmp<unsigned long long> d_blk, d_keys;
    sp_tmp<uint32_t> d_cnt;
    SP_HIP(ctx, d_blk.alloc((size_t)nblk));
    SP_HIP(ctx, d_keys.alloc((size_t)c.n_dump));
    SP_HIP(ctx, d_cnt.alloc((size_t)c.n_dump));
    sp_tabref T;
    T.tab = c.d_tab;
    T.ovf = c.d_ovf;
    T.n_ovf = c.n_ovf;
    SP_LAUNCH(ctx, "dump_count", dump_count, dim3((unsigned)nblk), dim3(DUMP_BLOCK), 0, T,
              ctx->nslots, (uint32_t)ctx->lower, d_blk.p);
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, d_blk.p, nblk,
              (unsigned long long *)nullptr);
    SP_LAUNCH(ctx, "dump_write", dump_write, dim3((unsigned)nblk), dim3(DUMP_BLOCK), 0, T,
              ctx->nslots, (uint32_t)ctx->lower, d_blk.p, kp, d_keys.p, d_cnt.p);
    SP_HIP(ctx, hipMemcpyAsync(keys, d_keys, (size_t)c.n_dump * sizeof(uint64_t), hipMemcpyDeviceToHost,
                              ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(counts, d_cnt, (size_t)c.n_dump * sizeof(uint32_t), hipMemcpyDeviceToHost,
                              ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}
}  // extern "C"
