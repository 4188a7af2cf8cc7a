// sp_count.hip -- K1 (canonical k-mer counting into a dense per-chromosome
// table), K2 (lower_count threshold + lengths), dump compaction.
//
// Replaces the external `jellyfish count --canonical | dump -c -L` step
// (reference: subphaser/Jellyfish.py:671-704).
#include "sp_device.h"

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

// ----------------------------------------------------------------- K2
// lengths[c] = sum of counts >= lower (Jellyfish.py:97,449) and the number of
// such k-mers (the dump size).  Pure streaming read of the table.
__global__ void __launch_bounds__(256)
k2_lengths(const uint32_t *__restrict__ tab, int64_t nslots, uint32_t lower,
           unsigned long long *__restrict__ out /*[0]=sum,[1]=n*/) {
    __shared__ unsigned long long red[16];
    const int64_t n4 = nslots >> 2;
    const uint4 *t4 = reinterpret_cast<const uint4 *>(tab);
    unsigned long long s = 0, n = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        uint4 v = t4[i];
        if (v.x >= lower) { s += v.x; n++; }
        if (v.y >= lower) { s += v.y; n++; }
        if (v.z >= lower) { s += v.z; n++; }
        if (v.w >= lower) { s += v.w; n++; }
    }
    if (blockIdx.x == 0)  // tail when nslots is not a multiple of 4 (tiny k)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < nslots; i += blockDim.x)
            if (tab[i] >= lower) { s += tab[i]; n++; }
    unsigned long long ts = sp_block_sum_u64(s, red);
    unsigned long long tn = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0) {
        if (ts) atomicAdd(&out[0], ts);
        if (tn) atomicAdd(&out[1], tn);
    }
}

// ----------------------------------------------------------------- dump
#define DUMP_PER_THREAD 16
#define DUMP_BLOCK 256
#define DUMP_SLOTS (DUMP_PER_THREAD * DUMP_BLOCK)

__global__ void __launch_bounds__(DUMP_BLOCK)
dump_count(const uint32_t *__restrict__ tab, int64_t nslots, uint32_t lower,
           unsigned long long *__restrict__ blk) {
    __shared__ unsigned long long red[16];
    int64_t base = (int64_t)blockIdx.x * DUMP_SLOTS;
    unsigned long long n = 0;
    for (int j = 0; j < DUMP_PER_THREAD; j++) {
        int64_t i = base + (int64_t)j * DUMP_BLOCK + threadIdx.x;
        if (i < nslots && tab[i] >= lower) n++;
    }
    unsigned long long t = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0) blk[blockIdx.x] = t;
}

// single-block exclusive scan of n u64 values (n up to a few million)
__global__ void __launch_bounds__(1024)
scan_excl_u64(unsigned long long *__restrict__ a, int64_t n, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long part[1024];
    const int T = 1024;
    int64_t per = (n + T - 1) / T;
    int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per;
    if (hi > n) hi = n;
    unsigned long long s = 0;
    for (int64_t i = lo; i < hi; i++) s += a[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < T; i++) {
            unsigned long long v = part[i];
            part[i] = run;
            run += v;
        }
        if (total) *total = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int64_t i = lo; i < hi; i++) {
        unsigned long long v = a[i];
        a[i] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(DUMP_BLOCK)
dump_write(const uint32_t *__restrict__ tab, int64_t nslots, uint32_t lower,
           const unsigned long long *__restrict__ blk, sp_kparams kp,
           unsigned long long *__restrict__ keys, uint32_t *__restrict__ counts) {
    __shared__ uint32_t lds[16];
    int64_t base = (int64_t)blockIdx.x * DUMP_SLOTS;
    unsigned long long off = blk[blockIdx.x];
    for (int j = 0; j < DUMP_PER_THREAD; j++) {
        int64_t i = base + (int64_t)j * DUMP_BLOCK + threadIdx.x;
        uint32_t c = (i < nslots) ? tab[i] : 0u;
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
                     unsigned long long *d_len2);  // sp_count2.hip
bool sp_engine2_supported(int64_t nslots);
int sp_sparse_count(sp_ctx *ctx, int k, int lower);                                   // sp_sparse.hip
int sp_sparse_count3(sp_ctx *ctx, int k, int lower);                                  // sp_sparse2.hip
int sp_sparse_dump(sp_ctx *ctx, int chrom, uint64_t *keys, uint32_t *counts);

// ------------------------------------------------------------------ wire format of a count table
// Multi-GPU runs ship slot-range slices of the count tables over xGMI.  Counts that matter are
// thresholded (>= lower_count) and almost all of them are small, so a table travels as one byte per
// slot: 0 = below the threshold, 1..254 = the count, 255 = "look in the overflow list" (slot, count).
// 4x less wire volume than the u32 table: at 2 and 4 GPUs the exchange is bound by ONE xGMI link per peer.
#define KX_STAGE 2048   // overflow pairs staged in LDS before one global reservation
__global__ void __launch_bounds__(256)
kx_narrow(const uint32_t *__restrict__ tab, int64_t n4 /* nslots / 4 */, uint32_t lower, uint32_t *__restrict__ out,
          uint2 *__restrict__ ovf, unsigned long long cap, unsigned long long *__restrict__ n_ovf) {
    // counts >= 255 are appended block-wise: a few million single-address global atomics (one per entry,
    // or even one per wave) cost more than streaming the 2-GiB table (2.1 ms against 0.6 ms)
    __shared__ uint2 stage[KX_STAGE + 1024];
    __shared__ uint32_t n_stage;
    __shared__ unsigned long long g_base;
    if (threadIdx.x == 0) n_stage = 0;
    __syncthreads();
    const uint4 *t4 = reinterpret_cast<const uint4 *>(tab);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n_iter = (n4 + stride - 1) / stride;
    for (int64_t it = 0; it <= n_iter; it++) {
        const int64_t i = it * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (it < n_iter && i < n4) {
            const uint4 v = t4[i];
            const uint32_t a[4] = {v.x, v.y, v.z, v.w};
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t c = a[j] >= lower ? a[j] : 0u;
                if (c >= 255u) {
                    stage[atomicAdd(&n_stage, 1u)] = make_uint2((uint32_t)(4 * i + j), c);
                    c = 255u;
                }
                packed |= c << (8 * j);
            }
            out[i] = packed;
        }
        __syncthreads();
        const uint32_t m = n_stage;
        if (m >= KX_STAGE || (it == n_iter && m > 0)) {   // block-uniform
            if (threadIdx.x == 0) g_base = atomicAdd(n_ovf, (unsigned long long)m);
            __syncthreads();
            for (uint32_t p = threadIdx.x; p < m; p += blockDim.x)
                if (g_base + p < cap) ovf[g_base + p] = stage[p];
            __syncthreads();
            if (threadIdx.x == 0) n_stage = 0;
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(256)
kx_widen(const uint32_t *__restrict__ in, int64_t n4, uint4 *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t p = in[i];
        out[i] = make_uint4(p & 255u, (p >> 8) & 255u, (p >> 16) & 255u, p >> 24);
    }
}

__global__ void __launch_bounds__(256)
kx_patch(uint32_t *__restrict__ tab, int64_t slot_base, int64_t n, const uint2 *__restrict__ ovf, int64_t n_ovf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ovf) return;
    const uint2 e = ovf[i];
    const int64_t s = (int64_t)e.x - slot_base;
    if (s >= 0 && s < n) tab[s] = e.y;
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
        int rcs = (engine == 1) ? sp_sparse_count(ctx, k, lower_count) : sp_sparse_count3(ctx, k, lower_count);
        if (rcs) return rcs;
        ctx->counted = true;
        return SP_OK;
    }
    ctx->sparse_mode = false;
    const sp_kparams kp = sp_make_kparams(k);
    const int64_t nslots = sp_dense_slots(k);
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
    const size_t C = ctx->chroms.size();
    void *scr = nullptr;
    int rc = sp_scratch(ctx, (int64_t)(2 * C * sizeof(unsigned long long)), &scr);
    if (rc) return rc;
    unsigned long long *d_len = (unsigned long long *)scr;
    SP_HIP(ctx, hipMemsetAsync(d_len, 0, 2 * C * sizeof(unsigned long long), ctx->stream));
    for (size_t ci = (size_t)first; ci < (size_t)last; ci++) {
        sp_chrom &c = ctx->chroms[ci];
        if (!c.d_pk && c.len > 0) return sp_fail(ctx, SP_EINVAL, "chromosome %zu not loaded", ci);
        if (!c.d_tab) SP_HIP(ctx, hipMalloc(&c.d_tab, (size_t)nslots * sizeof(uint32_t)));
        int eng = engine;
        if (eng == 0) eng = (sp_engine2_supported(nslots) && c.len >= (1 << 22)) ? 2 : 1;
        if (eng == 2) {
            rc = sp_count_engine2(ctx, c, kp, lower_count, d_len + 2 * ci);
            if (rc) return rc;
            continue;
        }
        SP_HIP(ctx, hipMemsetAsync(c.d_tab, 0, (size_t)nslots * sizeof(uint32_t), ctx->stream));
        int64_t n_units = (c.len + SP_UNIT - 1) / SP_UNIT;
        if (n_units > 0) {
            int grid = grid_for(ctx, n_units, 256, 16);
            SP_LAUNCH(ctx, "k1_count_atomic", k1_count_atomic, dim3(grid), dim3(256), 0, c.d_pk, c.d_nm,
                      n_units, sp_make_kparams32(k), c.d_tab);
        }
        int grid2 = grid_for(ctx, nslots / 4, 256, 16);
        SP_LAUNCH(ctx, "k2_lengths", k2_lengths, dim3(grid2), dim3(256), 0, c.d_tab, nslots,
                  (uint32_t)lower_count, d_len + 2 * ci);
    }
    std::vector<unsigned long long> h(2 * C);
    SP_HIP(ctx, hipMemcpyAsync(h.data(), d_len, 2 * C * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t ci = (size_t)first; ci < (size_t)last; ci++) {
        ctx->chroms[ci].length_sum = (int64_t)h[2 * ci];
        ctx->chroms[ci].n_dump = (int64_t)h[2 * ci + 1];
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

int sp_nslots(sp_ctx *ctx, int k, int64_t *nslots) {
    (void)ctx;
    if (!nslots || k < 1 || k > 15) return sp_fail(ctx, SP_EUNSUP, "sp_nslots: dense tables exist for k = 1..15");
    *nslots = sp_dense_slots(k);
    return SP_OK;
}

int sp_tables_bind(sp_ctx *ctx, int chrom, void *d_table) {
    if (!ctx || chrom < 0 || chrom >= (int)ctx->chroms.size())
        return sp_fail(ctx, SP_EINVAL, "sp_tables_bind: bad arguments");
    if (ctx->sparse_mode && d_table) return sp_fail(ctx, SP_EUNSUP, "sp_tables_bind: dense tables exist for k <= 15 only");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    sp_chrom &c = ctx->chroms[(size_t)chrom];
    if (c.d_tab && !c.tab_external) {
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(c.d_tab);
    }
    c.d_tab = (uint32_t *)d_table;
    c.tab_external = d_table != nullptr;
    ctx->counted = false;
    return SP_OK;
}

int sp_table_narrow(sp_ctx *ctx, int chrom, void *d_out_u8, void *d_ovf, int64_t cap, int64_t *n_ovf) {
    if (!ctx || !d_out_u8 || !n_ovf || cap < 0 || (cap > 0 && !d_ovf) || chrom < 0 || chrom >= (int)ctx->chroms.size())
        return sp_fail(ctx, SP_EINVAL, "sp_table_narrow: bad arguments");
    if (ctx->sparse_mode || !ctx->chroms[(size_t)chrom].d_tab || ctx->nslots <= 0)
        return sp_fail(ctx, SP_EINVAL, "sp_table_narrow: chromosome %d has no dense count table", chrom);
    if (ctx->nslots % 4) return sp_fail(ctx, SP_EUNSUP, "sp_table_narrow: table of %lld slots", (long long)ctx->nslots);
    SP_HIP(ctx, hipSetDevice(ctx->device));
    void *scr = nullptr;
    int rc = sp_scratch(ctx, 256, &scr);
    if (rc) return rc;
    unsigned long long *d_n = (unsigned long long *)scr;
    SP_HIP(ctx, hipMemsetAsync(d_n, 0, 8, ctx->stream));
    SP_LAUNCH(ctx, "kx_narrow", kx_narrow, dim3((unsigned)(ctx->n_cu * 8)), dim3(256), 0,
              (const uint32_t *)ctx->chroms[(size_t)chrom].d_tab, ctx->nslots / 4, (uint32_t)ctx->lower,
              (uint32_t *)d_out_u8, (uint2 *)d_ovf, (unsigned long long)cap, d_n);
    unsigned long long h = 0;
    SP_HIP(ctx, hipMemcpyAsync(&h, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_ovf = (int64_t)h;
    if ((int64_t)h > cap)
        return sp_fail(ctx, SP_ENOMEM, "sp_table_narrow: %lld counts >= 255 exceed the overflow capacity %lld",
                       (long long)h, (long long)cap);
    return SP_OK;
}

int sp_table_widen(sp_ctx *ctx, const void *d_in_u8, int64_t n, void *d_out_u32) {
    if (!ctx || !d_in_u8 || !d_out_u32 || n < 0 || (n % 4)) return sp_fail(ctx, SP_EINVAL, "sp_table_widen: bad arguments (n must be a multiple of 4)");
    if (n == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_LAUNCH(ctx, "kx_widen", kx_widen, dim3((unsigned)(ctx->n_cu * 8)), dim3(256), 0, (const uint32_t *)d_in_u8, n / 4,
              (uint4 *)d_out_u32);
    return SP_OK;   // asynchronous on the context's stream
}

int sp_table_patch(sp_ctx *ctx, void *d_tab_u32, int64_t slot_base, int64_t n, const void *d_ovf, int64_t n_ovf) {
    if (!ctx || !d_tab_u32 || slot_base < 0 || n < 0 || n_ovf < 0 || (n_ovf > 0 && !d_ovf))
        return sp_fail(ctx, SP_EINVAL, "sp_table_patch: bad arguments");
    if (n_ovf == 0 || n == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_LAUNCH(ctx, "kx_patch", kx_patch, dim3((unsigned)((n_ovf + 255) / 256)), dim3(256), 0, (uint32_t *)d_tab_u32,
              slot_base, n, (const uint2 *)d_ovf, n_ovf);
    return SP_OK;   // asynchronous on the context's stream
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
    if (ctx->sparse_mode) return sp_sparse_dump(ctx, chrom, keys, counts);
    const sp_kparams kp = sp_make_kparams(ctx->k);
    int64_t nblk = (ctx->nslots + DUMP_SLOTS - 1) / DUMP_SLOTS;
    sp_tmp<unsigned long long> d_blk, d_keys;
    sp_tmp<uint32_t> d_cnt;
    SP_HIP(ctx, d_blk.alloc((size_t)nblk));
    SP_HIP(ctx, d_keys.alloc((size_t)c.n_dump));
    SP_HIP(ctx, d_cnt.alloc((size_t)c.n_dump));
    SP_LAUNCH(ctx, "dump_count", dump_count, dim3((unsigned)nblk), dim3(DUMP_BLOCK), 0, c.d_tab,
              ctx->nslots, (uint32_t)ctx->lower, d_blk.p);
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, d_blk.p, nblk,
              (unsigned long long *)nullptr);
    SP_LAUNCH(ctx, "dump_write", dump_write, dim3((unsigned)nblk), dim3(DUMP_BLOCK), 0, c.d_tab,
              ctx->nslots, (uint32_t)ctx->lower, d_blk.p, kp, d_keys.p, d_cnt.p);
    SP_HIP(ctx, hipMemcpyAsync(keys, d_keys, (size_t)c.n_dump * sizeof(uint64_t), hipMemcpyDeviceToHost,
                              ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(counts, d_cnt, (size_t)c.n_dump * sizeof(uint32_t), hipMemcpyDeviceToHost,
                              ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}
}  // extern "C"
