// sp_filter.hip -- K3: chromosome x k-mer matrix (outer join of the per-
// chromosome tables) + SubPhaser's differential-k-mer filter.
//
// Replaces JellyfishDumps.to_matrix (Jellyfish.py:439-460) and
// JellyfishDumps.filter / _filter_kmer (Jellyfish.py:462-512, 611-648).
//
// Pass A (k3_eval) streams all C dense tables once (4*C bytes per slot, fully
// coalesced: lane i reads slot base+i of every table), evaluates the filter in
// fp64 with the reference's operation order and writes two slot bitmaps
// (row = differential k-mer, hist = fold-passing) plus per-block popcounts.
// Pass B (k3_emit) walks the bitmap (64 MiB at k=15) and gathers the M
// surviving rows, in ascending slot order, so the output is deterministic.
#include "sp_device.h"

#ifndef F_BLOCK
#define F_BLOCK 256
#endif
#ifndef F_GROUPS_PER_WAVE
#define F_GROUPS_PER_WAVE 64
#endif
#define F_WAVES (F_BLOCK / 64)
#define F_SLOTS_PER_BLOCK (F_WAVES * F_GROUPS_PER_WAVE * 64)  // 16384
#define F_MAXU 8

struct sp_filter_params {
    int C;
    int n_sets;
    int baseline;
    uint32_t lower;
    double min_fold, min_freq, max_freq, ratio;
    int64_t nslots;
};

// The per-k-mer decision of _filter_kmer (Jellyfish.py:611-648), shared by the dense (k3_eval) and
// the sparse (k > 15) engines.  cnt[c * stride] = thresholded count of chromosome c.
struct sp_fsets {
    int n_sets, baseline;
    const int32_t *set_off, *unit_off, *unit_chrom;
    const double *unit_den, *unit_inv;   // per-unit denominators and their reciprocals
    double min_fold, min_freq, max_freq, ratio;
};

__device__ __forceinline__ void sp_filter_decide(const uint32_t *cnt, int stride, unsigned long long tot,
                                                 const sp_fsets &F, bool &is_row, bool &is_hist) {
    is_row = is_hist = false;
    int include = 0, all = 0;
    for (int s = 0; s < F.n_sets; s++) {
        const int u0 = F.set_off[s], nu = F.set_off[s + 1] - u0;
        if (nu == 1) continue;  // singleton ignored (Jellyfish.py:621-622)
        all++;
        // descending order statistic: hi = f_(0), lo = f_(bi)   (:637-639)
        const int bi = F.baseline < 0 ? nu + F.baseline : F.baseline;
        double hi, lo;
        if (bi == 1 || bi == nu - 1) {
            // The two values the CLI allows (baseline 1 / -1) need only the running max, second max and
            // min.  k3_eval is issue-bound on this fp64 code (16-18 ms against 8.8 ms for the table
            // reads alone), so the set is first screened in fp32 on products with the precomputed
            // reciprocals: fp32 moves hi and lo by a relative 1e-6 at most, so outside a 1e-5 band
            // around the threshold the screen and the reference's fp64 quotient test agree; inside the
            // band the quotients are formed exactly as the reference does (:630-641).
            {
                float m1 = -1.0f, m2 = -1.0f, mn = 3e38f;
                for (int u = 0; u < nu; u++) {
                    unsigned long long num = 0;
                    for (int j = F.unit_off[u0 + u]; j < F.unit_off[u0 + u + 1]; j++)
                        num += cnt[F.unit_chrom[j] * stride];
                    const float x = (float)num * (float)F.unit_inv[u0 + u];
                    if (x > m1) {
                        m2 = m1;
                        m1 = x;
                    } else if (x > m2) {
                        m2 = x;
                    }
                    mn = x < mn ? x : mn;
                }
                const float thr = (float)F.min_fold * (((bi == 1) ? m2 : mn) + 1e-20f);
                if (m1 > thr * (1.0f + 1e-5f)) {
                    include++;
                    continue;
                }
                if (m1 < thr * (1.0f - 1e-5f)) continue;
            }
            double m1 = -1.0, m2 = -1.0, mn = 1e300;
            for (int u = 0; u < nu; u++) {
                unsigned long long num = 0;
                for (int j = F.unit_off[u0 + u]; j < F.unit_off[u0 + u + 1]; j++)
                    num += cnt[F.unit_chrom[j] * stride];
                const double x = (double)num / F.unit_den[u0 + u];  // count/len or sum/sum (:630,:634)
                if (x > m1) {
                    m2 = m1;
                    m1 = x;
                } else if (x > m2) {
                    m2 = x;
                }
                mn = x < mn ? x : mn;
            }
            hi = m1;
            lo = (bi == 1) ? m2 : mn;
        } else {
            double f[F_MAXU];
#pragma unroll
            for (int u = 0; u < F_MAXU; u++) {
                f[u] = 0.0;
                if (u < nu) {
                    unsigned long long num = 0;
                    for (int j = F.unit_off[u0 + u]; j < F.unit_off[u0 + u + 1]; j++)
                        num += cnt[F.unit_chrom[j] * stride];
                    f[u] = (double)num / F.unit_den[u0 + u];
                }
            }
            hi = f[0];
            lo = f[0];
#pragma unroll
            for (int u = 0; u < F_MAXU; u++) {
                if (u < nu) {
                    hi = f[u] > hi ? f[u] : hi;
                    int rank = 0;
#pragma unroll
                    for (int v = 0; v < F_MAXU; v++)
                        if (v < nu && (f[v] > f[u] || (f[v] == f[u] && v < u))) rank++;
                    if (rank == bi) lo = f[u];
                }
            }
        }
        if (1.0 * hi / (lo + 1e-20) >= F.min_fold) include++;  // :640-641
    }
    const double r = 1.0 * (double)include / (double)all;  // :642
    if (!(r < F.ratio)) {
        is_hist = true;
        const double t = (double)tot;
        is_row = !(t < F.min_freq || t > F.max_freq);  // :645-646
    }
}

__global__ void __launch_bounds__(F_BLOCK)
k3_eval(const uint32_t *const *__restrict__ tabs, sp_filter_params P,
        const int32_t *__restrict__ set_off, const int32_t *__restrict__ unit_off,
        const int32_t *__restrict__ unit_chrom, const double *__restrict__ unit_den,
        unsigned long long *__restrict__ bm_row, unsigned long long *__restrict__ bm_hist,
        unsigned long long *__restrict__ blk_row, unsigned long long *__restrict__ blk_hist,
        unsigned long long *__restrict__ n_union) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_cnt[];  // [4 waves][C][64]
    __shared__ unsigned long long red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *mine = lds_cnt + (size_t)wave * P.C * 64;
    unsigned long long nrow = 0, nhist = 0, nuni = 0;
    const int64_t blk_base = (int64_t)blockIdx.x * F_SLOTS_PER_BLOCK;
    for (int g = 0; g < F_GROUPS_PER_WAVE; g++) {
        const int64_t gbase = blk_base + ((int64_t)wave * F_GROUPS_PER_WAVE + g) * 64;
        if (gbase >= P.nslots) break;
        const int64_t slot = gbase + lane;
        const bool in = slot < P.nslots;
        unsigned long long tot = 0;
        // eight independent table loads in flight per lane (one load at a time is latency-bound:
        // 1.8 TB/s measured; batching reaches the streaming rate)
        for (int c0 = 0; c0 < P.C; c0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (in && c0 + j < P.C) ? tabs[c0 + j][slot] : 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (c0 + j < P.C) {
                    const uint32_t x = v[j] >= P.lower ? v[j] : 0u;
                    mine[(c0 + j) * 64 + lane] = x;
                    tot += x;
                }
            }
        }
        bool is_row = false, is_hist = false;
        if (tot > 0) {
            sp_fsets F;
            F.n_sets = P.n_sets;
            F.baseline = P.baseline;
            F.set_off = set_off;
            F.unit_off = unit_off;
            F.unit_chrom = unit_chrom;
            F.unit_den = unit_den;
            F.unit_inv = unit_den + set_off[P.n_sets];
            F.min_fold = P.min_fold;
            F.min_freq = P.min_freq;
            F.max_freq = P.max_freq;
            F.ratio = P.ratio;
            sp_filter_decide(mine + lane, 64, tot, F, is_row, is_hist);
        }
        const unsigned long long b_row = __ballot(is_row), b_hist = __ballot(is_hist),
                                 b_uni = __ballot(tot > 0);
        if (lane == 0) {
            bm_row[gbase >> 6] = b_row;
            bm_hist[gbase >> 6] = b_hist;
        }
        nrow += __popcll(b_row);
        nhist += __popcll(b_hist);
        nuni += __popcll(b_uni);
    }
    // every lane of a wave carries the same tallies: keep lane 0's
    if (lane != 0) nrow = nhist = nuni = 0;
    unsigned long long t_row = sp_block_sum_u64(nrow, red);
    unsigned long long t_hist = sp_block_sum_u64(nhist, red);
    unsigned long long t_uni = sp_block_sum_u64(nuni, red);
    if (threadIdx.x == 0) {
        blk_row[blockIdx.x] = t_row;
        blk_hist[blockIdx.x] = t_hist;
        if (t_uni) atomicAdd(n_union, t_uni);
    }
}

// scan defined in sp_count.hip
__global__ void scan_excl_u64(unsigned long long *a, int64_t n, unsigned long long *total);

// Pass B, step 1: ordered list of the surviving slots (bitmap walk only; the table gathers of a row used to
// run on the one lane that owned its slot -- one active lane per wave at 0.4 % density)
__global__ void __launch_bounds__(F_BLOCK)
k3_emit_slots(int64_t nslots, const unsigned long long *__restrict__ bm, const unsigned long long *__restrict__ blk_off,
              uint32_t *__restrict__ slots) {
    __shared__ unsigned long long wave_cnt[F_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t blk_base = (int64_t)blockIdx.x * F_SLOTS_PER_BLOCK;
    const int64_t g0 = (blk_base >> 6) + (int64_t)wave * F_GROUPS_PER_WAVE;
    const int64_t ngroups = (nslots + 63) >> 6;
    // wave totals first so that waves write disjoint, ordered ranges
    unsigned long long mycnt = 0;
    for (int g = lane; g < F_GROUPS_PER_WAVE; g += 64)
        if (g0 + g < ngroups) mycnt += __popcll(bm[g0 + g]);
    for (int o = 32; o > 0; o >>= 1) mycnt += __shfl_down(mycnt, o, 64);
    if (lane == 0) wave_cnt[wave] = mycnt;
    __syncthreads();
    unsigned long long off = blk_off[blockIdx.x];
    for (int w = 0; w < wave; w++) off += wave_cnt[w];
    for (int g = 0; g < F_GROUPS_PER_WAVE; g++) {
        if (g0 + g >= ngroups) break;
        const unsigned long long bits = bm[g0 + g];
        if (bits == 0) continue;
        if ((bits >> lane) & 1ULL)
            slots[off + __popcll(bits & ((1ULL << lane) - 1ULL))] = (uint32_t)(((g0 + g) << 6) + lane);
        off += __popcll(bits);
    }
}

// Pass B, step 2: one thread per surviving row gathers its C counts (independent loads, every lane busy)
__global__ void __launch_bounds__(256)
k3_emit(const uint32_t *const *__restrict__ tabs, int C, uint32_t lower, int64_t M, int64_t slot_base, sp_kparams kp,
        const uint32_t *__restrict__ slots, const double *__restrict__ chrom_len,
        unsigned long long *__restrict__ keys, uint32_t *__restrict__ counts,
        double *__restrict__ freqs, unsigned long long *__restrict__ tots) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const int64_t slot = slots[r];
    unsigned long long tot = 0;
    for (int c = 0; c < C; c++) {
        uint32_t v = tabs[c][slot];
        v = v >= lower ? v : 0u;
        tot += v;
        if (counts) counts[r * C + c] = v;
        if (freqs) freqs[r * C + c] = (double)v / chrom_len[c];  // :647
    }
    if (keys) keys[r] = sp_key_of_slot((uint64_t)(slot_base + slot), kp);
    if (tots) tots[r] = tot;
}

static void free_filter_buffers(sp_ctx *ctx) {
    if (ctx->d_flag_row) hipFree(ctx->d_flag_row);
    if (ctx->d_flag_hist) hipFree(ctx->d_flag_hist);
    if (ctx->d_blk_row) hipFree(ctx->d_blk_row);
    if (ctx->d_blk_hist) hipFree(ctx->d_blk_hist);
    ctx->d_flag_row = ctx->d_flag_hist = nullptr;
    ctx->d_blk_row = ctx->d_blk_hist = nullptr;
    ctx->filtered = false;
}

// small device-side parameter block kept in the scratch buffer
struct filter_dev {
    const uint32_t **tabs;
    double *chrom_len;
};

static int filter_C(sp_ctx *ctx) {
    return ctx->sv_on ? (int)ctx->sv_keys.size() : ctx->fv_on ? (int)ctx->fv_tabs.size() : (int)ctx->chroms.size();
}
static const uint32_t *filter_tab(sp_ctx *ctx, int i) {
    return ctx->fv_on ? ctx->fv_tabs[(size_t)i] : ctx->chroms[(size_t)i].d_tab;
}
static int64_t filter_len(sp_ctx *ctx, int i) {
    return (ctx->fv_on || ctx->sv_on) ? ctx->fv_lengths[(size_t)i] : ctx->chroms[(size_t)i].length_sum;
}
static int64_t filter_nslots(sp_ctx *ctx) { return ctx->fv_on ? ctx->fv_nslots : ctx->nslots; }
static int64_t filter_base(sp_ctx *ctx) { return ctx->fv_on ? ctx->fv_slot_base : 0; }

static int upload_tabs(sp_ctx *ctx, const uint32_t ***d_tabs, double **d_len) {
    const size_t C = (size_t)filter_C(ctx);
    size_t bytes = C * sizeof(void *) + C * sizeof(double);
    void *scr = nullptr;
    int rc = sp_scratch(ctx, (int64_t)bytes + 4096, &scr);
    if (rc) return rc;
    std::vector<const uint32_t *> h(C);
    std::vector<double> hl(C);
    for (size_t i = 0; i < C; i++) {
        h[i] = filter_tab(ctx, (int)i);
        hl[i] = (double)filter_len(ctx, (int)i);
    }
    SP_HIP(ctx, hipMemcpyAsync(scr, h.data(), C * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync((char *)scr + C * sizeof(void *), hl.data(), C * sizeof(double),
                              hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // h/hl go out of scope
    *d_tabs = (const uint32_t **)scr;
    *d_len = (double *)((char *)scr + C * sizeof(void *));
    return SP_OK;
}

int sp_sparse_filter(sp_ctx *ctx, int n_sets, const int32_t *set_off, const int32_t *unit_off,
                     const int32_t *unit_chrom, const std::vector<double> &den, double min_fold, int baseline,
                     double min_freq, double max_freq, double ratio);                      // sp_sparse.hip
int sp_sparse_fetch(sp_ctx *ctx, bool hist, uint64_t *keys, uint32_t *counts, double *freqs, uint64_t *tot);

extern "C" {

int sp_filter_view(sp_ctx *ctx, int C, const void *const *d_tabs, int64_t slot_base, int64_t nslots_view,
                   const int64_t *lengths, int k, int lower_count) {
    if (!ctx) return SP_EINVAL;
    if (d_tabs && ctx->sparse_mode) return sp_fail(ctx, SP_EUNSUP, "sp_filter_view: k <= 15 only");
    if (!d_tabs) {   // back to the local chromosomes
        ctx->fv_on = false;
        ctx->fv_tabs.clear();
        ctx->fv_lengths.clear();
        ctx->filtered = false;
        return SP_OK;
    }
    if (C <= 0 || !lengths || slot_base < 0 || nslots_view <= 0 || (slot_base % 64) != 0 || k < 1 || k > 15)
        return sp_fail(ctx, SP_EINVAL, "sp_filter_view: bad arguments (slot_base must be a multiple of 64)");
    ctx->fv_tabs.assign((size_t)C, nullptr);
    ctx->fv_lengths.assign((size_t)C, 0);
    for (int i = 0; i < C; i++) {
        if (!d_tabs[i]) return sp_fail(ctx, SP_EINVAL, "sp_filter_view: table %d is NULL", i);
        ctx->fv_tabs[(size_t)i] = (const uint32_t *)d_tabs[i];
        ctx->fv_lengths[(size_t)i] = lengths[i];
    }
    ctx->fv_slot_base = slot_base;
    ctx->fv_nslots = nslots_view;
    ctx->fv_on = true;
    ctx->filtered = false;
    if (ctx->k == 0) {   // a rank that owns no chromosome still filters its slot range
        ctx->k = k;
        ctx->nslots = sp_dense_slots(k);
    }
    if (ctx->k != k) return sp_fail(ctx, SP_EINVAL, "sp_filter_view: k=%d but the context counted with k=%d", k, ctx->k);
    ctx->lower = lower_count < 1 ? 1 : lower_count;
    return SP_OK;
}

int sp_filter(sp_ctx *ctx, int n_sets, const int32_t *set_off, const int32_t *unit_off,
              const int32_t *unit_chrom, double min_fold, int baseline, double min_freq,
              double max_freq, double ratio, int64_t *n_union, int64_t *n_rows, int64_t *n_hist) {
    if (!ctx || !set_off || !unit_off || !unit_chrom || n_sets <= 0)
        return sp_fail(ctx, SP_EINVAL, "sp_filter: bad arguments");
    if (!ctx->counted && !ctx->fv_on && !ctx->sv_on) return sp_fail(ctx, SP_EINVAL, "sp_filter: call sp_count first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int C = filter_C(ctx);
    // the reference's precondition checks, same messages (Jellyfish.py:474-489)
    if (min_freq > max_freq)
        return sp_fail(ctx, SP_ESTATE, "`min_freq` (%g) should be lower than `max_freq` (%g)", min_freq,
                       max_freq);
    int n_single = 0, n_units = set_off[n_sets];
    for (int s = 0; s < n_sets; s++) {
        int nu = set_off[s + 1] - set_off[s];
        if (nu == 1) n_single++;
        if (nu > F_MAXU)
            return sp_fail(ctx, SP_EUNSUP, "a homoeologous set has %d subgenome columns; this build supports <= %d",
                           nu, F_MAXU);
        if (nu > 1) {
            int bi = baseline < 0 ? nu + baseline : baseline;
            if (bi < 0 || bi >= nu) return sp_fail(ctx, SP_ESTATE, "list index out of range (baseline=%d)", baseline);
        }
    }
    if (n_single == n_sets) return sp_fail(ctx, SP_ESTATE, "All singletons are not allowed");
    for (int i = 0; i < C; i++)
        if (filter_len(ctx, i) == 0)
            return sp_fail(ctx, SP_ESTATE, "Chromosomes `[%d]` have only 0 kmers", i);
    const int n_uc = unit_off[n_units];
    for (int j = 0; j < n_uc; j++)
        if (unit_chrom[j] < 0 || unit_chrom[j] >= C)
            return sp_fail(ctx, SP_EINVAL, "sp_filter: chromosome index %d out of range", unit_chrom[j]);

    if (ctx->sparse_mode) {
        std::vector<double> den_s((size_t)n_units * 2);   // denominators, then their reciprocals
        for (int u = 0; u < n_units; u++) {
            int64_t d = 0;
            for (int j = unit_off[u]; j < unit_off[u + 1]; j++) d += filter_len(ctx, unit_chrom[j]);
            den_s[(size_t)u] = (double)d;
            den_s[(size_t)(n_units + u)] = 1.0 / (double)d;
        }
        int rcs = sp_sparse_filter(ctx, n_sets, set_off, unit_off, unit_chrom, den_s, min_fold, baseline, min_freq,
                                   max_freq, ratio);
        if (rcs) return rcs;
        if (n_union) *n_union = ctx->n_union;
        if (n_rows) *n_rows = ctx->n_rows;
        if (n_hist) *n_hist = ctx->n_hist;
        return SP_OK;
    }
    const int64_t nslots = filter_nslots(ctx);
    const int64_t nblk = (nslots + F_SLOTS_PER_BLOCK - 1) / F_SLOTS_PER_BLOCK;
    const int64_t ngroups = (nslots + 63) / 64;
    ctx->filtered = false;
    if (!ctx->d_flag_row || ctx->n_fblocks != nblk) {   // bitmaps are reused across calls
        free_filter_buffers(ctx);
        SP_HIP(ctx, hipMalloc(&ctx->d_flag_row, (size_t)ngroups * 8));
        SP_HIP(ctx, hipMalloc(&ctx->d_flag_hist, (size_t)ngroups * 8));
        SP_HIP(ctx, hipMalloc(&ctx->d_blk_row, (size_t)(nblk + 1) * 8));
        SP_HIP(ctx, hipMalloc(&ctx->d_blk_hist, (size_t)(nblk + 1) * 8));
    }
    ctx->n_fblocks = nblk;

    // device copies of the set structure + per-unit denominators
    std::vector<double> den((size_t)n_units * 2);   // denominators, then their reciprocals
    for (int u = 0; u < n_units; u++) {
        int64_t d = 0;
        for (int j = unit_off[u]; j < unit_off[u + 1]; j++) d += filter_len(ctx, unit_chrom[j]);
        den[(size_t)u] = (double)d;
        den[(size_t)(n_units + u)] = 1.0 / (double)d;
    }
    size_t b_set = (size_t)(n_sets + 1) * 4, b_uo = (size_t)(n_units + 1) * 4, b_uc = (size_t)(n_uc > 0 ? n_uc : 1) * 4,
           b_den = (size_t)n_units * 16;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t tot_b = al(b_set) + al(b_uo) + al(b_uc) + al(b_den) + al(C * sizeof(void *)) + 256;
    int rcb = sp_buf_ensure(ctx, ctx->b_fpar, (int64_t)tot_b);
    if (rcb) return rcb;
    char *d_par = (char *)ctx->b_fpar.p;
    char *p = d_par;
    int32_t *d_set = (int32_t *)p; p += al(b_set);
    int32_t *d_uo = (int32_t *)p; p += al(b_uo);
    int32_t *d_uc = (int32_t *)p; p += al(b_uc);
    double *d_den = (double *)p; p += al(b_den);
    const uint32_t **d_tabs = (const uint32_t **)p; p += al(C * sizeof(void *));
    unsigned long long *d_nuni = (unsigned long long *)p;   // [0] union count, [1..2] scan totals
    std::vector<const uint32_t *> htabs((size_t)C);
    for (int i = 0; i < C; i++) htabs[(size_t)i] = filter_tab(ctx, i);
    hipError_t e = hipSuccess;
    auto cp = [&](void *d, const void *h, size_t n) {
        if (e == hipSuccess && n) e = hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->stream);
    };
    cp(d_set, set_off, b_set);
    cp(d_uo, unit_off, b_uo);
    cp(d_uc, unit_chrom, (size_t)n_uc * 4);
    cp(d_den, den.data(), b_den);
    cp(d_tabs, htabs.data(), C * sizeof(void *));
    if (e == hipSuccess) e = hipMemsetAsync(d_nuni, 0, 32, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // host staging vectors go out of scope
    if (e != hipSuccess)
        return sp_fail(ctx, SP_EHIP, "sp_filter: parameter upload failed: %s", hipGetErrorString(e));
    sp_filter_params P;
    P.C = C;
    P.n_sets = n_sets;
    P.baseline = baseline;
    P.lower = (uint32_t)ctx->lower;
    P.min_fold = min_fold;
    P.min_freq = min_freq;
    P.max_freq = max_freq;
    P.ratio = ratio;
    P.nslots = nslots;
    size_t shmem = (size_t)F_WAVES * C * 64 * sizeof(uint32_t);
    if (shmem > 150 * 1024)
        return sp_fail(ctx, SP_EUNSUP, "sp_filter: %d chromosomes exceed the LDS staging budget", C);
    if (shmem > 64 * 1024)
        hipFuncSetAttribute((const void *)k3_eval, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    SP_LAUNCH(ctx, "k3_eval", k3_eval, dim3((unsigned)nblk), dim3(F_BLOCK), shmem, d_tabs, P, d_set, d_uo,
              d_uc, d_den, (unsigned long long *)ctx->d_flag_row, (unsigned long long *)ctx->d_flag_hist,
              (unsigned long long *)ctx->d_blk_row, (unsigned long long *)ctx->d_blk_hist, d_nuni);
    unsigned long long *d_tot = d_nuni + 1;
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0,
              (unsigned long long *)ctx->d_blk_row, nblk, d_tot);
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0,
              (unsigned long long *)ctx->d_blk_hist, nblk, d_tot + 1);
    unsigned long long h[3] = {0, 0, 0};
    SP_HIP(ctx, hipMemcpyAsync(h, d_tot, 16, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(h + 2, d_nuni, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_rows = (int64_t)h[0];
    ctx->n_hist = (int64_t)h[1];
    ctx->n_union = (int64_t)h[2];
    ctx->filtered = true;
    if (n_union) *n_union = ctx->n_union;
    if (n_rows) *n_rows = ctx->n_rows;
    if (n_hist) *n_hist = ctx->n_hist;
    return SP_OK;
}

static int emit_common(sp_ctx *ctx, bool hist, uint64_t *keys, uint32_t *counts, double *freqs,
                       uint64_t *tot, int64_t cap) {
    if (!ctx->filtered) return sp_fail(ctx, SP_EINVAL, "call sp_filter first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t M = hist ? ctx->n_hist : ctx->n_rows;
    if (cap < M) return sp_fail(ctx, SP_EINVAL, "capacity %lld < %lld rows", (long long)cap, (long long)M);
    if (M == 0) return SP_OK;
    if (ctx->sparse_mode) return sp_sparse_fetch(ctx, hist, keys, counts, freqs, tot);
    const int C = filter_C(ctx);
    const uint32_t **d_tabs = nullptr;
    double *d_len = nullptr;
    int rc = upload_tabs(ctx, &d_tabs, &d_len);
    if (rc) return rc;
    unsigned long long *d_keys = nullptr, *d_tot = nullptr;
    uint32_t *d_counts = nullptr;
    double *d_freqs = nullptr;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t need = (keys ? al((size_t)M * 8) : 0) + (tot ? al((size_t)M * 8) : 0) +
                  (counts ? al((size_t)M * C * 4) : 0) + (freqs ? al((size_t)M * C * 8) : 0);
    rc = sp_buf_ensure(ctx, ctx->b_emit, (int64_t)need);
    if (rc) return rc;
    {
        char *q = (char *)ctx->b_emit.p;
        if (keys) { d_keys = (unsigned long long *)q; q += al((size_t)M * 8); }
        if (tot) { d_tot = (unsigned long long *)q; q += al((size_t)M * 8); }
        if (counts) { d_counts = (uint32_t *)q; q += al((size_t)M * C * 4); }
        if (freqs) { d_freqs = (double *)q; q += al((size_t)M * C * 8); }
    }
    const sp_kparams kp = sp_make_kparams(ctx->k);
    rc = sp_buf_ensure(ctx, ctx->b_slots, M * 4 + 64);
    if (rc) return rc;
    uint32_t *d_slots = (uint32_t *)ctx->b_slots.p;
    SP_LAUNCH(ctx, "k3_emit_slots", k3_emit_slots, dim3((unsigned)ctx->n_fblocks), dim3(F_BLOCK), 0, filter_nslots(ctx),
              (const unsigned long long *)(hist ? ctx->d_flag_hist : ctx->d_flag_row),
              (const unsigned long long *)(hist ? ctx->d_blk_hist : ctx->d_blk_row), d_slots);
    SP_LAUNCH(ctx, hist ? "k3_emit_hist" : "k3_emit", k3_emit, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, d_tabs, C,
              (uint32_t)ctx->lower, M, filter_base(ctx), kp, (const uint32_t *)d_slots, (const double *)d_len, d_keys,
              d_counts, d_freqs, d_tot);
    if (keys) SP_HIP(ctx, hipMemcpyAsync(keys, d_keys, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (tot) SP_HIP(ctx, hipMemcpyAsync(tot, d_tot, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (counts) SP_HIP(ctx, hipMemcpyAsync(counts, d_counts, (size_t)M * C * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (freqs) SP_HIP(ctx, hipMemcpyAsync(freqs, d_freqs, (size_t)M * C * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_filter_fetch(sp_ctx *ctx, uint64_t *keys, uint32_t *counts, double *freqs, uint64_t *tot,
                    int64_t cap_rows) {
    if (!ctx) return SP_EINVAL;
    return emit_common(ctx, false, keys, counts, freqs, tot, cap_rows);
}

int sp_filter_fetch_device(sp_ctx *ctx, void *d_keys, void *d_counts, void *d_tot, int64_t cap_rows) {
    if (!ctx) return SP_EINVAL;
    if (!ctx->filtered) return sp_fail(ctx, SP_EINVAL, "call sp_filter first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t M = ctx->n_rows;
    if (cap_rows < M) return sp_fail(ctx, SP_EINVAL, "capacity %lld < %lld rows", (long long)cap_rows, (long long)M);
    if (M == 0) return SP_OK;
    const int C = filter_C(ctx);
    if (ctx->sparse_mode) {   // the sparse filter leaves its rows in device buffers already
        if (d_keys) SP_HIP(ctx, hipMemcpyAsync(d_keys, ctx->b_sf_keys.p, (size_t)M * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_counts) SP_HIP(ctx, hipMemcpyAsync(d_counts, ctx->b_sf_counts.p, (size_t)M * C * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_tot) SP_HIP(ctx, hipMemcpyAsync(d_tot, ctx->b_sf_tot.p, (size_t)M * 8, hipMemcpyDeviceToDevice, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return SP_OK;
    }
    const uint32_t **d_tabs = nullptr;
    double *d_len = nullptr;
    int rc = upload_tabs(ctx, &d_tabs, &d_len);
    if (rc) return rc;
    const sp_kparams kp = sp_make_kparams(ctx->k);
    rc = sp_buf_ensure(ctx, ctx->b_slots, M * 4 + 64);
    if (rc) return rc;
    uint32_t *d_slots = (uint32_t *)ctx->b_slots.p;
    SP_LAUNCH(ctx, "k3_emit_slots", k3_emit_slots, dim3((unsigned)ctx->n_fblocks), dim3(F_BLOCK), 0, filter_nslots(ctx),
              (const unsigned long long *)ctx->d_flag_row, (const unsigned long long *)ctx->d_blk_row, d_slots);
    SP_LAUNCH(ctx, "k3_emit", k3_emit, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, d_tabs, C, (uint32_t)ctx->lower, M,
              filter_base(ctx), kp, (const uint32_t *)d_slots, (const double *)d_len, (unsigned long long *)d_keys,
              (uint32_t *)d_counts, (double *)nullptr, (unsigned long long *)d_tot);
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_filter_hist(sp_ctx *ctx, uint64_t *tot, int64_t cap) {
    if (!ctx || !tot) return sp_fail(ctx, SP_EINVAL, "sp_filter_hist: bad arguments");
    return emit_common(ctx, true, nullptr, nullptr, nullptr, tot, cap);
}
}  // extern "C"
