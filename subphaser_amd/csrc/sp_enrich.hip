// sp_enrich.hip -- K6: per-window Fisher right tail + enrichment decision.
//
// Replaces Stats.fisher_test / enrich / _enrich / Pvalues.get_enriched
// (reference: subphaser/Stats.py:14-31, 140-192).  One thread per window.
//
// The hypergeometric point mass is evaluated with Loader's saddle-point
// formulation (stirlerr / bd0), which stays accurate to ~1e-14 relative for
// margins of 2e8 where a difference of lgamma() values loses 6-7 digits; the
// tail is then summed by the exact term recurrence.  All arithmetic is fp64.
#include "sp_device.h"

#define SP_MAX_INT_CLAMP (2147483647LL / 10)  // Stats.py:9
#define SP_ENRICH_MAXS 32

__device__ static const double c_sfe[16] = {
    0.0,                           /* 0: unused */
    0.08106146679532726,           /* 1 */
    0.04134069595540929,           /* 2 */
    0.02767792568499834,           /* 3 */
    0.02079067210376509,           /* 4 */
    0.01664469118982119,           /* 5 */
    0.01387612882307075,           /* 6 */
    0.01189670994589177,           /* 7 */
    0.010411265261972096,          /* 8 */
    0.009255462182712733,          /* 9 */
    0.008330563433362871,          /* 10 */
    0.007573675487951841,          /* 11 */
    0.006942840107209530,          /* 12 */
    0.006408994188004207,          /* 13 */
    0.005951370112758848,          /* 14 */
    0.005554733551962801           /* 15 */
};

__device__ __forceinline__ double d_stirlerr(double n) {
    const double S0 = 0.083333333333333333333, S1 = 0.00277777777777777777778,
                 S2 = 0.00079365079365079365079365, S3 = 0.000595238095238095238095238,
                 S4 = 0.0008417508417508417508417508;
    if (n <= 15.0) return c_sfe[(int)n];
    double nn = n * n;
    if (n > 500) return (S0 - S1 / nn) / n;
    if (n > 80) return (S0 - (S1 - S2 / nn) / nn) / n;
    if (n > 35) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
    return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}

__device__ __forceinline__ double d_bd0(double x, double np) {
    if (fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        if (fabs(s) < 2.2250738585072014e-308) return s;
        double ej = 2 * x * v;
        v = v * v;
        for (int j = 1; j < 1000; j++) {
            ej *= v;
            double s1 = s + ej / ((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
    }
    return x * log(x / np) + np - x;
}

// log of the binomial point mass, Loader's dbinom_raw
__device__ __forceinline__ double d_ldbinom(double x, double n, double p, double q) {
    const double NEG_INF = -INFINITY;
    if (p == 0) return x == 0 ? 0.0 : NEG_INF;
    if (q == 0) return x == n ? 0.0 : NEG_INF;
    if (x == 0) {
        if (n == 0) return 0.0;
        return (p < 0.1) ? -d_bd0(n, n * q) - n * p : n * log(q);
    }
    if (x == n) return (q < 0.1) ? -d_bd0(n, n * p) - n * q : n * log(p);
    if (x < 0 || x > n) return NEG_INF;
    double lc = d_stirlerr(n) - d_stirlerr(x) - d_stirlerr(n - x) - d_bd0(x, n * p) - d_bd0(n - x, n * q);
    double lf = 1.837877066409345483560659472811 + log(x) + log1p(-x / n);
    return lc - 0.5 * lf;
}

// P[X = x], x white drawn when n are drawn from r white + b black
__device__ __forceinline__ double d_dhyper(double x, double r, double b, double n) {
    if (n < x || r < x || n - x > b) return 0.0;
    if (n == 0) return x == 0 ? 1.0 : 0.0;
    double p = n / (r + b), q = (r + b - n) / (r + b);
    double l = d_ldbinom(x, r, p, q) + d_ldbinom(n - x, b, p, q) - d_ldbinom(n, r + b, p, q);
    return exp(l);
}

// P[X >= a], X ~ Hypergeom(N = a+b+c+d, K = a+b, n = a+c): what
// fisher.pvalue(a,b,c,d).right_tail returns (Stats.py:26)
__device__ double d_right_tail(long long a, long long b, long long c, long long d) {
    const double K = (double)(a + b), NK = (double)(c + d), n = (double)(a + c);
    const double N = K + NK;
    long long lo = (a + c) - (c + d);
    if (lo < 0) lo = 0;
    const long long hi = (a + b) < (a + c) ? (a + b) : (a + c);
    if (a <= lo) return 1.0;
    if (a > hi) return 0.0;
    const double mode = floor((n + 1.0) * (K + 1.0) / (N + 2.0));
    if ((double)a > mode) {
        double x = (double)a;
        double term = d_dhyper(x, K, NK, n);
        double s = term;
        while (x < (double)hi && term > 0.0) {
            term *= (K - x) * (n - x) / ((x + 1.0) * (NK - n + x + 1.0));
            x += 1.0;
            s += term;
            if (term < s * 1e-18) break;
        }
        return s;
    }
    double x = (double)a - 1.0;
    double term = d_dhyper(x, K, NK, n);
    double s = term;
    while (x > (double)lo && term > 0.0) {
        term *= x * (NK - n + x) / ((K - x + 1.0) * (n - x + 1.0));
        x -= 1.0;
        s += term;
        if (term < s * 1e-18) break;
    }
    return 1.0 - s;
}

// numpy's float64 add.reduce order for a short contiguous vector
__device__ __forceinline__ double d_np_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

__global__ void __launch_bounds__(64)
k6_enrich(const long long *__restrict__ counts, const long long *__restrict__ total /* S column sums, then their sum */,
          long long W, int S, double max_pval, double min_ratio, double *__restrict__ pvals,
          int *__restrict__ argmin, unsigned char *__restrict__ sig, double *__restrict__ ratios) {
    long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const long long sum_total = total[S];
    const long long *row = counts + w * S;
    double *p = pvals + w * S;
    double *q = ratios + w * S;
    long long sum_each = 0;
    for (int j = 0; j < S; j++) sum_each += row[j];
    for (int j = 0; j < S; j++) {
        // fisher_test margins incl. the x22 quirk and clamps (Stats.py:20-25)
        long long x11 = row[j];
        long long x12 = sum_each - x11;
        long long x21 = total[j] - x11;
        long long x22 = sum_total - x21 - x12;
        if (x21 > SP_MAX_INT_CLAMP) x21 = SP_MAX_INT_CLAMP;
        if (x22 > SP_MAX_INT_CLAMP) x22 = SP_MAX_INT_CLAMP;
        p[j] = d_right_tail(x11, x12, x21, x22);
    }
    // Pvalues.get_enriched (Stats.py:181-192): stable sort by p, first two
    int m = 0;
    for (int j = 1; j < S; j++)
        if (p[j] < p[m]) m = j;
    int s2 = -1;
    for (int j = 0; j < S; j++) {
        if (j == m) continue;
        if (s2 < 0 || p[j] < p[s2]) s2 = j;
    }
    bool sg = true;
    if (p[m] > max_pval) sg = false;
    if (p[m] == 0) {
    } else if (p[s2] / p[m] < max_pval / p[s2] * 1.0)
        sg = false;
    // _enrich (Stats.py:157-162)
    for (int j = 0; j < S; j++) q[j] = (double)row[j] / (double)total[j];
    double qs = d_np_sum(q, S);
    for (int j = 0; j < S; j++) q[j] = q[j] / qs;
    if (q[m] < min_ratio) sg = false;
    argmin[w] = m;
    sig[w] = sg ? 1 : 0;
}

// ----------------------------------------------------------------- f-1: subgenome-specific k-mer test
// Cluster.output_kmers / _output_kmers (Cluster.py:151-194) for test_method = ttest_ind: per differential
// k-mer, the chromosome frequencies count/length are grouped by subgenome, the groups are ordered by mean
// (descending, ties in subgenome order) and scipy.stats.ttest_ind(top, second) -- pooled variance, two-sided --
// gives the p-value: 2 * stdtr(df, -|t|) = I_x(df/2, 1/2) with x = df / (df + t^2) (regularised incomplete beta,
// continued fraction in fp64).  One thread per k-mer.
#define SP_TT_MAXG 64
__device__ double d_betacf(double a, double b, double x) {
    const double tiny = 1e-300;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 500; m++) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c;
        if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        // a few eps: 1e-16 lies below the spacing of doubles around 1 (1.1e-16 / 2.2e-16) and only ever fired on
        // del == 1.0 exactly -- otherwise every thread ran all 500 iterations.  p-values agree with scipy's
        // stdtr to a relative 1e-9 (tests/test_gpu_parity.py::test_kmer_ttest_device_vs_scipy), not bit for bit.
        if (fabs(del - 1.0) < 3e-16) break;
    }
    return h;
}
__device__ double d_betainc(double a, double b, double x) {
    if (!(x > 0.0)) return 0.0;
    if (!(x < 1.0)) return 1.0;
    const double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x));
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * d_betacf(a, b, x) / a;
    return 1.0 - bt * d_betacf(b, a, 1.0 - x) / b;
}

__global__ void __launch_bounds__(128)
k7_ttest(const uint32_t *__restrict__ counts, long long M, int C, const double *__restrict__ chrom_len, int G,
         const int *__restrict__ goff, const int *__restrict__ gchrom, int *__restrict__ top, int *__restrict__ second,
         double *__restrict__ pvals, double *__restrict__ means) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const uint32_t *row = counts + r * C;
    int t1 = -1, t2 = -1;
    double m1 = 0, m2 = 0;
    for (int g = 0; g < G; g++) {            // means in numpy's summation order; descending, ties in group order
        double v[SP_TT_MAXG];
        const int n = goff[g + 1] - goff[g];
        for (int i = 0; i < n; i++) v[i] = (double)row[gchrom[goff[g] + i]] / chrom_len[gchrom[goff[g] + i]];
        const double mean = d_np_sum(v, n) / (double)n;
        means[r * G + g] = mean;
        if (t1 < 0 || mean > m1) { t2 = t1; m2 = m1; t1 = g; m1 = mean; }
        else if (t2 < 0 || mean > m2) { t2 = g; m2 = mean; }
    }
    if (t2 < 0) t2 = t1;
    top[r] = t1;
    second[r] = t2;
    double var[2], mu[2];
    int nn[2];
    for (int s = 0; s < 2; s++) {
        const int g = s ? t2 : t1;
        double v[SP_TT_MAXG];
        const int n = goff[g + 1] - goff[g];
        for (int i = 0; i < n; i++) v[i] = (double)row[gchrom[goff[g] + i]] / chrom_len[gchrom[goff[g] + i]];
        const double mean = d_np_sum(v, n) / (double)n;
        for (int i = 0; i < n; i++) { const double d = v[i] - mean; v[i] = d * d; }
        var[s] = n > 1 ? d_np_sum(v, n) / (double)(n - 1) : 0.0;
        mu[s] = mean;
        nn[s] = n;
    }
    const double df = (double)(nn[0] + nn[1]) - 2.0;
    double p;
    if (!(df > 0.0)) {
        p = __longlong_as_double(0x7ff8000000000000LL);
    } else {
        const double svar = ((nn[0] - 1) * var[0] + (nn[1] - 1) * var[1]) / df;
        const double denom = sqrt(svar * (1.0 / nn[0] + 1.0 / nn[1]));
        const double t = (mu[0] - mu[1]) / denom;
        if (t != t) p = t;                               // 0 / 0: NaN, kept by the caller like the reference does
        else if (isinf(t)) p = 0.0;
        else p = d_betainc(0.5 * df, 0.5, df / (df + t * t));
    }
    pvals[r] = p;
}

extern "C" int sp_kmer_ttest(sp_ctx *ctx, const uint32_t *counts, int64_t M, int C, const int64_t *lengths, int n_groups,
                             const int32_t *group_off, const int32_t *group_chrom, int32_t *top, int32_t *second,
                             double *pvals, double *means) {
    if (!ctx || M < 0 || C < 1 || n_groups < 1 || !lengths || !group_off || !group_chrom ||
        (M > 0 && (!counts || !top || !second || !pvals || !means)))
        return sp_fail(ctx, SP_EINVAL, "sp_kmer_ttest: bad arguments");
    for (int g = 0; g < n_groups; g++) {
        const int n = group_off[g + 1] - group_off[g];
        if (n < 1 || n > SP_TT_MAXG) return sp_fail(ctx, SP_EUNSUP, "sp_kmer_ttest: a subgenome with %d chromosomes (1..%d supported)", n, SP_TT_MAXG);
        for (int j = group_off[g]; j < group_off[g + 1]; j++)
            if (group_chrom[j] < 0 || group_chrom[j] >= C) return sp_fail(ctx, SP_EINVAL, "sp_kmer_ttest: chromosome index out of range");
    }
    if (M == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nuc = (size_t)group_off[n_groups];
    // `counts` may be a host or a DEVICE pointer (a caller that staged the rows earlier, Context.stage_rows): rows
    // that already live on this device are read in place -- no second M x C copy inside the workspace
    bool on_device = false;
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, counts) == hipSuccess)
            on_device = at.type == hipMemoryTypeDevice && at.device == ctx->device;
        else
            (void)hipGetLastError();   // a plain malloc'ed pointer is "invalid value" for this query on some runtimes
    }
    const size_t rows_bytes = on_device ? 0 : al((size_t)M * C * 4);
    const size_t need = rows_bytes + al((size_t)C * 8) + al((size_t)(n_groups + 1) * 4) + al(nuc * 4) +
                        2 * al((size_t)M * 4) + al((size_t)M * 8) + al((size_t)M * n_groups * 8);
    int rc = sp_buf_ensure(ctx, ctx->b_tt, (int64_t)need);
    if (rc) return rc;
    char *q = (char *)ctx->b_tt.p;
    const uint32_t *d_counts = on_device ? counts : (const uint32_t *)q; q += rows_bytes;
    double *d_len = (double *)q; q += al((size_t)C * 8);
    int *d_goff = (int *)q; q += al((size_t)(n_groups + 1) * 4);
    int *d_gch = (int *)q; q += al(nuc * 4);
    int *d_top = (int *)q; q += al((size_t)M * 4);
    int *d_sec = (int *)q; q += al((size_t)M * 4);
    double *d_p = (double *)q; q += al((size_t)M * 8);
    double *d_means = (double *)q;
    std::vector<double> hl((size_t)C);
    for (int c = 0; c < C; c++) hl[(size_t)c] = (double)lengths[c];
    if (!on_device)
        SP_HIP(ctx, hipMemcpyAsync((void *)d_counts, counts, (size_t)M * C * 4, hipMemcpyDefault, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_len, hl.data(), (size_t)C * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_goff, group_off, (size_t)(n_groups + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_gch, group_chrom, nuc * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_LAUNCH(ctx, "k7_ttest", k7_ttest, dim3((unsigned)((M + 127) / 128)), dim3(128), 0, d_counts,
              (long long)M, C, (const double *)d_len, n_groups, (const int *)d_goff, (const int *)d_gch, d_top, d_sec, d_p,
              d_means);
    SP_HIP(ctx, hipMemcpyAsync(top, d_top, (size_t)M * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(second, d_sec, (size_t)M * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(pvals, d_p, (size_t)M * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(means, d_means, (size_t)M * n_groups * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

// column sums of the window table (Stats.py:142) + their sum, on the device
__global__ void __launch_bounds__(256)
k6_totals(const long long *__restrict__ counts, long long W, int S, long long *__restrict__ total /* S + 1 */) {
    __shared__ unsigned long long acc[SP_ENRICH_MAXS + 1];
    if (threadIdx.x <= S) acc[threadIdx.x] = 0;
    __syncthreads();
    for (int j = 0; j < S; j++) {
        unsigned long long s = 0;
        for (long long w = threadIdx.x; w < W; w += blockDim.x) s += (unsigned long long)counts[w * S + j];
        if (s) atomicAdd(&acc[j], s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int j = 0; j < S; j++) {
            total[j] = (long long)acc[j];
            t += acc[j];
        }
        total[S] = (long long)t;
    }
}

extern "C" int sp_enrich_dev(sp_ctx *ctx, const void *d_counts, int64_t W, int S, double max_pval, double min_ratio,
                             double *pvals, int32_t *argmin, uint8_t *sig, double *ratios) {
    if (!ctx || W < 0 || (W > 0 && (!d_counts || !pvals || !argmin || !sig || !ratios)))
        return sp_fail(ctx, SP_EINVAL, "sp_enrich: bad arguments");
    if (S < 2) return sp_fail(ctx, SP_ESTATE, "sp_enrich: at least 2 subgenome columns required (Stats.py:172)");
    if (S > SP_ENRICH_MAXS) return sp_fail(ctx, SP_EUNSUP, "sp_enrich: S=%d > %d", S, SP_ENRICH_MAXS);
    if (W == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nWS = (size_t)W * S;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // one growth-only buffer: totals | p | ratios | argmin | sig   (no per-call hipMalloc)
    int rc = sp_buf_ensure(ctx, ctx->b_enr, (int64_t)(al((size_t)(S + 1) * 8) + 2 * al(nWS * 8) + al((size_t)W * 4) + al((size_t)W)));
    if (rc) return rc;
    char *q = (char *)ctx->b_enr.p;
    long long *d_total = (long long *)q; q += al((size_t)(S + 1) * 8);
    double *d_p = (double *)q; q += al(nWS * 8);
    double *d_q = (double *)q; q += al(nWS * 8);
    int *d_arg = (int *)q; q += al((size_t)W * 4);
    unsigned char *d_sig = (unsigned char *)q;
    SP_LAUNCH(ctx, "k6_totals", k6_totals, dim3(1), dim3(256), 0, (const long long *)d_counts, (long long)W, S, d_total);
    SP_LAUNCH(ctx, "k6_enrich", k6_enrich, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, (const long long *)d_counts,
              (const long long *)d_total, (long long)W, S, max_pval, min_ratio, d_p, d_arg, d_sig, d_q);
    SP_HIP(ctx, hipMemcpyAsync(pvals, d_p, nWS * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(ratios, d_q, nWS * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(argmin, d_arg, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(sig, d_sig, (size_t)W, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

extern "C" int sp_enrich(sp_ctx *ctx, const int64_t *counts, int64_t W, int S, double max_pval,
                         double min_ratio, double *pvals, int32_t *argmin, uint8_t *sig, double *ratios) {
    if (!ctx || W < 0 || (W > 0 && (!counts || !pvals || !argmin || !sig || !ratios)))
        return sp_fail(ctx, SP_EINVAL, "sp_enrich: bad arguments");
    if (W == 0 || S < 2 || S > SP_ENRICH_MAXS) return sp_enrich_dev(ctx, counts, W, S, max_pval, min_ratio, pvals, argmin, sig, ratios);
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nWS = (size_t)W * S;
    int rc = sp_buf_ensure(ctx, ctx->b_wtab, (int64_t)nWS * 8 + 64);
    if (rc) return rc;
    SP_HIP(ctx, hipMemcpyAsync(ctx->b_wtab.p, counts, nWS * 8, hipMemcpyHostToDevice, ctx->stream));
    return sp_enrich_dev(ctx, ctx->b_wtab.p, W, S, max_pval, min_ratio, pvals, argmin, sig, ratios);
}
