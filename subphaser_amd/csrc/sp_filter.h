// sp_filter.h -- the per-k-mer decision of the differential filter, shared by the dense (sp_filter.hip)
// and the sparse (sp_sparse.hip) engines.
#pragma once
#include "sp_device.h"

#define F_MAXU 8

// The per-k-mer decision of _filter_kmer (Jellyfish.py:611-648), shared by the dense (k3_eval) and
// the sparse (k > 15) engines.  cnt[c * stride] = thresholded count of chromosome c.
struct sp_fsets {
    int n_sets, baseline;
    int n_multi;   // sets with more than one unit (the reference's `_all`)
    const int32_t *set_off, *unit_off, *unit_chrom;
    const double *unit_den, *unit_inv;   // per-unit denominators and their reciprocals
    double min_fold, min_freq, max_freq, ratio;
};

template <typename CNT>
__device__ __forceinline__ void sp_filter_decide(CNT &&cnt, unsigned long long tot,
                                                 const sp_fsets &F, bool &is_row, bool &is_hist) {
    is_row = is_hist = false;
    int include = 0, all = 0;
    for (int s = 0; s < F.n_sets; s++) {
        const int u0 = F.set_off[s], nu = F.set_off[s + 1] - u0;
        if (nu == 1) continue;  // singleton ignored (Jellyfish.py:621-622)
        // even if every set still to come passed, include / _all would stay below `ratio` (:642-644): the
        // quotient is monotone in its numerator, so the k-mer is rejected exactly as the full loop would
        if ((double)(include + (F.n_multi - all)) / (double)F.n_multi < F.ratio) return;
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
                        num += cnt(F.unit_chrom[j]);
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
                    num += cnt(F.unit_chrom[j]);
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
                        num += cnt(F.unit_chrom[j]);
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

