/*
 * sp_oracle.c -- CPU restatement ("oracle") of SubPhaser's k-mer hot path.
 * TEST INFRASTRUCTURE ONLY (see sp_oracle.h).  Plain C11 + OpenMP.
 *
 * Reference citations are into /root/reference/subphaser/.
 */
#define _GNU_SOURCE
#include "sp_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#include <sys/mman.h>
#endif

static __thread char g_err[512];
const char *spo_last_error(void) { return g_err; }
int spo_version(void) { return 1; }

/* ------------------------------------------------------------------ D1 --- */
/* Base coding: jellyfish's mer_dna codes A/a=0 C/c=1 G/g=2 T/t=3, anything
 * else breaks the k-mer window (the reference's call
 * site is Jellyfish.py:697-699, `jellyfish count -m K --canonical`). */
static int8_t g_code[256];
static int g_code_init = 0;
static double g_t0;
static int timing_on(void) { static int v = -1; if (v < 0) v = getenv("SPO_TIMING") != NULL; return v; }
#define TICK(name) do { if (timing_on()) { double t__ = omp_get_wtime(); fprintf(stderr, "[spo] %-28s %.3f s\n", name, t__ - g_t0); g_t0 = t__; } } while (0)
/* large arrays: 2-MiB aligned + MADV_HUGEPAGE, so that 256 threads first-touching them do not serialise on
 * 4-KiB page faults */
static void *big_alloc(size_t bytes) {   /* transparent huge pages unless SPO_NO_THP is set */
    size_t n = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (n == 0) n = (size_t)2 << 20;
    void *p = NULL;
    if (posix_memalign(&p, (size_t)2 << 20, n)) return NULL;
    if (!getenv("SPO_NO_THP")) madvise(p, n, MADV_HUGEPAGE);
    return p;
}
static void init_code(void) {
    if (g_code_init) return;
    memset(g_code, -1, sizeof g_code);
    g_code['A'] = g_code['a'] = 0;
    g_code['C'] = g_code['c'] = 1;
    g_code['G'] = g_code['g'] = 2;
    g_code['T'] = g_code['t'] = 3;
    g_code_init = 1;
}

struct spo_counts {
    int64_t n;       /* distinct canonical keys with count >= 1 */
    uint64_t *keys;  /* ascending */
    uint32_t *counts;
};

static void radix_sort_u64(uint64_t *a, int64_t n, int bits) {
    uint64_t *tmp = (uint64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(uint64_t));
    uint64_t *src = a, *dst = tmp;
    for (int shift = 0; shift < bits; shift += 8) {
        int64_t hist[257];
        memset(hist, 0, sizeof hist);
        for (int64_t i = 0; i < n; i++) hist[((src[i] >> shift) & 0xff) + 1]++;
        for (int i = 0; i < 256; i++) hist[i + 1] += hist[i];
        for (int64_t i = 0; i < n; i++) dst[hist[(src[i] >> shift) & 0xff]++] = src[i];
        uint64_t *t = src;
        src = dst;
        dst = t;
    }
    if (src != a) memcpy(a, src, (size_t)n * sizeof(uint64_t));
    free(tmp);
}

/* Visit every valid window whose LAST base index lies in [lo, hi). */
#define SCAN_WINDOWS(ascii, len, k, lo, hi, BODY)                                        \
    do {                                                                                 \
        const uint64_t mask__ = (k) == 32 ? ~0ULL : ((1ULL << (2 * (k))) - 1);           \
        const int rcsh__ = 2 * ((k)-1);                                                  \
        uint64_t fwd = 0, rc = 0;                                                        \
        int64_t run = 0;                                                                 \
        int64_t st__ = (lo) - ((k)-1);                                                   \
        if (st__ < 0) st__ = 0;                                                          \
        for (int64_t i = st__; i < (hi); i++) {                                          \
            int c__ = g_code[(ascii)[i]];                                                \
            if (c__ < 0) {                                                               \
                run = 0;                                                                 \
                continue;                                                                \
            }                                                                            \
            fwd = ((fwd << 2) | (uint64_t)c__) & mask__;                                 \
            rc = (rc >> 2) | ((uint64_t)(3 - c__) << rcsh__);                            \
            if (++run >= (k) && i >= (lo)) {                                             \
                uint64_t key = fwd < rc ? fwd : rc;                                      \
                int64_t start = i - (k) + 1;                                             \
                (void)start;                                                             \
                BODY                                                                     \
            }                                                                            \
        }                                                                                \
    } while (0)

spo_counts *spo_count(const uint8_t *ascii, int64_t len, int k, int nthreads) {
    init_code();
    if (k < 1 || k > 32) {
        snprintf(g_err, sizeof g_err, "k=%d unsupported (1..32)", k);
        return NULL;
    }
    if (nthreads < 1) nthreads = 1;
    spo_counts *res = (spo_counts *)calloc(1, sizeof *res);
    int dense = 0;
    if (k <= 15) {
        int64_t slots = 1LL << (2 * k);
        if (len >= slots / 16) dense = 1;
    }
    if (dense) {
        /* Same result as one direct-addressed table indexed by the canonical value, built the way a tuned
         * CPU counter builds it: every thread scatters the keys of its share of the sequence into
         * key-range buckets (two scans: count, place -- no atomics), then each bucket is counted in a
         * table small enough to stay in the core's cache and its non-zero slots go out in ascending order. */
        int pb = 2 * k - 8;
        if (pb > 12) pb = 12;
        if (pb < 0) pb = 0;
        const int sh = 2 * k - pb;
        const int64_t nbkt = 1LL << pb, bslots = 1LL << sh;
        const int64_t nblk = nthreads;
        int64_t per = (len + nblk - 1) / nblk;
        if (per < 1) per = 1;
        g_t0 = omp_get_wtime();
        int64_t *off = (int64_t *)calloc((size_t)(nblk * nbkt + 1), sizeof(int64_t));
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
        for (int64_t b = 0; b < nblk; b++) {
            int64_t lo = b * per, hi = lo + per;
            if (hi > len) hi = len;
            if (lo >= hi) continue;
            int64_t *mine = (int64_t *)calloc((size_t)nbkt, sizeof(int64_t));
            SCAN_WINDOWS(ascii, len, k, lo, hi, { mine[key >> sh]++; });
            for (int64_t q = 0; q < nbkt; q++) off[q * nblk + b + 1] = mine[q];
            free(mine);
        }
        TICK("count: scan 1 (bucket sizes)");
        for (int64_t i = 0; i < nblk * nbkt; i++) off[i + 1] += off[i];
        const int64_t total = off[nblk * nbkt];
        uint32_t *part = (uint32_t *)big_alloc((size_t)(total ? total : 1) * sizeof(uint32_t));
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
        for (int64_t b = 0; b < nblk; b++) {
            int64_t lo = b * per, hi = lo + per;
            if (hi > len) hi = len;
            if (lo >= hi) continue;
            /* software write combining: 16 keys (one cache line) per bucket are staged before they go out */
            int64_t *cur = (int64_t *)malloc((size_t)nbkt * sizeof(int64_t));
            uint32_t *stage = (uint32_t *)malloc((size_t)nbkt * 16 * sizeof(uint32_t));
            uint8_t *fill = (uint8_t *)calloc((size_t)nbkt, 1);
            for (int64_t q = 0; q < nbkt; q++) cur[q] = off[q * nblk + b];
            SCAN_WINDOWS(ascii, len, k, lo, hi, {
                const int64_t q = (int64_t)(key >> sh);
                stage[q * 16 + fill[q]] = (uint32_t)(key & (uint64_t)(bslots - 1));
                if (++fill[q] == 16) {
                    memcpy(part + cur[q], stage + q * 16, 64);
                    cur[q] += 16;
                    fill[q] = 0;
                }
            });
            for (int64_t q = 0; q < nbkt; q++)
                if (fill[q]) memcpy(part + cur[q], stage + q * 16, (size_t)fill[q] * 4);
            free(cur);
            free(stage);
            free(fill);
        }
        TICK("count: scan 2 (scatter)");
        int64_t *nd = (int64_t *)calloc((size_t)nbkt + 1, sizeof(int64_t));
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) {
                for (int64_t q = 0; q < nbkt; q++) nd[q + 1] += nd[q];
                res->n = nd[nbkt];
                res->keys = (uint64_t *)big_alloc((size_t)(res->n ? res->n : 1) * sizeof(uint64_t));
                res->counts = (uint32_t *)big_alloc((size_t)(res->n ? res->n : 1) * sizeof(uint32_t));
            }
#pragma omp parallel num_threads(nthreads)
            {
                uint32_t *tab = (uint32_t *)malloc((size_t)bslots * sizeof(uint32_t));
#pragma omp for schedule(dynamic, 1)
                for (int64_t q = 0; q < nbkt; q++) {
                    const int64_t lo = off[q * nblk], hi = off[(q + 1) * nblk];
                    if (lo == hi) continue;
                    memset(tab, 0, (size_t)bslots * sizeof(uint32_t));
                    for (int64_t i = lo; i < hi; i++) tab[part[i]]++;
                    if (pass == 0) {
                        int64_t c = 0;
                        for (int64_t i = 0; i < bslots; i++) c += tab[i] != 0;
                        nd[q + 1] = c;
                    } else {
                        int64_t o = nd[q];
                        for (int64_t i = 0; i < bslots; i++)
                            if (tab[i]) {
                                res->keys[o] = ((uint64_t)q << sh) | (uint64_t)i;
                                res->counts[o] = tab[i];
                                o++;
                            }
                    }
                }
                free(tab);
            }
            TICK(pass ? "count: buckets, emit" : "count: buckets, distinct");
        }
        free(nd);
        free(part);
        free(off);
        return res;
    }
    /* sort path: all canonical keys, radix sort, run-length encode */
    uint64_t *all = (uint64_t *)malloc((size_t)(len > 0 ? len : 1) * sizeof(uint64_t));
    int64_t m = 0;
    SCAN_WINDOWS(ascii, len, k, 0, len, { all[m++] = key; });
    radix_sort_u64(all, m, 2 * k);
    int64_t nd = 0;
    for (int64_t i = 0; i < m; i++)
        if (i == 0 || all[i] != all[i - 1]) nd++;
    res->n = nd;
    res->keys = (uint64_t *)malloc((size_t)(nd ? nd : 1) * sizeof(uint64_t));
    res->counts = (uint32_t *)malloc((size_t)(nd ? nd : 1) * sizeof(uint32_t));
    int64_t o = -1;
    for (int64_t i = 0; i < m; i++) {
        if (i == 0 || all[i] != all[i - 1]) {
            o++;
            res->keys[o] = all[i];
            res->counts[o] = 0;
        }
        res->counts[o]++;
    }
    free(all);
    return res;
}

int64_t spo_counts_n(const spo_counts *c, uint32_t lower) {
    int64_t n = 0;
#pragma omp parallel for reduction(+ : n) if (c->n > (1 << 20))
    for (int64_t i = 0; i < c->n; i++) n += c->counts[i] >= lower;
    return n;
}
/* `jellyfish dump -c -L lower_count` keeps counts >= lower_count (Jellyfish.py:699) */
int64_t spo_counts_fetch(const spo_counts *c, uint32_t lower, uint64_t *keys, uint32_t *counts) {
    enum { NB = 512 };
    int64_t start[NB + 1];
    const int64_t per = (c->n + NB - 1) / NB;
    start[0] = 0;
#pragma omp parallel for if (c->n > (1 << 20))
    for (int b = 0; b < NB; b++) {
        int64_t lo = b * per, hi = lo + per, m = 0;
        if (hi > c->n) hi = c->n;
        for (int64_t i = lo; i < hi; i++) m += c->counts[i] >= lower;
        start[b + 1] = m;
    }
    for (int b = 0; b < NB; b++) start[b + 1] += start[b];
#pragma omp parallel for if (c->n > (1 << 20))
    for (int b = 0; b < NB; b++) {
        int64_t lo = b * per, hi = lo + per, n = start[b];
        if (hi > c->n) hi = c->n;
        for (int64_t i = lo; i < hi; i++)
            if (c->counts[i] >= lower) {
                keys[n] = c->keys[i];
                counts[n] = c->counts[i];
                n++;
            }
    }
    return start[NB];
}
void spo_counts_free(spo_counts *c) {
    if (!c) return;
    free(c->keys);
    free(c->counts);
    free(c);
}

/* ------------------------------------------------------------------ D2 --- */
struct spo_filtered {
    int C;
    int64_t n_union, n_rows, n_hist;
    int64_t *lengths;
    uint64_t *keys;
    uint32_t *counts;
    double *freqs;
    uint64_t *tot;
    uint64_t *hist;
    int64_t cap_rows, cap_hist;
};

static int cmp_desc(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x < y) - (x > y);
}

/* JellyfishDumps.to_matrix (Jellyfish.py:439-460) + JellyfishDumps.filter
 * (:462-512) + _filter_kmer (:611-648), outfig always set (__main__.py:421). */
spo_filtered *spo_filter(int C, const int64_t *dump_off, const uint64_t *keys_all,
                         const uint32_t *counts_all, int n_sets, const int32_t *set_off,
                         const int32_t *unit_off, const int32_t *unit_chrom, double min_fold,
                         int baseline, double min_freq, double max_freq, double ratio,
                         const int64_t *lengths_override) {
    /* Jellyfish.py:474-475 */
    if (min_freq > max_freq) {
        snprintf(g_err, sizeof g_err, "`min_freq` (%g) should be lower than `max_freq` (%g)",
                 min_freq, max_freq);
        return NULL;
    }
    /* Jellyfish.py:477-483 */
    int nsingle = 0;
    for (int s = 0; s < n_sets; s++) nsingle += (set_off[s + 1] - set_off[s]) == 1;
    if (nsingle == n_sets) {
        snprintf(g_err, sizeof g_err, "All singletons are not allowed");
        return NULL;
    }
    spo_filtered *f = (spo_filtered *)calloc(1, sizeof *f);
    f->C = C;
    f->lengths = (int64_t *)calloc((size_t)C, sizeof(int64_t));
    /* lengths[i] = tot of dump i (Jellyfish.py:97,449) */
    for (int c = 0; c < C; c++)
        for (int64_t i = dump_off[c]; i < dump_off[c + 1]; i++) f->lengths[c] += counts_all[i];
    /* a slot-range shard of the matrix is filtered against the GLOBAL lengths (multi-GPU tests) */
    if (lengths_override)
        for (int c = 0; c < C; c++) f->lengths[c] = lengths_override[c];
    /* Jellyfish.py:487-489 */
    for (int c = 0; c < C; c++)
        if (f->lengths[c] == 0) {
            snprintf(g_err, sizeof g_err, "Chromosomes `[%d]` have only 0 kmers", c);
            free(f->lengths);
            free(f);
            return NULL;
        }
    int64_t *head = (int64_t *)malloc((size_t)C * sizeof(int64_t));
    for (int c = 0; c < C; c++) head[c] = dump_off[c];
    uint32_t *row = (uint32_t *)malloc((size_t)C * sizeof(uint32_t));
    int maxu = 0;
    for (int s = 0; s < n_sets; s++)
        if (set_off[s + 1] - set_off[s] > maxu) maxu = set_off[s + 1] - set_off[s];
    double *freqs = (double *)malloc((size_t)(maxu ? maxu : 1) * sizeof(double));
    int bad_index = 0;
    for (;;) {
        /* next key of the outer join */
        uint64_t key = 0;
        int any = 0;
        for (int c = 0; c < C; c++)
            if (head[c] < dump_off[c + 1]) {
                uint64_t kk = keys_all[head[c]];
                if (!any || kk < key) key = kk;
                any = 1;
            }
        if (!any) break;
        uint64_t tot = 0;
        for (int c = 0; c < C; c++) {
            row[c] = 0;
            if (head[c] < dump_off[c + 1] && keys_all[head[c]] == key) {
                row[c] = counts_all[head[c]];
                head[c]++;
            }
            tot += row[c];
        }
        f->n_union++;
        /* _filter_kmer */
        int include = 0, all = 0;
        for (int s = 0; s < n_sets; s++) {
            int nu = set_off[s + 1] - set_off[s];
            if (nu == 1) continue;
            all++;
            for (int u = 0; u < nu; u++) {
                int uu = set_off[s] + u;
                int64_t num = 0, den = 0;
                for (int j = unit_off[uu]; j < unit_off[uu + 1]; j++) {
                    num += row[unit_chrom[j]];
                    den += f->lengths[unit_chrom[j]];
                }
                freqs[u] = (double)num / (double)den; /* :630,:634 (by_count never set) */
            }
            qsort(freqs, (size_t)nu, sizeof(double), cmp_desc); /* :637 */
            int bi = baseline < 0 ? nu + baseline : baseline;   /* python indexing :639 */
            if (bi < 0 || bi >= nu) {
                bad_index = 1;
                break;
            }
            double hi = freqs[0], lo = freqs[bi];
            if (1.0 * hi / (lo + 1e-20) >= min_fold) include++; /* :640-641 */
        }
        if (bad_index) break;
        double r = 1.0 * (double)include / (double)all; /* :642 */
        if (r < ratio) continue;                         /* :643-644 */
        if ((double)tot < min_freq || (double)tot > max_freq) { /* :645-646 */
            if (f->n_hist == f->cap_hist) {
                f->cap_hist = f->cap_hist ? f->cap_hist * 2 : 1024;
                f->hist = (uint64_t *)realloc(f->hist, (size_t)f->cap_hist * sizeof(uint64_t));
            }
            f->hist[f->n_hist++] = tot;
            continue;
        }
        if (f->n_rows == f->cap_rows) {
            f->cap_rows = f->cap_rows ? f->cap_rows * 2 : 1024;
            f->keys = (uint64_t *)realloc(f->keys, (size_t)f->cap_rows * sizeof(uint64_t));
            f->tot = (uint64_t *)realloc(f->tot, (size_t)f->cap_rows * sizeof(uint64_t));
            f->counts =
                (uint32_t *)realloc(f->counts, (size_t)f->cap_rows * (size_t)C * sizeof(uint32_t));
            f->freqs = (double *)realloc(f->freqs, (size_t)f->cap_rows * (size_t)C * sizeof(double));
        }
        if (f->n_hist == f->cap_hist) {
            f->cap_hist = f->cap_hist ? f->cap_hist * 2 : 1024;
            f->hist = (uint64_t *)realloc(f->hist, (size_t)f->cap_hist * sizeof(uint64_t));
        }
        f->hist[f->n_hist++] = tot; /* survivors feed tot_freqs too (:501-502) */
        f->keys[f->n_rows] = key;
        f->tot[f->n_rows] = tot;
        for (int c = 0; c < C; c++) {
            f->counts[f->n_rows * C + c] = row[c];
            f->freqs[f->n_rows * C + c] = (double)row[c] / (double)f->lengths[c]; /* :647 */
        }
        f->n_rows++;
    }
    free(head);
    free(row);
    free(freqs);
    if (bad_index) {
        snprintf(g_err, sizeof g_err, "list index out of range (baseline=%d)", baseline);
        spo_filtered_free(f);
        return NULL;
    }
    return f;
}
int64_t spo_filtered_n_union(const spo_filtered *f) { return f->n_union; }
int64_t spo_filtered_n_rows(const spo_filtered *f) { return f->n_rows; }
int64_t spo_filtered_n_hist(const spo_filtered *f) { return f->n_hist; }
void spo_filtered_lengths(const spo_filtered *f, int64_t *lengths) {
    memcpy(lengths, f->lengths, (size_t)f->C * sizeof(int64_t));
}
void spo_filtered_fetch(const spo_filtered *f, uint64_t *keys, uint32_t *counts, double *freqs,
                        uint64_t *tot) {
    size_t M = (size_t)f->n_rows, C = (size_t)f->C;
    if (keys) memcpy(keys, f->keys, M * sizeof(uint64_t));
    if (counts) memcpy(counts, f->counts, M * C * sizeof(uint32_t));
    if (freqs) memcpy(freqs, f->freqs, M * C * sizeof(double));
    if (tot) memcpy(tot, f->tot, M * sizeof(uint64_t));
}
void spo_filtered_hist(const spo_filtered *f, uint64_t *tot) {
    memcpy(tot, f->hist, (size_t)f->n_hist * sizeof(uint64_t));
}
void spo_filtered_free(spo_filtered *f) {
    if (!f) return;
    free(f->lengths);
    free(f->keys);
    free(f->counts);
    free(f->freqs);
    free(f->tot);
    free(f->hist);
    free(f);
}

/* ------------------------------------------------------------------ D3 --- */
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

/* Seqs.map_kmer3 (Seqs.py:74-119) + chunk_chromfiles (:121-139) + map_kmer_each4
 * (:209-237).  d_kmers holds each significant k-mer and its reverse complement
 * (Cluster.py:174-175), i.e. lookup is by canonical key.  The reference
 * upper-cases before lookup (:129,:147) so case never matters; any other
 * character can never match. */
int64_t spo_map_bins(const uint8_t *ascii, int64_t len, int k, const uint64_t *lab_keys,
                     const uint8_t *lab_sg, int64_t n_lab, int S, int64_t bin_size,
                     int64_t chunk_size, int32_t *slot_counts, int64_t nslots, uint8_t *hit,
                     int nthreads) {
    init_code();
    if (nthreads < 1) nthreads = 1;
    int64_t cap = 16;
    while (cap < 2 * n_lab) cap <<= 1;
    uint64_t *hk = (uint64_t *)malloc((size_t)cap * sizeof(uint64_t));
    int64_t *hv = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    memset(hk, 0xff, (size_t)cap * sizeof(uint64_t));
    for (int64_t i = 0; i < n_lab; i++) {
        uint64_t h = mix64(lab_keys[i]) & (uint64_t)(cap - 1);
        while (hk[h] != ~0ULL) h = (h + 1) & (uint64_t)(cap - 1);
        hk[h] = lab_keys[i];
        hv[h] = i;
    }
    /* cache-sized prefilter: one bit per hashed key, so that most windows never touch the big table */
    const uint64_t pf_mask = ((uint64_t)1 << 27) - 1;
    uint64_t *pf = (uint64_t *)calloc((size_t)1 << 21, sizeof(uint64_t));
    for (int64_t i = 0; i < n_lab; i++) {
        uint64_t h = (mix64(lab_keys[i]) >> 20) & pf_mask;
        pf[h >> 6] |= (uint64_t)1 << (h & 63);
    }
    int64_t mapped = 0;
    int64_t nblk = nthreads * 8;
    int64_t per = (len + nblk - 1) / nblk;
    if (per < 1) per = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) reduction(+ : mapped)
    for (int64_t b = 0; b < nblk; b++) {
        int64_t lo = b * per, hi = lo + per;
        if (hi > len) hi = len;
        if (lo >= hi) continue;
        /* bins this block can touch: starts lo-(k-1) .. hi-k; counted privately, merged once */
        int64_t s0 = lo - (k - 1);
        if (s0 < 0) s0 = 0;
        const int64_t slot0 = s0 / bin_size + (chunk_size > 0 ? s0 / chunk_size : 0);
        const int64_t slot1 = hi / bin_size + (chunk_size > 0 ? (hi + k) / chunk_size : 0) + 2;
        int32_t *mine = (int32_t *)calloc((size_t)((slot1 - slot0 + 1) * S), sizeof(int32_t));
        SCAN_WINDOWS(ascii, len, k, lo, hi, {
            const uint64_t hm = mix64(key);
            const uint64_t hb = (hm >> 20) & pf_mask;
            if (!((pf[hb >> 6] >> (hb & 63)) & 1)) continue;
            uint64_t h = hm & (uint64_t)(cap - 1);
            while (hk[h] != ~0ULL && hk[h] != key) h = (h + 1) & (uint64_t)(cap - 1);
            if (hk[h] == key) {
                int64_t li = hv[h];
                int64_t chunk = 0;
                if (chunk_size > 0 && start >= chunk_size - (k - 1))
                    chunk = (start + (k - 1)) / chunk_size;
                int64_t slot = start / bin_size + chunk;
                if (slot < nslots) mine[(slot - slot0) * S + lab_sg[li]]++;
                if (hit && !hit[li]) hit[li] = 1;
                mapped++;
            }
        });
        for (int64_t q = slot0; q <= slot1 && q < nslots; q++)
            for (int j = 0; j < S; j++)
                if (mine[(q - slot0) * S + j])
                    __atomic_fetch_add(&slot_counts[q * S + j], mine[(q - slot0) * S + j], __ATOMIC_RELAXED);
        free(mine);
    }
    free(pf);
    free(hk);
    free(hv);
    return mapped;
}

/* ------------------------------------------------------------------ D5 --- */
static long double lchoosel(long double n, long double k) {
    return lgammal(n + 1.0L) - lgammal(k + 1.0L) - lgammal(n - k + 1.0L);
}

/* Definition-level right tail P[X >= a] of the hypergeometric law the
 * reference obtains from fisher.pvalue(x11,x12,x21,x22).right_tail
 * (Stats.py:26), in 80-bit long double. */
double spo_hypergeom_right_tail(int64_t a, int64_t b, int64_t c, int64_t d) {
    long double N = (long double)a + b + c + d, K = (long double)a + b, n = (long double)a + c;
    int64_t lo = (a + c) - (c + d);
    if (lo < 0) lo = 0;
    int64_t hi = (a + b) < (a + c) ? (a + b) : (a + c);
    if (a <= lo) return 1.0;
    if (a > hi) return 0.0;
    long double logden = lchoosel(N, n);
    long double mode = floorl((n + 1.0L) * (K + 1.0L) / (N + 2.0L));
    if ((long double)a > mode) {
        long double x = (long double)a;
        long double term = expl(lchoosel(K, x) + lchoosel(N - K, n - x) - logden);
        long double s = term;
        while (x < (long double)hi) {
            term *= (K - x) * (n - x) / ((x + 1.0L) * (N - K - n + x + 1.0L));
            x += 1.0L;
            s += term;
            if (term < s * 1e-25L) break;
        }
        return (double)s;
    } else {
        long double x = (long double)a - 1.0L;
        long double term = expl(lchoosel(K, x) + lchoosel(N - K, n - x) - logden);
        long double s = term;
        while (x > (long double)lo) {
            term *= x * (N - K - n + x) / ((K - x + 1.0L) * (n - x + 1.0L));
            x -= 1.0L;
            s += term;
            if (term < s * 1e-25L) break;
        }
        return (double)(1.0L - s);
    }
}

#define SPO_MAX_INT (2147483647LL / 10) /* Stats.py:9 */

/* fisher_test margins (Stats.py:17-25), including the x22 quirk:
 * x22 = sum_total - x21 - x12 uses the UNclamped x21 and is N - x12 - x21. */
void spo_fisher_cells(const int64_t *each, const int64_t *total, int S, int j, int64_t cells[4]) {
    int64_t sum_each = 0, sum_total = 0;
    for (int i = 0; i < S; i++) {
        sum_each += each[i];
        sum_total += total[i];
    }
    int64_t x11 = each[j];
    int64_t x12 = sum_each - x11;
    int64_t x21 = total[j] - x11;
    int64_t x22 = sum_total - x21 - x12;
    if (x21 > SPO_MAX_INT) x21 = SPO_MAX_INT;
    if (x22 > SPO_MAX_INT) x22 = SPO_MAX_INT;
    cells[0] = x11;
    cells[1] = x12;
    cells[2] = x21;
    cells[3] = x22;
}

/* numpy's float64 add.reduce order for a short contiguous vector */
static double np_sum(const double *a, int n) {
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

/* Stats.enrich (:140-148) + _enrich (:150-168) + Pvalues.get_enriched (:181-192) */
void spo_enrich(const int64_t *counts, int64_t W, int S, double max_pval, double min_ratio,
                double *pvals, int32_t *argmin, uint8_t *sig, double *ratios) {
    int64_t *total = (int64_t *)calloc((size_t)S, sizeof(int64_t));
    for (int64_t w = 0; w < W; w++)
        for (int j = 0; j < S; j++) total[j] += counts[w * S + j];
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t w = 0; w < W; w++) {
        const int64_t *row = counts + w * S;
        double *p = pvals + w * S;
        for (int j = 0; j < S; j++) {
            int64_t x[4];
            spo_fisher_cells(row, total, S, j, x);
            p[j] = spo_hypergeom_right_tail(x[0], x[1], x[2], x[3]);
        }
        /* stable sort by p: first two */
        int m = 0;
        for (int j = 1; j < S; j++)
            if (p[j] < p[m]) m = j;
        int s2 = -1;
        for (int j = 0; j < S; j++) {
            if (j == m) continue;
            if (s2 < 0 || p[j] < p[s2]) s2 = j;
        }
        int sg = 1;
        if (p[m] > max_pval) sg = 0;
        if (p[m] == 0) {
        } else if (p[s2] / p[m] < max_pval / p[s2] * 1.0)
            sg = 0;
        double *q = ratios + w * S;
        for (int j = 0; j < S; j++) q[j] = (double)row[j] / (double)total[j];
        double qs = np_sum(q, S);
        for (int j = 0; j < S; j++) q[j] = q[j] / qs;
        if (q[m] < min_ratio) sg = 0;
        argmin[w] = m;
        sig[w] = (uint8_t)sg;
    }
    free(total);
}
