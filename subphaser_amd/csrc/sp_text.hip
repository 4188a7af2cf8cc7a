// sp_text.hip -- host-side writers of the two big text outputs (no kernels).
//
// `.kmer.mat` (Jellyfish.py:515-520: k-mer + str(count/length) per chromosome) and `.sig.kmer-subgenome.tsv`
// (Cluster.py:165-176: k-mer, subgenome, p-value, comma-joined group means) are 0.55 GB and 0.19 GB of text for the
// wheat-like genome: 150 M floats printed the way Python prints them.  The host mirror formatted them with fork()ed
// worker pools -- and a GPU process with live forked children pays for it in every later pageable host<->device copy
// (the driver re-registers user pages against a copy-on-write address space: sp_labels_set went from 3 ms to 1 s).
// Here the rows are formatted by threads of this process into per-chunk buffers and written in order.
//
// repr(float): CPython's float_repr_style 'short' = the shortest digit string that round-trips (David Gay, mode 0;
// std::to_chars produces the same digits), laid out by format_float_short('r'): exponent form when the decimal point
// position decpt <= -4 or > 16, at least two exponent digits, ".0" appended to integral fixed values, "nan", "inf".
#include "sp_common.h"

#include <atomic>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>

// returns the number of characters written (out needs >= 40 bytes)
static int sp_py_repr(double x, char *out) {
    if (std::isnan(x)) {
        memcpy(out, "nan", 3);
        return 3;
    }
    char *o = out;
    if (std::signbit(x)) {
        *o++ = '-';
        x = -x;
    }
    if (std::isinf(x)) {
        memcpy(o, "inf", 3);
        return (int)(o - out) + 3;
    }
    if (x == 0.0) {
        memcpy(o, "0.0", 3);
        return (int)(o - out) + 3;
    }
    char sci[48];
    auto r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
    char digits[24];
    int nd = 0;
    const char *p = sci;
    for (; p < r.ptr && *p != 'e'; p++)
        if (*p != '.') digits[nd++] = *p;
    int e10 = 0;
    {
        const char *q = p + 1;
        bool neg = false;
        if (*q == '+' || *q == '-') neg = *q++ == '-';
        for (; q < r.ptr; q++) e10 = e10 * 10 + (*q - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;     // value = 0.d1d2...dn x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *o++ = digits[0];
        if (nd > 1) {
            *o++ = '.';
            memcpy(o, digits + 1, (size_t)(nd - 1));
            o += nd - 1;
        }
        *o++ = 'e';
        int e = decpt - 1;
        *o++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int ne = 0;
        do {
            eb[ne++] = (char)('0' + e % 10);
            e /= 10;
        } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *o++ = eb[--ne];
    } else if (decpt <= 0) {
        *o++ = '0';
        *o++ = '.';
        for (int i = 0; i < -decpt; i++) *o++ = '0';
        memcpy(o, digits, (size_t)nd);
        o += nd;
    } else if (decpt >= nd) {
        memcpy(o, digits, (size_t)nd);
        o += nd;
        for (int i = nd; i < decpt; i++) *o++ = '0';
        *o++ = '.';
        *o++ = '0';
    } else {
        memcpy(o, digits, (size_t)decpt);
        o += decpt;
        *o++ = '.';
        memcpy(o, digits + decpt, (size_t)(nd - decpt));
        o += nd - decpt;
    }
    return (int)(o - out);
}

// test hook: repr of every x[i], back to back; off[i] .. off[i + 1] (out: >= 40 bytes per value)
extern "C" int sp_text_repr(const double *x, int64_t n, char *out, int64_t *off) {
    if (n < 0 || (n > 0 && (!x || !out)) || !off) return SP_EINVAL;
    int64_t w = 0;
    for (int64_t i = 0; i < n; i++) {
        off[i] = w;
        w += sp_py_repr(x[i], out + w);
    }
    off[n] = w;
    return SP_OK;
}

static inline void sp_text_kmer(uint64_t key, int k, char *o) {
    for (int j = 0; j < k; j++) o[j] = "ACGT"[(key >> (2 * (k - 1 - j))) & 3ULL];
}

// rows [lo, hi) -> buf; chunks are formatted by a pool of threads, waves of chunks are written to fd in order.
// A worker that fails (std::bad_alloc in a buffer, anything else thrown by the formatter) raises a shared flag and
// stops; every thread is joined before the wave's verdict is read, so nothing escapes a std::thread (that would be
// std::terminate for the whole GPU process) and no joinable thread is ever destroyed.  write() is retried on EINTR;
// any other failure is SP_EIO with errno in sp_last_error (ENOSPC on a 0.5-GB `.kmer.mat` must be readable).
template <typename F>
static int sp_text_rows(int64_t M, int threads, int fd, int64_t *bytes, F &&format_rows) {
    const int64_t CH = 8192;
    const int64_t n_ch = (M + CH - 1) / CH;
    const int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    const int64_t wave = (int64_t)T * 4;
    std::vector<std::string> bufs;
    try {
        bufs.resize((size_t)wave);
    } catch (const std::bad_alloc &) {
        return SP_ENOMEM;
    }
    int64_t total = 0;
    for (int64_t c0 = 0; c0 < n_ch; c0 += wave) {
        const int64_t c1 = c0 + wave < n_ch ? c0 + wave : n_ch;
        std::atomic<int64_t> next(c0);
        std::atomic<int> failed(0);      // 0 ok, 1 out of memory, 2 other
        auto run = [&]() noexcept {
            try {
                for (;;) {
                    const int64_t c = next.fetch_add(1, std::memory_order_relaxed);
                    if (c >= c1 || failed.load(std::memory_order_relaxed)) return;
                    std::string &b = bufs[(size_t)(c - c0)];
                    b.clear();
                    format_rows(c * CH, (c + 1) * CH < M ? (c + 1) * CH : M, b);
                }
            } catch (const std::bad_alloc &) {
                failed.store(1);
            } catch (...) {
                failed.store(2);
            }
        };
        std::vector<std::thread> pool;
        const int nt = (int)((c1 - c0) < T ? (c1 - c0) : T);
        try {
            pool.reserve((size_t)nt);
            for (int t = 1; t < nt; t++) pool.emplace_back(run);   // may throw std::system_error: the threads started so far still run
        } catch (...) {
        }
        run();
        for (auto &th : pool) th.join();
        if (failed.load()) {
            g_sp_err = failed.load() == 1 ? "text writer: out of memory while formatting rows" : "text writer: formatter failed";
            return failed.load() == 1 ? SP_ENOMEM : SP_EINVAL;
        }
        for (int64_t c = c0; c < c1; c++) {
            const std::string &b = bufs[(size_t)(c - c0)];
            size_t done = 0;
            while (done < b.size()) {
                const ssize_t w = write(fd, b.data() + done, b.size() - done);
                if (w < 0) {
                    if (errno == EINTR) continue;
                    char msg[160];
                    snprintf(msg, sizeof msg, "text writer: write(fd %d) failed after %lld bytes: %s (errno %d)", fd,
                             (long long)(total + (int64_t)done), strerror(errno), errno);
                    g_sp_err = msg;
                    return SP_EIO;
                }
                done += (size_t)w;
            }
            total += (int64_t)b.size();
        }
    }
    if (bytes) *bytes = total;
    return SP_OK;
}

// Generic TSV rows for the feature-scale outputs (`.custom.enrich` / `.ltr.enrich`, Stats.py:33-73; the feature-mode
// `.bin.count`, Seqs.py:228-244): columns are joined by '\t', rows end in '\n'.
extern "C" int sp_text_table(const sp_text_col *cols, int n_cols, int64_t M, int threads, int fd, int64_t *bytes) {
    if (M < 0 || n_cols < 1 || !cols) return SP_EINVAL;
    for (int c = 0; c < n_cols; c++) {
        const sp_text_col &q = cols[c];
        if (q.kind < SP_COL_STR || q.kind > SP_COL_IVAL || (M > 0 && !q.data)) return SP_EINVAL;
        if ((q.kind == SP_COL_STR || q.kind == SP_COL_NAME || q.kind == SP_COL_IVAL) && !q.off) return SP_EINVAL;
        if (q.kind == SP_COL_IVAL) {
            if (!q.names || q.width < 1) return SP_EINVAL;
            const int64_t *iv = (const int64_t *)q.data;
            for (int64_t i = 0; i < M; i++)
                if (iv[3 * i] < 0 || iv[3 * i] >= q.width) return SP_EINVAL;
        }
        if ((q.kind == SP_COL_I64 || q.kind == SP_COL_F64) && q.width < 1) return SP_EINVAL;
        if (q.kind == SP_COL_NAME) {
            if (!q.names || q.width < 1) return SP_EINVAL;
            const int32_t *ix = (const int32_t *)q.data;
            for (int64_t i = 0; i < M; i++)
                if (ix[i] < 0 || ix[i] >= q.width) return SP_EINVAL;
        }
    }
    return sp_text_rows(M, threads, fd, bytes, [&](int64_t lo, int64_t hi, std::string &b) {
        char tmp[48];
        for (int64_t i = lo; i < hi; i++) {
            for (int c = 0; c < n_cols; c++) {
                const sp_text_col &q = cols[c];
                if (c) b.push_back('\t');
                switch (q.kind) {
                case SP_COL_STR: {
                    const char *s = (const char *)q.data;
                    b.append(s + q.off[i], (size_t)(q.off[i + 1] - q.off[i]));
                    break;
                }
                case SP_COL_NAME: {
                    const int32_t j = ((const int32_t *)q.data)[i];
                    b.append(q.names + q.off[j], (size_t)(q.off[j + 1] - q.off[j]));
                    break;
                }
                case SP_COL_IVAL: {      // name:start-end
                    const int64_t *iv = (const int64_t *)q.data + 3 * i;
                    b.append(q.names + q.off[iv[0]], (size_t)(q.off[iv[0] + 1] - q.off[iv[0]]));
                    b.push_back(':');
                    auto r1 = std::to_chars(tmp, tmp + sizeof tmp, (long long)iv[1]);
                    b.append(tmp, (size_t)(r1.ptr - tmp));
                    b.push_back('-');
                    auto r2 = std::to_chars(tmp, tmp + sizeof tmp, (long long)iv[2]);
                    b.append(tmp, (size_t)(r2.ptr - tmp));
                    break;
                }
                case SP_COL_I64: {
                    const int64_t *row = (const int64_t *)q.data + i * q.width;
                    for (int j = 0; j < q.width; j++) {
                        if (j) b.push_back(q.join);
                        auto r = std::to_chars(tmp, tmp + sizeof tmp, (long long)row[j]);
                        b.append(tmp, (size_t)(r.ptr - tmp));
                    }
                    break;
                }
                default: {
                    const double *row = (const double *)q.data + i * q.width;
                    for (int j = 0; j < q.width; j++) {
                        if (j) b.push_back(q.join);
                        b.append(tmp, (size_t)sp_py_repr(row[j], tmp));
                    }
                }
                }
            }
            b.push_back('\n');
        }
    });
}

// rows of `.kmer.mat` (without the header line): kmer \t repr(f[0]) \t ... \t repr(f[C-1]) \n
extern "C" int sp_text_kmer_matrix(const uint64_t *keys, int k, const double *freqs, int64_t M, int C, int threads,
                                   int fd, int64_t *bytes) {
    if (M < 0 || C < 1 || k < 1 || k > 32 || (M > 0 && (!keys || !freqs))) return SP_EINVAL;
    {
        return sp_text_rows(M, threads, fd, bytes, [&](int64_t lo, int64_t hi, std::string &b) {
            char tmp[48];
            b.reserve((size_t)(hi - lo) * (size_t)(k + 1 + C * 24));
            for (int64_t i = lo; i < hi; i++) {
                sp_text_kmer(keys[i], k, tmp);
                b.append(tmp, (size_t)k);
                const double *row = freqs + i * C;
                for (int c = 0; c < C; c++) {
                    b.push_back('\t');
                    b.append(tmp, (size_t)sp_py_repr(row[c], tmp));
                }
                b.push_back('\n');
            }
        });
    }
}

// rows of `.sig.kmer-subgenome.tsv` (without the header line): kmer \t name[top[i]] \t repr(p[i]) \t repr(m[0]),...
// names: n_names strings, '\0'-separated
extern "C" int sp_text_sig_kmers(const uint64_t *keys, int k, const int32_t *top, const char *names, int n_names,
                                 const double *pvals, const double *means, int G, int64_t M, int threads, int fd,
                                 int64_t *bytes) {
    if (M < 0 || G < 1 || k < 1 || k > 32 || n_names < 1 || !names || (M > 0 && (!keys || !top || !pvals || !means)))
        return SP_EINVAL;
    std::vector<std::string> nm;
    {
        const char *p = names;
        for (int i = 0; i < n_names; i++) {
            nm.emplace_back(p);
            p += nm.back().size() + 1;
        }
    }
    for (int64_t i = 0; i < M; i++)
        if (top[i] < 0 || top[i] >= n_names) return SP_EINVAL;
    {
        return sp_text_rows(M, threads, fd, bytes, [&](int64_t lo, int64_t hi, std::string &b) {
            char tmp[48];
            b.reserve((size_t)(hi - lo) * (size_t)(k + 40 + G * 24));
            for (int64_t i = lo; i < hi; i++) {
                sp_text_kmer(keys[i], k, tmp);
                b.append(tmp, (size_t)k);
                b.push_back('\t');
                b.append(nm[(size_t)top[i]]);
                b.push_back('\t');
                b.append(tmp, (size_t)sp_py_repr(pvals[i], tmp));
                b.push_back('\t');
                const double *row = means + i * G;
                for (int g = 0; g < G; g++) {
                    if (g) b.push_back(',');
                    b.append(tmp, (size_t)sp_py_repr(row[g], tmp));
                }
                b.push_back('\n');
            }
        });
    }
}
