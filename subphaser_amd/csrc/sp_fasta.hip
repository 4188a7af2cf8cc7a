// sp_fasta.hip -- host-side FASTA scanner (no kernels).
//
// The reference reads genomes and feature sets record by record through Bio.SeqIO (Seqs.py:27-71, 121-153); after
// the kernels that is what a run waits for: 14 GB of wheat FASTA, or 2 x 10^6 feature records.  This scanner walks
// the file image (mmap or decompressed bytes) once with a pool of threads:
//   1. record starts:  '>' at offset 0 or right after a line break
//   2. header end:     first line break after the start; the id is the first blank-delimited token of the header
//   3. sequence bytes: everything > 0x20 between the header and the next record, line breaks and blanks dropped,
//      written back to back into ONE caller-owned buffer (may be page-locked) + int64 offsets per record
// Work is cut into pieces of <= 8 MiB of file (a 670-Mb chromosome is ~80 pieces, a 5-kb feature is one), counted
// (kept bytes per piece), prefix-summed, then copied: every thread writes a disjoint range of the output.
#include "sp_common.h"

#include <atomic>
#include <cstring>
#include <thread>

struct sp_fasta {
    const uint8_t *data = nullptr;
    int64_t n = 0;
    int threads = 1;
    std::vector<int64_t> start, hdr_end;   // per record: offset of '>', offset of the line break ending the header
    std::vector<int64_t> seq_len;          // per record: bases kept
    struct piece {
        int64_t lo, hi, rec, out;          // file range, record, output offset
    };
    std::vector<piece> pieces;
    std::vector<int64_t> item;             // work items: pieces [item[i], item[i+1]) hold ~8 MiB of file together
    int64_t n_bases = 0;
};

// body(i) for every item on a pool of threads.  A body that throws (std::bad_alloc from a vector that grows) must
// not escape its std::thread -- that is std::terminate for the whole GPU process -- and the calling thread must not
// unwind past joinable threads: failures raise a flag, everything is joined, then ONE std::bad_alloc is rethrown
// on the caller's thread, where sp_fasta_open turns it into SP_ENOMEM.
template <typename F>
static void fasta_parallel(int threads, int64_t n_items, F &&body) {
    if (n_items <= 0) return;
    if (threads <= 1 || n_items == 1) {
        for (int64_t i = 0; i < n_items; i++) body(i);
        return;
    }
    std::atomic<int64_t> next(0);
    std::atomic<int> failed(0);
    auto run = [&]() noexcept {
        try {
            for (;;) {
                const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= n_items || failed.load(std::memory_order_relaxed)) return;
                body(i);
            }
        } catch (...) {
            failed.store(1);
        }
    };
    std::vector<std::thread> pool;
    const int nt = (int)(n_items < threads ? n_items : threads);
    try {
        pool.reserve((size_t)nt);
        for (int t = 1; t < nt; t++) pool.emplace_back(run);
    } catch (...) {   // thread creation failed: the ones that started (and this thread) do the work
    }
    run();
    for (auto &th : pool) th.join();
    if (failed.load()) throw std::bad_alloc();
}

static inline int64_t fasta_count_kept(const uint8_t *p, int64_t n) {
    int64_t c = 0;
    for (int64_t i = 0; i < n; i++) c += p[i] > 32;
    return c;
}

extern "C" int sp_fasta_open(const void *data, int64_t n, int threads, sp_fasta **out) {
    if (!out || n < 0 || (n > 0 && !data)) return SP_EINVAL;
    sp_fasta *h = new (std::nothrow) sp_fasta();
    if (!h) return SP_ENOMEM;
    h->data = (const uint8_t *)data;
    h->n = n;
    h->threads = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    const uint8_t *d = h->data;
    try {
        // 1. record starts, chunk by chunk
        const int64_t CH = (int64_t)16 << 20;
        const int64_t n_ch = (n + CH - 1) / CH;
        std::vector<std::vector<int64_t>> found((size_t)n_ch);
        fasta_parallel(h->threads, n_ch, [&](int64_t c) {
            const int64_t lo = c * CH, hi = lo + CH < n ? lo + CH : n;
            const uint8_t *p = d + lo, *e = d + hi;
            auto &v = found[(size_t)c];
            while (p < e) {
                p = (const uint8_t *)memchr(p, '>', (size_t)(e - p));
                if (!p) break;
                const int64_t i = p - d;
                if (i == 0 || d[i - 1] == '\n') v.push_back(i);
                p++;
            }
        });
        size_t R = 0;
        for (auto &v : found) R += v.size();
        h->start.reserve(R);
        for (auto &v : found) h->start.insert(h->start.end(), v.begin(), v.end());
        found.clear();
        h->hdr_end.assign(R, n);
        h->seq_len.assign(R, 0);
        // 2. header ends
        const int64_t RB = 4096;   // records per work item
        fasta_parallel(h->threads, ((int64_t)R + RB - 1) / RB, [&](int64_t b) {
            const int64_t r1 = (b + 1) * RB < (int64_t)R ? (b + 1) * RB : (int64_t)R;
            for (int64_t r = b * RB; r < r1; r++) {
                const int64_t s = h->start[(size_t)r];
                const void *q = memchr(d + s, '\n', (size_t)(n - s));
                h->hdr_end[(size_t)r] = q ? (const uint8_t *)q - d : n;
            }
        });
        // 3. pieces of the sequence regions
        const int64_t PIECE = (int64_t)8 << 20;
        for (size_t r = 0; r < R; r++) {
            int64_t lo = h->hdr_end[r] + 1;
            const int64_t hi = r + 1 < R ? h->start[r + 1] : n;
            if (lo > hi) lo = hi;
            if (lo == hi) continue;
            for (int64_t a = lo; a < hi; a += PIECE) h->pieces.push_back({a, a + PIECE < hi ? a + PIECE : hi, (int64_t)r, 0});
        }
        const int64_t P = (int64_t)h->pieces.size();
        {   // small pieces (features) are batched so that an item is worth a fetch_add
            int64_t acc = 0;
            h->item.push_back(0);
            for (int64_t i = 0; i < P; i++) {
                acc += h->pieces[(size_t)i].hi - h->pieces[(size_t)i].lo;
                if (acc >= PIECE) {
                    h->item.push_back(i + 1);
                    acc = 0;
                }
            }
            if (h->item.back() != P) h->item.push_back(P);
        }
        fasta_parallel(h->threads, (int64_t)h->item.size() - 1, [&](int64_t b) {
            const int64_t p1 = h->item[(size_t)b + 1];
            for (int64_t i = h->item[(size_t)b]; i < p1; i++) {
                auto &pc = h->pieces[(size_t)i];
                pc.out = fasta_count_kept(d + pc.lo, pc.hi - pc.lo);   // count now, offset after the prefix sum
            }
        });
        int64_t run = 0;
        for (auto &pc : h->pieces) {
            const int64_t c = pc.out;
            h->seq_len[(size_t)pc.rec] += c;
            pc.out = run;
            run += c;
        }
        h->n_bases = run;
    } catch (const std::bad_alloc &) {
        delete h;
        return SP_ENOMEM;
    }
    *out = h;
    return SP_OK;
}

extern "C" int sp_fasta_counts(const sp_fasta *h, int64_t *n_records, int64_t *n_bases) {
    if (!h || !n_records || !n_bases) return SP_EINVAL;
    *n_records = (int64_t)h->start.size();
    *n_bases = h->n_bases;
    return SP_OK;
}

// hdr_start[r] .. hdr_end[r]: the header line without '>' and without its line break (the caller takes the first
// token as the id); seq_off[r] .. seq_off[r + 1]: the record's bases in `cat` (n_bases bytes, caller-owned)
extern "C" int sp_fasta_fetch(const sp_fasta *h, int64_t *hdr_start, int64_t *hdr_end, int64_t *seq_off, void *cat) {
    if (!h || !hdr_start || !hdr_end || !seq_off || (h->n_bases > 0 && !cat)) return SP_EINVAL;
    const size_t R = h->start.size();
    int64_t run = 0;
    for (size_t r = 0; r < R; r++) {
        hdr_start[r] = h->start[r] + 1;
        int64_t e = h->hdr_end[r];
        if (e > hdr_start[r] && h->data[e - 1] == '\r') e--;
        hdr_end[r] = e < hdr_start[r] ? hdr_start[r] : e;
        seq_off[r] = run;
        run += h->seq_len[r];
    }
    seq_off[R] = run;
    uint8_t *out = (uint8_t *)cat;
    const uint8_t *d = h->data;
    try {
    fasta_parallel(h->threads, (int64_t)h->item.size() - 1, [&](int64_t b) {
        const int64_t p1 = h->item[(size_t)b + 1];
        for (int64_t i = h->item[(size_t)b]; i < p1; i++) {
            const auto &pc = h->pieces[(size_t)i];
            const uint8_t *p = d + pc.lo, *e = d + pc.hi;
            uint8_t *o = out + pc.out;
            // line by line: one memchr + one memcpy per line; a line that carries blanks takes the byte loop
            while (p < e) {
                const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
                const uint8_t *le = nl ? nl : e;
                const int64_t len = le - p;
                const int64_t kept = fasta_count_kept(p, len);
                if (kept == len) {
                    memcpy(o, p, (size_t)len);
                    o += len;
                } else {
                    for (const uint8_t *q = p; q < le; q++)
                        if (*q > 32) *o++ = *q;     // (no speculative store: the next byte of `out` is another piece's)
                }
                p = nl ? nl + 1 : e;
            }
        }
    });
    } catch (...) {
        return SP_ENOMEM;
    }
    return SP_OK;
}

extern "C" void sp_fasta_close(sp_fasta *h) { delete h; }
