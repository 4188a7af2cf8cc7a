// sp_sparse2.hip -- hand-written count engine for k = 16..32 ("engine 3").
//
// The first sparse engine sorted every chromosome's 64-bit keys with a device-wide LSD radix sort
// (rocPRIM, 5-6 passes x 16 B per key r+w: 480 of the 780 ms of a wheat-like pass at k = 17).  This
// engine is the 64-bit twin of engine 2: an MSD partition that moves every key twice, in shrinking
// form, and finishes inside LDS.
//
//   s3_hist1   scan the packed chromosome (64-bit direct-window scan, sp_device.h), histogram the top
//              B1 key bits (LDS, F1 <= 1024 bins)
//   s3_part1   scan again, the 32 keys of a thread stay in registers; LDS counting sort of a tile on
//              those bits (rank = what the histogram atomic returns); every bucket run goes out as one
//              burst (its run found by prefix popcount over a bitmap of run heads); only the low
//              R1 = 2k - B1 bits survive (u32 for k <= 21, else u64)
//   s3_hist2   per level-1 bucket: histogram of the next B2 = 9 bits  -> exact fine-bucket offsets
//   s3_part2   per level-1 bucket: LDS counting sort of 8 K-key tiles on those bits, software-pipelined;
//              R2 = R1 - 9 bits survive (u32 for k <= 25)
//   s3_final_bitmap (k = 16, 17: R2 <= 16) one workgroup per fine bucket (2^18..2^19 of them, ~1-3 K
//              keys): bitmap + prefix popcount ranking, no sort
//   s3_final_small / s3_final (k >= 18, and what the bitmap kernel hands over): block radix sort of
//              the residuals in registers/LDS, run-length encode, keep count >= lower_count
//   s3_gather  ordered compaction of the kept (key, count) pairs -> the chromosome's sorted list
//
// Fine buckets that neither fit one workgroup nor split (hot keys such as telomere repeats: ~70 per wheat-sized
// chromosome at k = 21) are sorted together by one segmented device sort and run-length encoded by s3_big_rle.
// Bytes per key (k = 21): 0.375 x 3 scans + 4 w + 4 r + 4 r + 4 w + 4 r  ~ 21 B against ~100 B.
#include "sp_device.h"

#define S3_MAXF 1024
#define S3_P1_UNIT 32
#ifndef S3_P2_THREADS
#define S3_P2_THREADS 512
#endif
#ifndef S3_P2_PER
#define S3_P2_PER 16
#endif
#define S3_P2_KEYS (S3_P2_THREADS * S3_P2_PER)   // keys per part2 / hist2 tile
#define S3_SORT_THREADS 256
#ifndef S3_SORT_PER
#define S3_SORT_PER 8      // (16: s3_final 140 -> 131 ms per wheat-like pass at k = 21, but 11 -> 16 ms at k = 17 and more big-list buckets)
#endif
#define S3_SORT_CAP (S3_SORT_THREADS * S3_SORT_PER)   // keys one workgroup sorts

struct s3_plan {
    int T, B1, B2, R1, R2, F1, F2;
    bool wide1, wide2;   // residuals after level 1 / level 2 need 64 bits
};

static s3_plan s3_make_plan(int k) {
    s3_plan p;
    p.T = 2 * k;
    p.B1 = 10;   // k <= 21: the residual (2k - 10 bits) fits 32 bits
#ifndef S3_B2
#define S3_B2 9
#endif
    // Round 6: k = 16 / 17 split level 2 into 2^8 instead of 2^9: the bitmap finish takes 16-bit residuals anyway, and with
    // half as many fine buckets of twice the keys (2.6 K on a wheat-sized chromosome) s3_part2 writes runs of twice the
    // length and every per-bucket step of the finish -- seven barriers, two block scans -- is paid half as often:
    // wheat-like k = 17 pass 213.2 -> 195.5 ms (s3_part2 149 -> 115 ms of events).  k >= 18 (hash finish) stays at 2^9:
    // 2^8 sends its buckets past the table's capacity (k = 21: 429 ms), 2^10 is slower as well (228 against 220).
    p.B2 = k <= 17 ? 8 : S3_B2;
    p.R1 = p.T - p.B1;
    p.R2 = p.R1 - p.B2;
    p.F1 = 1 << p.B1;
    p.F2 = 1 << p.B2;
    p.wide1 = p.R1 > 32;
    p.wide2 = p.R2 > 32;
    return p;
}

// block-wide exclusive scan of hist[0..F) (F <= 1024) -> start[]; returns the total
template <int THREADS>
__device__ __forceinline__ uint32_t s3_block_scan(const uint32_t *hist, uint32_t *start, int F, uint32_t *wsum) {
    constexpr int NW = THREADS / 64;
    const int per = (F + THREADS - 1) / THREADS;
    const int lo = threadIdx.x * per;
    uint32_t v = 0;
    for (int i = 0; i < per; i++)
        if (lo + i < F) v += hist[lo + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t s = wsum[w];
        if (w < wave) base += s;
        total += s;
    }
    uint32_t run = base + incl - v;
    for (int i = 0; i < per; i++)
        if (lo + i < F) {
            start[lo + i] = run;
            run += hist[lo + i];
        }
    __syncthreads();
    return total;
}

// exclusive prefix of v over a THREADS-wide block (value in a register); *total = block sum.  Ends with a barrier.
template <int THREADS>
__device__ __forceinline__ uint32_t s3_scan_reg_t(uint32_t v, uint32_t *wsum /* >= THREADS/64 */, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    __syncthreads();            // wsum may still be read by a previous scan's consumers
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) {
        const uint32_t x = wsum[w];
        if (w < wave) base += x;
        tot += x;
    }
    *total = tot;
    __syncthreads();
    return base + incl - v;
}

// Round 3: the level-1 regions, too, come from a sample: s3_hist1 scans ONE STRIPE of 64 units (2048 starts) IN 16,
// rotating through the groups like engine 2's sampler; a level-1 bucket receives ~40 K sampled keys of a wheat-sized
// chromosome, so the estimate is good to a fraction of a per cent and the capacity s * 16 * 33/32 + slack wastes
// nothing to speak of.  Runs that do not fit land in a trash area behind the regions, s3_tiles raises the flag and
// the chromosome is counted again from the exact histograms.
#define S3_STRIPE 64
#define S3_SAMPLE_SHIFT 4
#define S3_SAMPLE_MIN_LEN (1LL << 24)     // shorter chromosomes: exact level-1 histogram (and the key count with it)
__device__ __forceinline__ int64_t s3_sample_unit(int64_t j, int sample_shift) {
    if (sample_shift == 0) return j;
    const int64_t g = j / S3_STRIPE;
    const int64_t phase = (g * 7 + (g >> 4)) & ((1 << sample_shift) - 1);
    return ((g << sample_shift) + phase) * S3_STRIPE + (j % S3_STRIPE);
}
__global__ void __launch_bounds__(1024)
s3_caps1(unsigned long long *__restrict__ hist1, int F1, unsigned long long mult32, unsigned long long slack) {
    const int b = threadIdx.x;
    if (b < F1) hist1[b] = ((hist1[b] << S3_SAMPLE_SHIFT) * mult32) / 32ULL + slack;
}
// ---------------------------------------------------------------- s3_hist1
__global__ void __launch_bounds__(256)
s3_hist1(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm,
         int64_t n_units /* of 32 starts */, int64_t n_visit, int sample_shift, sp_kparams kp, int R1, int F1,
         unsigned long long *__restrict__ ghist) {
    __shared__ uint32_t lh[S3_MAXF];
    for (int i = threadIdx.x; i < F1; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_visit; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = s3_sample_unit(j, sample_shift);
        if (u >= n_units) continue;
        sp_scan32_valid64(pk, pm, nm, u * S3_P1_UNIT, kp, [&](int64_t, uint64_t fwd, uint64_t rc) {
            const uint64_t key = fwd < rc ? fwd : rc;
            atomicAdd(&lh[key >> R1], 1u);
        });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F1; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&ghist[i], (unsigned long long)v);
    }
}

// ---------------------------------------------------------------- s3_part1
// Output runs are reserved with one
// global atomic per (tile, non-empty bucket); the order inside a bucket does not matter downstream.
// Round 5, last third: TWO workgroups per CU for 32-bit residuals (k <= 21).  The kernel ran one (89 KB of LDS, 160 VGPRs at 512
// threads), and its twin of the dense engine, c2_part1, loses half of its speed when it is held to one (0.78 -> 1.16 ms).  LDS:
// the runs' global bases and the region ends are values a thread writes and reads back itself (the same buckets every tile):
// registers, 8 KB less -- 80 944 B.  VGPRs: a key of <= 42 bits carries its 14-bit rank inside the run in its bits 48..61
// instead of in a second array of 32 registers (S3_P1_PACK) -- 128 without a spill.
#ifndef S3_P1_MINW
#define S3_P1_MINW 4      // waves per SIMD the register allocation leaves room for (512 threads: two workgroups per CU)
#endif
template <typename KR1, int THREADS>
__global__ void __launch_bounds__(THREADS, (sizeof(KR1) == 4 ? S3_P1_MINW : 1))
s3_part1(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm,
         int64_t n_units /* of 32 starts */, sp_kparams kp, int R1, int F1, unsigned long long *__restrict__ cursor1, KR1 *__restrict__ buf1,
         int64_t n_tiles, const unsigned long long *__restrict__ off1 /* F1+1: where the regions start */, uint32_t trash) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s3_lds[];
    KR1 *keys = reinterpret_cast<KR1 *>(s3_lds);                        // [THREADS * 32]
    __shared__ uint32_t hist[S3_MAXF], start[S3_MAXF + 1], delta[S3_MAXF], wsum[THREADS / 64];
    __shared__ uint32_t head[THREADS];      // tile positions / 32 = THREADS words
    if (n_tiles <= 0 || n_units <= 0) return;      // (the prefetch below is unconditional)
    constexpr int BPT = (S3_MAXF + THREADS - 1) / THREADS;      // level-1 buckets per thread: b = threadIdx.x + t * THREADS
    constexpr bool PACK = sizeof(KR1) == 4;                     // keys of <= 42 bits: the rank rides in bits 48..61
    uint32_t lim_[BPT];                                         // end of the thread's buckets' regions (read once)
#pragma unroll
    for (int t = 0; t < BPT; t++) {
        const int b = threadIdx.x + t * THREADS;
        lim_[t] = b < F1 ? (uint32_t)off1[b + 1] : 0u;
    }
    __shared__ uint16_t hpre[THREADS];
    const uint64_t rmask = (R1 >= 64) ? ~0ULL : ((1ULL << R1) - 1ULL);
    // the NEXT tile's unit (validity words + both packed streams) travels while this tile is sorted and written: copy
    // to working registers, issue the next loads, then work (profiles/r03_notes.md)
    sp_words64 p_x;
    uint32_t p_nm0 = 0, p_nm1 = 0;
    auto fetch = [&](int64_t t) {
        // (round 5: unconditional, from a clamped unit -- a load under a condition is merged with the old value through
        // moves that wait for it on the spot, which made this "prefetch" a plain load; as in c2_part1)
        int64_t u = (t < n_tiles ? t : n_tiles - 1) * THREADS + threadIdx.x;
        u = u < n_units ? u : n_units - 1;
        p_nm0 = nm[u];              // (u * 32) >> 5
        p_nm1 = nm[u + 1];
        p_x = sp_load_words64(pk, pm, u * S3_P1_UNIT);
    };
    fetch(blockIdx.x);
    static_assert(S3_MAXF <= 2 * THREADS || THREADS == 256, "two cursor atomics per thread (four at 256 threads)");
    static_assert(THREADS * S3_P1_UNIT <= (1 << 14), "a rank inside a tile fits 14 bits");
    const uint64_t kmask = PACK ? ((1ULL << 48) - 1ULL) : ~0ULL;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (int i = threadIdx.x; i < F1; i += THREADS) hist[i] = 0;
        const sp_words64 x = p_x;
        const uint32_t c_nm0 = p_nm0, c_nm1 = p_nm1;
        fetch(tile + gridDim.x);
        __syncthreads();
        // ONE scan: the unit's 32 canonical keys stay in registers (the 64-KiB tile leaves one block per CU, so
        // registers are plentiful); the rank inside the bucket run is what the histogram atomic returns
        const int64_t u = tile * THREADS + threadIdx.x;
        uint64_t key[32];
        uint32_t rank[PACK ? 1 : 32], ok = 0;
        if (u < n_units) {
            ok = ~(uint32_t)sp_bad_from_words64((uint64_t)c_nm0 | ((uint64_t)c_nm1 << 32), kp.k);
            if (ok) {
                sp_scan32_keys64(x, kp, [&](int j, uint64_t fwd, uint64_t rc) {
                    key[j] = fwd < rc ? fwd : rc;
                    if ((ok >> j) & 1u) {
                        const uint32_t rk = atomicAdd(&hist[key[j] >> R1], 1u);
                        if (PACK) key[j] |= (uint64_t)rk << 48;
                        else rank[j] = rk;
                    }
                });
            }
        }
        __syncthreads();
        // (the cursors' answers are looked at only after the scan and the placement: stored to LDS here, the threads
        // sat out the atomics' round trip in front of the barriers the whole workgroup waits at)
        unsigned long long at_[BPT];
#pragma unroll
        for (int t = 0; t < BPT; t++) {
            const int b = threadIdx.x + t * THREADS;
            at_[t] = 0;
            if (b < F1) {
                const uint32_t c = hist[b];
                if (c) at_[t] = atomicAdd(&cursor1[b], (unsigned long long)c);
            }
        }
        const uint32_t total = s3_block_scan<THREADS>(hist, start, F1, wsum);
        if (threadIdx.x == 0) start[F1] = total;
        head[threadIdx.x] = 0;                                          // one bit per tile position: a run starts here
        __syncthreads();
        for (int b = threadIdx.x; b < F1; b += THREADS)
            if (hist[b]) atomicOr(&head[start[b] >> 5], 1u << (start[b] & 31));
        // (the run starts of eight keys are read together: one LDS round trip per eight keys instead of one per key)
#pragma unroll
        for (int j0 = 0; j0 < 32; j0 += 8) {
            uint32_t st[8];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) st[jj] = (ok >> (j0 + jj)) & 1u ? start[(uint32_t)((key[j0 + jj] & kmask) >> R1)] : 0u;
#pragma unroll
            for (int jj = 0; jj < 8; jj++)
                if ((ok >> (j0 + jj)) & 1u)
                    keys[st[jj] + (PACK ? (uint32_t)(key[j0 + jj] >> 48) : rank[PACK ? 0 : j0 + jj])] = (KR1)(key[j0 + jj] & rmask);
        }
        // Which run does sorted position i belong to?  A binary search over start[] (10 dependent LDS reads and ~60
        // VALU instructions per key) was most of this kernel; instead: rank of the last run head at or before i,
        // from a prefix popcount over the head bitmap, indexes the runs' (global base - tile start) table.
        // (round 5: a BARRIER between the run-head bits and their first read.  Word w of head[] is read by thread w and
        // set by whichever threads own the buckets that start in it; without the barrier a wave that got ahead counted a
        // word before another wave had set its bits, the run ranks of the tile shifted, and a few dozen runs went to
        // their neighbours' regions -- keys counted under another bucket's high bits, one pass in ~300 at wheat scale
        // with three chains in flight, found by tools/stress_lanes.py.  Since round 3 the 32 dependent placements
        // between the two had kept the window shut; batching them in round 5 opened it.)
        __syncthreads();
        uint32_t tot_runs;
        const uint32_t wpre = s3_scan_reg_t<THREADS>((uint32_t)__popc(head[threadIdx.x]), wsum, &tot_runs);
        hpre[threadIdx.x] = (uint16_t)wpre;      // ends with a barrier: placement is complete as well
        __syncthreads();
#pragma unroll
        for (int t = 0; t < BPT; t++) {
            const int b = threadIdx.x + t * THREADS;
            if (b < F1 && hist[b]) {       // (the placement cursor = the run's length)
                const uint32_t p0 = start[b];
                const uint32_t r = (uint32_t)hpre[p0 >> 5] + (uint32_t)__popc(head[p0 >> 5] & ((1u << (p0 & 31)) - 1u));
                // (sampled regions: a run that does not fit goes to the trash area behind them; s3_tiles sees the cursor)
                const uint32_t at = (uint32_t)at_[t];
                delta[r] = ((unsigned long long)at + hist[b] <= (unsigned long long)lim_[t] ? at : trash) - p0;
            }
        }
        __syncthreads();
        // (four positions per round: their LDS reads, then the run look-ups, then the stores)
        for (uint32_t i0 = threadIdx.x; i0 < total; i0 += 4u * THREADS) {
            uint32_t hp[4], hd[4];
            KR1 kv[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t i = i0 + (uint32_t)t * THREADS, ic = i < total ? i : total - 1u;
                hp[t] = hpre[ic >> 5];
                hd[t] = head[ic >> 5];
                kv[t] = keys[ic];
            }
            uint32_t dl[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t i = i0 + (uint32_t)t * THREADS, ic = i < total ? i : total - 1u;
                dl[t] = delta[hp[t] + (uint32_t)__popc(hd[t] & (0xFFFFFFFFu >> (31 - (ic & 31)))) - 1u];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t i = i0 + (uint32_t)t * THREADS;
                if (i < total) buf1[(size_t)(dl[t] + i)] = kv[t];
            }
        }
        __syncthreads();
    }
}

// tile bookkeeping shared by hist2 / part2: level-1 bucket b owns keys [off1[b], end1[b]) of buf1 (what part1's cursor
// says it wrote; the region may be larger) and tiles [tile_start[b], tile_start[b+1]) of S3_P2_KEYS keys.  Also the
// number of keys of the chromosome and the overrun flag.  (One thread walked the 1024 buckets until round 3: 95 us.)
__global__ void __launch_bounds__(1024)
s3_tiles(const unsigned long long *__restrict__ off1 /* F1+1 */, const unsigned long long *__restrict__ cursor1, int F1,
         unsigned long long *__restrict__ tile_start /* F1+1 */, unsigned long long *__restrict__ end1 /* F1 */,
         unsigned long long *__restrict__ n_keys, unsigned long long *__restrict__ flag) {
    __shared__ unsigned long long lds[16];
    const int b = threadIdx.x;
    unsigned long long n = 0;
    if (b < F1) {
        const unsigned long long lo = off1[b], cap = off1[b + 1] - lo;
        n = cursor1[b] - lo;
        if (n > cap) {
            n = cap;
            atomicAdd(flag, 1ULL);
        }
        end1[b] = lo + n;
    }
    unsigned long long tot_tiles, tot_keys;
    const unsigned long long t0 = sp_block_excl_scan<unsigned long long>((n + S3_P2_KEYS - 1) / S3_P2_KEYS, lds, tot_tiles);
    sp_block_excl_scan<unsigned long long>(n, lds, tot_keys);
    if (b < F1) tile_start[b] = t0;
    if (b == 0) {
        tile_start[F1] = tot_tiles;
        *n_keys = tot_keys;
    }
}

__device__ __forceinline__ int s3_bucket_of(const unsigned long long *__restrict__ tile_start, int F1,
                                            unsigned long long tile) {
    int lo = 0, hi = F1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_start[mid] <= tile) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------- s3_hist2
// Each block walks a contiguous range of tiles and keeps the histogram of the current level-1 bucket in
// LDS; it is flushed when the bucket changes.
template <typename KR1>
__global__ void __launch_bounds__(S3_P2_THREADS)
s3_hist2(const KR1 *__restrict__ buf1, const unsigned long long *__restrict__ off1, const unsigned long long *__restrict__ end1,
         const unsigned long long *__restrict__ tile_start, int F1, int F2, int R2,
         unsigned long long *__restrict__ hist2 /* F1 * F2 */, int sample /* 1: two rows of 16 of every tile */) {
    __shared__ uint32_t lh[S3_MAXF];
    __shared__ int s_b;
    const unsigned long long n_tiles = tile_start[F1];
    const unsigned long long per = (n_tiles + gridDim.x - 1) / gridDim.x;
    unsigned long long t0 = (unsigned long long)blockIdx.x * per, t1 = t0 + per;
    if (t1 > n_tiles) t1 = n_tiles;
    if (t0 >= t1) return;
    const uint32_t mask2 = (uint32_t)F2 - 1u;
    for (int i = threadIdx.x; i < F2; i += blockDim.x) lh[i] = 0;
    if (threadIdx.x == 0) s_b = s3_bucket_of(tile_start, F1, t0);
    __syncthreads();
    int cb = s_b;
    for (unsigned long long tile = t0; tile < t1; tile++) {
        if (tile >= tile_start[cb + 1]) {   // block-uniform: flush and move to the bucket of this tile
            __syncthreads();
            for (int i = threadIdx.x; i < F2; i += blockDim.x) {
                const uint32_t v = lh[i];
                if (v) atomicAdd(&hist2[(size_t)cb * F2 + i], (unsigned long long)v);
                lh[i] = 0;
            }
            while (tile >= tile_start[cb + 1]) cb++;
            __syncthreads();
        }
        const unsigned long long base = off1[cb] + (tile - tile_start[cb]) * S3_P2_KEYS, end = end1[cb];
        if (sample) {       // one 64-key group in eight, from every row of every tile: wave (tile mod 8) reads its slices
            if ((int)(threadIdx.x >> 6) == (int)(tile & 7ULL)) {
#pragma unroll
                for (int j = 0; j < S3_P2_PER; j++) {
                    const unsigned long long idx = base + (unsigned long long)j * S3_P2_THREADS + threadIdx.x;
                    if (idx < end) atomicAdd(&lh[(uint32_t)(buf1[idx] >> R2) & mask2], 1u);
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < S3_P2_PER; j++) {
            const unsigned long long idx = base + (unsigned long long)j * S3_P2_THREADS + threadIdx.x;
            if (idx < end) atomicAdd(&lh[(uint32_t)(buf1[idx] >> R2) & mask2], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F2; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist2[(size_t)cb * F2 + i], (unsigned long long)v);
    }
}

// Round 3: level-2 regions from a SAMPLE.  s3_hist2 reads one 64-key group in eight of every row of every tile (an
// eighth of the level-1 buffer in 256-byte pieces: 17 -> 5 ms per wheat-like pass; whole rows of 512 keys were cheaper
// still but missed the ~2 K-key chunks in which a tandem array reaches a bucket, and the 30 K-key telomere buckets of the
// wheat-like genome overran their regions on most chromosomes); s3_caps turns a sampled count s into a capacity 10 s + 8 ceil(7 sqrt(s)) +
// S3_CAP_SLACK (1.25 x the estimate, seven standard deviations of it, and a constant -- memory is plentiful), the scan of the capacities lays the regions out, s3_part2 drops what does not fit and s3_spans -- which
// also turns the cursors, updated by atomics, into plainly stored {offset, size} records for the kernels that follow --
// raises a flag; the chromosome is then counted again with exact sizes (sample = 0), so results never depend on the
// estimate.
#ifndef S3_CAP_SLACK
#define S3_CAP_SLACK 1024ULL
#endif
#define S3_DROP (~0ULL)
static_assert(S3_P2_THREADS == 512, "the sample takes one of the eight waves of a tile row");
#define S3_CAP_MULT 10ULL         // x sampled count (the sample is an eighth: 1.25 x the estimate)
__global__ void __launch_bounds__(256)
s3_caps(unsigned long long *__restrict__ hist2, int64_t n_fine, unsigned long long mult, unsigned long long slack) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_fine) return;
    const unsigned long long s = hist2[f];
    hist2[f] = mult * s + 8ULL * (unsigned long long)ceilf(7.0f * sqrtf((float)s)) + slack;
}
__global__ void __launch_bounds__(256)
s3_spans(const unsigned long long *__restrict__ off_fine, const unsigned long long *__restrict__ cursor2, int64_t n_fine,
         ulonglong2 *__restrict__ span, unsigned long long *__restrict__ flag) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_fine) return;
    const unsigned long long lo = off_fine[f], cap = off_fine[f + 1] - lo;
    unsigned long long n = cursor2[f];
    if (n > cap) {      // a dropped run leaves part of the region unwritten: nobody may read it (the chromosome is recounted)
        n = 0;
        atomicAdd(flag, 1ULL);
    }
    span[f] = make_ulonglong2(lo, n);
}

// ---------------------------------------------------------------- s3_part2
template <typename KR1, typename KR2>
__global__ void __launch_bounds__(S3_P2_THREADS)
s3_part2(const KR1 *__restrict__ buf1, const unsigned long long *__restrict__ off1, const unsigned long long *__restrict__ end1,
         const unsigned long long *__restrict__ tile_start, int F1, int F2, int R2,
         const unsigned long long *__restrict__ off_fine, unsigned long long *__restrict__ cursor2,
         KR2 *__restrict__ buf2) {
    __shared__ uint32_t hist[S3_MAXF], start[S3_MAXF], wsum[S3_P2_THREADS / 64];
    __shared__ unsigned long long gdelta[S3_MAXF];     // global base of a run - its start inside the tile
    __shared__ KR1 keys[S3_P2_KEYS];
    __shared__ int s_b[2];
    const unsigned long long n_tiles = tile_start[F1];
    const uint32_t mask2 = (uint32_t)F2 - 1u;
    const uint64_t rmask = (R2 >= 64) ? ~0ULL : ((1ULL << R2) - 1ULL);
    unsigned long long tile = blockIdx.x;
    if (tile >= n_tiles) return;
    if (threadIdx.x == 0) s_b[0] = s3_bucket_of(tile_start, F1, tile);
    __syncthreads();
    // software pipeline as in c2_part2: the keys of the block's NEXT tile travel while this one is ranked (the rank
    // of a key inside its run is what the histogram atomic returns), scattered and written out
    KR1 nxt[S3_P2_PER];
    int nnext = 0;
    auto fetch = [&](int nb, unsigned long long t) {
        const unsigned long long base = off1[nb] + (t - tile_start[nb]) * S3_P2_KEYS, end = end1[nb];
        nnext = 0;
#pragma unroll
        for (int j = 0; j < S3_P2_PER; j++) {
            const unsigned long long idx = base + (unsigned long long)j * S3_P2_THREADS + threadIdx.x;
            if (idx < end) {
                nxt[j] = buf1[idx];
                nnext = j + 1;
            }
        }
    };
    fetch(s_b[0], tile);
    int p = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        const int b1 = s_b[p];
        KR1 my[S3_P2_PER];
#pragma unroll
        for (int j = 0; j < S3_P2_PER; j++) my[j] = nxt[j];
        const int nmine = nnext;
        const unsigned long long ntile = tile + gridDim.x;
        for (int i = threadIdx.x; i < F2; i += S3_P2_THREADS) hist[i] = 0;
        if (threadIdx.x == 0 && ntile < n_tiles) s_b[p ^ 1] = s3_bucket_of(tile_start, F1, ntile);
        __syncthreads();   // (A) also: the previous tile's copy-out has finished reading keys / start / gdelta
        uint32_t rank[S3_P2_PER];
#pragma unroll
        for (int j = 0; j < S3_P2_PER; j++)
            if (j < nmine) rank[j] = atomicAdd(&hist[(uint32_t)(my[j] >> R2) & mask2], 1u);
        __syncthreads();   // (B)
        unsigned long long g[(S3_MAXF + S3_P2_THREADS - 1) / S3_P2_THREADS];
#pragma unroll
        for (int q = 0; q < (S3_MAXF + S3_P2_THREADS - 1) / S3_P2_THREADS; q++) {
            const int d = threadIdx.x + q * S3_P2_THREADS;
            g[q] = 0;
            if (d < F2) {
                const uint32_t c = hist[d];
                const size_t fine = (size_t)b1 * F2 + d;
                const unsigned long long o0 = off_fine[fine], at = c ? atomicAdd(&cursor2[fine], (unsigned long long)c) : 0ULL;
                // (sampled sizes: a run that does not fit its region is dropped, s3_spans raises the flag)
                g[q] = at + c <= off_fine[fine + 1] - o0 ? o0 + at : S3_DROP;
            }
        }
        nnext = 0;
        if (ntile < n_tiles) fetch(s_b[p ^ 1], ntile);
        const uint32_t total = s3_block_scan<S3_P2_THREADS>(hist, start, F2, wsum);
#pragma unroll
        for (int q = 0; q < (S3_MAXF + S3_P2_THREADS - 1) / S3_P2_THREADS; q++) {
            const int d = threadIdx.x + q * S3_P2_THREADS;
            if (d < F2) gdelta[d] = g[q] == S3_DROP ? S3_DROP : g[q] - start[d];
        }
#pragma unroll
        for (int j = 0; j < S3_P2_PER; j++)
            if (j < nmine) keys[start[(uint32_t)(my[j] >> R2) & mask2] + rank[j]] = my[j];
        __syncthreads();   // (D)
        for (uint32_t i = threadIdx.x; i < total; i += S3_P2_THREADS) {
            const KR1 kk = keys[i];
            const unsigned long long gd = gdelta[(uint32_t)(kk >> R2) & mask2];
            if (gd != S3_DROP) buf2[gd + i] = (KR2)((uint64_t)kk & rmask);
        }
        p ^= 1;
    }
}

// ---------------------------------------------------------------- s3_final
// One workgroup per fine bucket.  <= 4096 residuals: block radix sort in registers/LDS, run-length
// encode, keep runs >= lower.  Larger buckets are first split on their next <= 10 residual bits into a
// scratch area (the level-1 buffer, free by now; the bucket is L2-sized, so this costs no HBM traffic) and
// the pieces go through the same sort one after the other; a piece that is still too large is counted
// directly in LDS when <= 13 residual bits are left (always the case for k <= 21), is a single hot key, or
// sends the whole bucket to the device-wide fallback.  Kept (residual, count) pairs are written at the
// bucket's own offset of the scratch arrays in ascending order, their number into kept[bucket].
#define S3_SPLIT_BITS 8
#define S3_DIRECT_BITS 12

#define S3_SMALL_PER 8
#define S3_SMALL_CAP (S3_SORT_THREADS * S3_SMALL_PER)   // buckets up to this size take the light kernel

template <typename KR2, int PER>
struct s3_sort_lds {
    KR2 s[S3_SORT_THREADS * PER];
    uint32_t heads[S3_SORT_THREADS * PER + 1];
    uint32_t wsum[S3_SORT_THREADS / 64], cnt1[S3_SORT_THREADS], cnt2[S3_SORT_THREADS];
};

#define S3_HASH_SLOTS 2048                       // LDS hash table of the hot-key path
#define S3_HASH_MAX (S3_HASH_SLOTS * 3 / 4)      // distinct residuals it accepts

// (s3_final_hash's table -- described at the kernel below -- is also what s3_final counts its pieces with)
#define S3H_COUNTERS 4096
#define S3H_SLOTS 2048
#define S3H_CAP 1536            // distinct residuals the table accepts
#define S3H_PER 8               // keys per thread held in registers: buckets up to 2048 keys are read once
template <typename KR2>
struct __attribute__((aligned(16))) s3h_lds {
    union __attribute__((aligned(16))) {
        uint32_t c1[S3H_COUNTERS];
        struct {
            KR2 key[S3H_SLOTS];
            uint32_t cnt[S3H_SLOTS];
        } t;
    } u;
    KR2 lk[S3H_CAP + 64];                       // the kept residuals: the list is written only when the table did not close, so
                                                // it holds <= S3H_CAP of them (their counts are looked up in the table again when they
                                                // are written: a list of counts next to this one was 7 KB more -- 30.2 KB,
                                                // five workgroups per CU; 22.9 KB: seven.  s3_final_hash follows its occupancy:
                                                // four workgroups 4.53 ms, five 3.96 at k = 21)
    uint32_t n_distinct, n_kept, abort_;
    unsigned long long red[16];
};
__device__ __forceinline__ uint32_t s3h_hash2(unsigned long long v) {
    v ^= v >> 31;
    v *= 0xD6E8FEB86659FD93ULL;
    return (uint32_t)(v >> 37);
}

template <typename KR2>
struct s3_final_lds {
    union {
        s3_sort_lds<KR2, S3_SORT_PER> q;     // also the hot-key path's (key, count) staging
        s3_sort_lds<KR2, 4> q4;
        s3_sort_lds<KR2, 2> q2;
        s3h_lds<KR2> h;                      // pieces counted instead of sorted (k >= 18)
    };
    union {
        uint32_t direct[1 << S3_DIRECT_BITS];
        struct {
            KR2 key[S3_HASH_SLOTS];
            uint32_t cnt[S3_HASH_SLOTS];
        } hash;
    } u;
    uint32_t sub_cnt[1 << S3_SPLIT_BITS], sub_off[(1 << S3_SPLIT_BITS) + 1];
    unsigned long long red[16];
    uint32_t ones, distinct;
};

__device__ __forceinline__ uint32_t s3_hash(unsigned long long v) {
    v ^= v >> 29;
    v *= 0x9E3779B97F4A7C15ULL;
    return (uint32_t)(v >> 40);
}

// ---- the workgroup's sort: a bitonic network over S3_SORT_THREADS x PER keys, thread t owning the PER consecutive
// positions t * PER ...  Compare-exchange steps whose partner sits in the same thread run in registers (distance < PER),
// steps whose partner sits in the same wave run as lane exchanges (distance < 64 PER), and only the steps across waves
// go through LDS and a barrier: 3 of the 66 steps of a 2048-key sort, 1 of 36 at 256 keys.  Keys come back in the
// blocked arrangement they went in with, ascending.  With PAIRS a 32-bit value travels with each key and equal keys
// are ordered by DESCENDING value (a network is not stable: this is what keeps a real (all-ones, count) entry ahead of
// the (all-ones, 0) padding).  xk / xv: S3_SORT_THREADS * PER words each of exchange space, free on entry (the caller
// put a barrier behind its last use), free again on return.
template <typename K, bool PAIRS>
__device__ __forceinline__ bool s3_kv_less(K ka, uint32_t va, K kb, uint32_t vb) {
    if (PAIRS) return ka < kb || (ka == kb && va > vb);
    return ka < kb;
}
template <typename K, bool PAIRS>
__device__ __forceinline__ void s3_cmpx(K &ka, uint32_t &va, K &kb, uint32_t &vb, bool up) {
    // up: the smaller one ends in a
    const bool sw = up ? s3_kv_less<K, PAIRS>(kb, vb, ka, va) : s3_kv_less<K, PAIRS>(ka, va, kb, vb);
    const K ta = sw ? kb : ka, tb = sw ? ka : kb;
    ka = ta;
    kb = tb;
    if (PAIRS) {
        const uint32_t ua = sw ? vb : va, ub = sw ? va : vb;
        va = ua;
        vb = ub;
    }
}
__device__ __forceinline__ uint32_t s3_lane_xor(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }
__device__ __forceinline__ unsigned long long s3_lane_xor(unsigned long long v, int m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}
template <typename K, int PER, bool PAIRS>
__device__ __forceinline__ void s3_block_sort(K (&k)[PER], uint32_t (&v)[PER], K *__restrict__ xk, uint32_t *__restrict__ xv) {
    constexpr int N = S3_SORT_THREADS * PER;
    const int t = threadIdx.x;
    // phases 2 .. PER: all inside the thread
#pragma unroll
    for (int kk = 2; kk <= PER; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int a = 0; a < PER; a++) {
                if (a & j) continue;
                const bool up = (kk < PER) ? ((a & kk) == 0) : ((t & 1) == 0);
                s3_cmpx<K, PAIRS>(k[a], v[a], k[a | j], v[a | j], up);
            }
        }
    }
    // phases 2 PER .. N: distances >= PER first (other threads), then the tail inside the thread
#pragma unroll 1
    for (int kt = 2; kt <= S3_SORT_THREADS; kt <<= 1) {
        const bool up = (t & kt) == 0;
#pragma unroll 1
        for (int pt = kt >> 1; pt >= 1; pt >>= 1) {
            const bool keep_min = ((t & pt) == 0) == up;
            if (pt >= 64) {          // (workgroup-uniform)
#pragma unroll
                for (int a = 0; a < PER; a++) {
                    xk[a * S3_SORT_THREADS + t] = k[a];
                    if (PAIRS) xv[a * S3_SORT_THREADS + t] = v[a];
                }
                __syncthreads();
#pragma unroll
                for (int a = 0; a < PER; a++) {
                    const K ok = xk[a * S3_SORT_THREADS + (t ^ pt)];
                    const uint32_t ov = PAIRS ? xv[a * S3_SORT_THREADS + (t ^ pt)] : 0u;
                    const bool take = keep_min ? s3_kv_less<K, PAIRS>(ok, ov, k[a], v[a]) : s3_kv_less<K, PAIRS>(k[a], v[a], ok, ov);
                    k[a] = take ? ok : k[a];
                    if (PAIRS) v[a] = take ? ov : v[a];
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int a = 0; a < PER; a++) {
                    const K ok = s3_lane_xor(k[a], pt);
                    const uint32_t ov = PAIRS ? s3_lane_xor(v[a], pt) : 0u;
                    const bool take = keep_min ? s3_kv_less<K, PAIRS>(ok, ov, k[a], v[a]) : s3_kv_less<K, PAIRS>(k[a], v[a], ok, ov);
                    k[a] = take ? ok : k[a];
                    if (PAIRS) v[a] = take ? ov : v[a];
                }
            }
        }
#pragma unroll
        for (int j = PER >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int a = 0; a < PER; a++) {
                if (a & j) continue;
                s3_cmpx<K, PAIRS>(k[a], v[a], k[a | j], v[a | j], up);
            }
        }
    }
    (void)N;
}

// the value lane ^ PT holds, without the LDS crossbar: DPP row operations for distances 1, 2, 4, 8 (4 = a shift by four
// lanes in each direction, each written to the banks it is right for), the gfx950 row / half swaps for 16 and 32
template <int PT>
__device__ __forceinline__ uint32_t s3_partner32(uint32_t v) {
    if constexpr (PT == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);          // quad_perm [1,0,3,2]
    else if constexpr (PT == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
    else if constexpr (PT == 4) {
        const int a = __builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xA, false);                               // row_shr:4 -> banks 1, 3
        return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)v, 0x104, 0xF, 0x5, false);                           // row_shl:4 -> banks 0, 2
    } else {
        static_assert(PT == 8, "DPP lane distance");
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);                            // row_ror:8
    }
}
template <int PT> __device__ __forceinline__ uint32_t s3_partner(uint32_t v) { return s3_partner32<PT>(v); }
template <int PT> __device__ __forceinline__ unsigned long long s3_partner(unsigned long long v) {
    return ((unsigned long long)s3_partner32<PT>((uint32_t)(v >> 32)) << 32) | s3_partner32<PT>((uint32_t)v);
}
// distances 16 and 32: both ends of every pair side by side after one swap (r0: the low-side lane's key, r1: the
// high-side lane's, on both lanes)
template <int PT>
__device__ __forceinline__ void s3_pair_sides(uint32_t v, uint32_t &r0, uint32_t &r1) {
    if constexpr (PT == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);    // rows 0,0,2,2 | rows 1,1,3,3
        r0 = r[0];
        r1 = r[1];
    } else {
        static_assert(PT == 32, "swap lane distance");
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);    // low half twice | high half twice
        r0 = r[0];
        r1 = r[1];
    }
}
template <int PT>
__device__ __forceinline__ void s3_pair_sides(unsigned long long v, unsigned long long &r0, unsigned long long &r1) {
    uint32_t a0, a1, b0, b1;
    s3_pair_sides<PT>((uint32_t)v, a0, a1);
    s3_pair_sides<PT>((uint32_t)(v >> 32), b0, b1);
    r0 = ((unsigned long long)b0 << 32) | a0;
    r1 = ((unsigned long long)b1 << 32) | a1;
}
// one compare-exchange step at lane distance PT: the low-side lane keeps the smaller key (ties: either)
template <typename K, int PER, int PT>
__device__ __forceinline__ void s3_lane_step(K (&k)[PER]) {
    const bool low = (threadIdx.x & PT) == 0;
#pragma unroll
    for (int a = 0; a < PER; a++) {
        if constexpr (PT >= 16) {
            K r0, r1;
            s3_pair_sides<PT>(k[a], r0, r1);
            const K lo = r0 < r1 ? r0 : r1, hi = r0 < r1 ? r1 : r0;
            k[a] = low ? lo : hi;
        } else {
            // (min, max, select: measured faster than compare, flip the mask by the lane's side, select)
            const K o = s3_partner<PT>(k[a]);
            const K lo = k[a] < o ? k[a] : o, hi = k[a] < o ? o : k[a];
            k[a] = low ? lo : hi;
        }
    }
}
template <typename K, int PER>
__device__ __forceinline__ void s3_wave_step(K (&k)[PER], K *__restrict__ xk, int pt) {
    const int t = threadIdx.x;
    const bool low = (t & pt) == 0;
#pragma unroll
    for (int a = 0; a < PER; a++) xk[a * S3_SORT_THREADS + t] = k[a];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < PER; a++) {
        const K o = xk[a * S3_SORT_THREADS + (t ^ pt)];
        const K lo = k[a] < o ? k[a] : o, hi = k[a] < o ? o : k[a];
        k[a] = low ? lo : hi;
    }
    __syncthreads();
}

// keys only: a thread's direction in a phase is one bit of t, the same for both ends of every compare-exchange of
// that phase, so keys of a descending thread are kept COMPLEMENTED for the phase and every step is an ascending one:
// min / max in registers, min-or-max by the lane's side of the exchange across lanes.
template <typename K, int PER>
__device__ __forceinline__ void s3_block_sort_keys(K (&k)[PER], K *__restrict__ xk) {
    const int t = threadIdx.x;
#pragma unroll
    for (int kk = 2; kk < PER; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int a = 0; a < PER; a++) {
                if (a & j) continue;
                const K lo = k[a] < k[a | j] ? k[a] : k[a | j], hi = k[a] < k[a | j] ? k[a | j] : k[a];
                k[a] = (a & kk) ? hi : lo;
                k[a | j] = (a & kk) ? lo : hi;
            }
        }
    }
    K flip = 0;
#pragma unroll 1
    for (int kt = (PER > 1 ? 1 : 2); kt <= S3_SORT_THREADS; kt <<= 1) {
        {
            const K m = (t & kt) ? (K)~(K)0 : (K)0, d = m ^ flip;
#pragma unroll
            for (int a = 0; a < PER; a++) k[a] ^= d;
            flip = m;
        }
        // (kt is workgroup-uniform: one copy of each step's code, the phase picks where it enters the sequence)
#pragma unroll 1
        for (int pt = kt >> 1; pt >= 64; pt >>= 1) s3_wave_step<K, PER>(k, xk, pt);
        if (kt > 32) s3_lane_step<K, PER, 32>(k);
        if (kt > 16) s3_lane_step<K, PER, 16>(k);
        if (kt > 8) s3_lane_step<K, PER, 8>(k);
        if (kt > 4) s3_lane_step<K, PER, 4>(k);
        if (kt > 2) s3_lane_step<K, PER, 2>(k);
        if (kt > 1) s3_lane_step<K, PER, 1>(k);
#pragma unroll
        for (int j = PER >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int a = 0; a < PER; a++) {
                if (a & j) continue;
                const K lo = k[a] < k[a | j] ? k[a] : k[a | j], hi = k[a] < k[a | j] ? k[a | j] : k[a];
                k[a] = lo;
                k[a | j] = hi;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < PER; a++) k[a] ^= flip;      // (the last phase is ascending everywhere: flip is 0 already)
}

// sort + run-length encode + keep of one segment of <= S3_SORT_THREADS * PER residuals (all threads of the
// block); appends at out[0...], returns the number kept; lsum accumulates the kept counts (per thread)
template <typename KR2, int PER>
__device__ __forceinline__ uint32_t s3_sort_segment(const KR2 *__restrict__ seg, uint32_t n, int bits, uint32_t lower,
                                                    s3_sort_lds<KR2, PER> &L, KR2 *__restrict__ out_keys,
                                                    uint32_t *__restrict__ out_cnts, unsigned long long &lsum) {
    KR2 items[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t i = threadIdx.x * PER + j;
        items[j] = (i < n) ? seg[i] : (KR2)~(KR2)0;   // padding sorts last (stable: after equal valid keys)
    }
    __syncthreads();
    s3_block_sort_keys<KR2, PER>(items, L.s);     // (full-width compares: the bits above `bits` are equal inside a segment)
    (void)bits;
#pragma unroll
    for (int j = 0; j < PER; j++) L.s[threadIdx.x * PER + j] = items[j];
    __syncthreads();
    uint32_t nh = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t i = threadIdx.x * PER + j;
        if (i < n && (i == 0 || L.s[i] != L.s[i - 1])) nh++;
    }
    L.cnt1[threadIdx.x] = nh;
    __syncthreads();
    const uint32_t n_runs = s3_block_scan<S3_SORT_THREADS>(L.cnt1, L.cnt2, S3_SORT_THREADS, L.wsum);
    {
        uint32_t r = L.cnt2[threadIdx.x];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t i = threadIdx.x * PER + j;
            if (i < n && (i == 0 || L.s[i] != L.s[i - 1])) L.heads[r++] = i;
        }
        if (threadIdx.x == 0) L.heads[n_runs] = n;
    }
    __syncthreads();
    const uint32_t per = (n_runs + S3_SORT_THREADS - 1) / S3_SORT_THREADS;
    const uint32_t r0 = threadIdx.x * per, r1 = (r0 + per < n_runs) ? r0 + per : n_runs;
    uint32_t nk = 0;
    for (uint32_t r = r0; r < r1; r++) nk += (L.heads[r + 1] - L.heads[r]) >= lower;
    L.cnt1[threadIdx.x] = nk;
    __syncthreads();
    const uint32_t n_kept = s3_block_scan<S3_SORT_THREADS>(L.cnt1, L.cnt2, S3_SORT_THREADS, L.wsum);
    uint32_t w = L.cnt2[threadIdx.x];
    for (uint32_t r = r0; r < r1; r++) {
        const uint32_t len = L.heads[r + 1] - L.heads[r];
        if (len >= lower) {
            out_keys[w] = L.s[L.heads[r]];
            out_cnts[w] = len;
            lsum += len;
            w++;
        }
    }
    __syncthreads();
    return n_kept;
}

// light kernel: buckets of <= 1024 residuals (4 per thread, ~14 KB of LDS: many blocks per CU)
template <typename KR2>
__global__ void __launch_bounds__(S3_SORT_THREADS)
s3_final_small(const KR2 *__restrict__ buf2, const ulonglong2 *__restrict__ span, int64_t n_fine, int R2,
               uint32_t lower, KR2 *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_cnts,
               unsigned long long *__restrict__ kept, unsigned long long *__restrict__ len_sum) {
    // size classes: a 200-key bucket must not pay for a padded 2048-key sort
    __shared__ union {
        s3_sort_lds<KR2, 1> l1;
        s3_sort_lds<KR2, 2> l2;
        s3_sort_lds<KR2, 4> l4;
        s3_sort_lds<KR2, S3_SMALL_PER> l8;
    } L;
    __shared__ unsigned long long red[16];
    unsigned long long lsum = 0;
    for (int64_t bucket = blockIdx.x; bucket < n_fine; bucket += gridDim.x) {
        const ulonglong2 sp_ = span[bucket];
        const unsigned long long o = sp_.x, n64 = sp_.y;
        if (n64 > S3_SMALL_CAP) continue;            // the heavy kernel's
        if (n64 == 0) {
            if (threadIdx.x == 0) kept[bucket] = 0;
            continue;
        }
        const uint32_t n = (uint32_t)n64;
        uint32_t nk;
        if (n <= S3_SORT_THREADS)
            nk = s3_sort_segment<KR2, 1>(buf2 + o, n, R2, lower, L.l1, tmp_keys + o, tmp_cnts + o, lsum);
        else if (n <= 2 * S3_SORT_THREADS)
            nk = s3_sort_segment<KR2, 2>(buf2 + o, n, R2, lower, L.l2, tmp_keys + o, tmp_cnts + o, lsum);
        else if (n <= 4 * S3_SORT_THREADS)
            nk = s3_sort_segment<KR2, 4>(buf2 + o, n, R2, lower, L.l4, tmp_keys + o, tmp_cnts + o, lsum);
        else
            nk = s3_sort_segment<KR2, S3_SMALL_PER>(buf2 + o, n, R2, lower, L.l8, tmp_keys + o, tmp_cnts + o, lsum);
        if (threadIdx.x == 0) kept[bucket] = nk;
    }
    const unsigned long long tot = sp_block_sum_u64(lsum, red);
    if (threadIdx.x == 0 && tot) atomicAdd(len_sum, tot);
}

// ---------------------------------------------------------------- s3_final_bitmap (k = 16, 17)
// With <= 16 residual bits a fine bucket is a sparse set over <= 65536 values (~1.3 K keys for a wheat-sized
// chromosome at k = 17), and sorting is more than the job needs: (1) every key sets its bit in an LDS bitmap,
// (2) a prefix popcount over the bitmap words turns "bit position" into "rank among the distinct residuals",
// (3) every key adds one to cnt[rank], (4) each thread walks the set bits of ITS words in ascending order and
// writes the (residual, count) pairs that pass `lower` at the offset an ordered block scan gives it.  Ascending
// output without a sort: ~6 LDS operations per key and two block scans per bucket instead of four radix passes
// (s3_final_small + s3_final were 200 of the 507 ms of a wheat-like pass at k = 17; this kernel + the rest of
// s3_final: 86 ms.  PMC: VALU 43 % and LDS 45 % busy, half of the LDS cycles bank conflicts of the random bitmap /
// counter accesses; a variant whose first-arriving copy writes the pair out instead of the bit walk was no faster).
#define S3_BM_MAXBITS 16
#ifndef S3_BM_CAP
#define S3_BM_CAP 4096      // distinct residuals per bucket this kernel takes (cnt[] entries); beyond: s3_final
#endif
#define S3_BM_NMAX (1u << 22)   // keys per bucket it is willing to stream (hot buckets: few distinct, many copies)
#define S3_BM_PER 8             // keys per thread prefetched into registers (buckets up to 2048 keys never re-read)
#define S3_BM_PUNT (~0ULL)      // kept[bucket] value that hands the bucket to s3_final
// ... and the bucket goes on the punt list s3_final walks (thread 0 of the workgroup; round 3: s3_final used to look at
// kept[] of all 2^19 buckets, 32 per workgroup with two dependent loads each -- most of its time)
__device__ __forceinline__ void s3_punt(unsigned long long *__restrict__ kept, int64_t bucket, uint32_t *__restrict__ punt_list,
                                        unsigned long long *__restrict__ n_punt) {
    kept[bucket] = S3_BM_PUNT;
    punt_list[atomicAdd(n_punt, 1ULL)] = (uint32_t)bucket;
}

// exclusive prefix of v over the block (value in a register); *total = block sum.  Ends with a barrier.
__device__ __forceinline__ uint32_t s3_scan_reg(uint32_t v, uint32_t *wsum /* >= THREADS/64 */, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < S3_SORT_THREADS / 64; w++) {
        const uint32_t x = wsum[w];
        if (w < wave) base += x;
        tot += x;
    }
    *total = tot;
    return base + incl - v;
}

#ifndef S3_BM_MINW
#define S3_BM_MINW 1      // waves per SIMD the register allocation of s3_final_bitmap must leave room for
#endif
template <typename KR2>
__global__ void __launch_bounds__(S3_SORT_THREADS, S3_BM_MINW)
s3_final_bitmap(const KR2 *__restrict__ buf2, const ulonglong2 *__restrict__ span, int64_t n_fine, int R2,
                uint32_t lower, KR2 *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_cnts,
                unsigned long long *__restrict__ kept, unsigned long long *__restrict__ len_sum,
                uint32_t *__restrict__ punt_list, unsigned long long *__restrict__ n_punt) {
    extern __shared__ __attribute__((aligned(16))) uint32_t bm_lds[];   // bm[nw] | cnt[S3_BM_CAP] | pre[nw] (u16)
    __shared__ uint32_t ws1[S3_SORT_THREADS / 64], ws2[S3_SORT_THREADS / 64];
    __shared__ unsigned long long red[16];
    const int nw = R2 > 5 ? 1 << (R2 - 5) : 1;
    uint32_t *bm = bm_lds, *cnt = bm_lds + nw;
    uint16_t *pre = reinterpret_cast<uint16_t *>(cnt + S3_BM_CAP);
    const int wpt = (nw + S3_SORT_THREADS - 1) / S3_SORT_THREADS;
    const int w0 = (int)threadIdx.x * wpt, w1 = w0 + wpt < nw ? w0 + wpt : nw;
    unsigned long long lsum = 0;
    // software pipeline: offsets and the first S3_BM_PER x 256 keys of the NEXT bucket travel while this one is
    // processed (a bucket is ~7 barrier-separated steps of a few hundred cycles; two exposed global round trips on
    // top of them made the first version of this kernel 40 % slower)
    unsigned long long p_o = 0, p_n = 0;
    uint32_t p_key[S3_BM_PER];
    auto fetch = [&](int64_t b) {
        p_n = 0;
        if (b >= n_fine) return;
        const ulonglong2 sp_ = span[b];
        p_o = sp_.x;
        p_n = sp_.y;
        if (p_n > S3_BM_NMAX) return;
#pragma unroll
        for (int j = 0; j < S3_BM_PER; j++) {
            const uint32_t i = threadIdx.x + (uint32_t)j * S3_SORT_THREADS;
            if (i < p_n) p_key[j] = (uint32_t)buf2[p_o + i];
        }
    };
    fetch(blockIdx.x);
    for (int64_t bucket = blockIdx.x; bucket < n_fine; bucket += gridDim.x) {
        const unsigned long long o = p_o, n64 = p_n;
        uint32_t key[S3_BM_PER];
#pragma unroll
        for (int j = 0; j < S3_BM_PER; j++) key[j] = p_key[j];
        fetch(bucket + gridDim.x);
        if (n64 == 0 || n64 > S3_BM_NMAX) {
            if (threadIdx.x == 0) {
                if (n64) s3_punt(kept, bucket, punt_list, n_punt);
                else kept[bucket] = 0ULL;
            }
            continue;
        }
        const uint32_t n = (uint32_t)n64;
        const KR2 *seg = buf2 + o;
        for (int i = threadIdx.x; i < nw; i += S3_SORT_THREADS) bm[i] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < S3_BM_PER; j++)
            if (threadIdx.x + (uint32_t)j * S3_SORT_THREADS < n) atomicOr(&bm[key[j] >> 5], 1u << (key[j] & 31));
        for (uint32_t i = threadIdx.x + S3_BM_PER * S3_SORT_THREADS; i < n; i += S3_SORT_THREADS) {
            const uint32_t r = (uint32_t)seg[i];
            atomicOr(&bm[r >> 5], 1u << (r & 31));
        }
        __syncthreads();
        uint32_t mine = 0;
        for (int w = w0; w < w1; w++) mine += (uint32_t)__popc(bm[w]);
        uint32_t distinct;
        const uint32_t first = s3_scan_reg(mine, ws1, &distinct);   // rank of this thread's first distinct residual
        if (distinct > S3_BM_CAP) {                                 // block-uniform
            if (threadIdx.x == 0) s3_punt(kept, bucket, punt_list, n_punt);
            __syncthreads();
            continue;
        }
        {
            uint32_t run = first;
            for (int w = w0; w < w1; w++) {
                pre[w] = (uint16_t)run;
                run += (uint32_t)__popc(bm[w]);
            }
        }
        for (uint32_t i = threadIdx.x; i < distinct; i += S3_SORT_THREADS) cnt[i] = 0;
        __syncthreads();
        auto bump = [&](uint32_t r) {
            atomicAdd(&cnt[(uint32_t)pre[r >> 5] + (uint32_t)__popc(bm[r >> 5] & ((1u << (r & 31)) - 1u))], 1u);
        };
#pragma unroll
        for (int j = 0; j < S3_BM_PER; j++)
            if (threadIdx.x + (uint32_t)j * S3_SORT_THREADS < n) bump(key[j]);
        for (uint32_t i = threadIdx.x + S3_BM_PER * S3_SORT_THREADS; i < n; i += S3_SORT_THREADS) bump((uint32_t)seg[i]);
        __syncthreads();
        uint32_t nk = 0;
        for (uint32_t q = first; q < first + mine; q++) nk += cnt[q] >= lower;
        uint32_t n_kept;
        uint32_t wpos = s3_scan_reg(nk, ws2, &n_kept);
        if (nk) {
            uint32_t q = first;
            for (int w = w0; w < w1; w++) {
                uint32_t bits = bm[w];
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1u;
                    const uint32_t c = cnt[q++];
                    if (c >= lower) {
                        tmp_keys[o + wpos] = (KR2)(((uint32_t)w << 5) | (uint32_t)b);
                        tmp_cnts[o + wpos] = c;
                        lsum += c;
                        wpos++;
                    }
                }
            }
        }
        if (threadIdx.x == 0) kept[bucket] = n_kept;
        __syncthreads();   // bm / cnt / pre are rewritten at the top of the next bucket
    }
    const unsigned long long tot = sp_block_sum_u64(lsum, red);
    if (threadIdx.x == 0 && tot) atomicAdd(len_sum, tot);
}

// ---------------------------------------------------------------- s3_final_hash (k >= 18)
// Residuals wider than 16 bits cannot take the bitmap finish, and round 2 left the job to a register-resident block
// radix sort (a library block primitive inside s3_final_small / s3_final: 254 of 476 ms per wheat-like pass at
// k = 21).  But the OUTPUT of a fine bucket is tiny: only residuals with count >= lower_count survive, and on a real
// genome most k-mers of a bucket occur once.  So nothing is sorted that does not have to be:
//   stage 1  every key bumps one of 4096 LDS counters chosen by a hash (a counting Bloom filter with one hash): a key
//            whose counter stays below lower_count cannot have lower_count copies -- it is dropped, unsorted, uncounted;
//   stage 2  the survivors (true repeats + a few per cent of colliding singletons) are counted exactly in a small
//            open-addressing table that reuses the counters' LDS;
//   emit     table entries with count >= lower_count -- a few dozen per bucket -- are ranked by residual among
//            themselves (quadratic in THEIR number) and written in ascending order.
// Buckets above 2048 keys (repeat families) skip stage 1 and stream straight into the table: many copies of few
// residuals.  A bucket with more distinct survivors than the table takes (S3H_CAP) is handed to s3_final
// (kept[] = S3_BM_PUNT), which still sorts.

// exact count of one key in the table; the table closes (abort_) when it has taken S3H_CAP distinct residuals
template <typename KR2>
__device__ __forceinline__ void s3h_insert(s3h_lds<KR2> &L, KR2 key) {
    const KR2 EMPTY = (KR2)~(KR2)0;
    uint32_t h = s3h_hash2((unsigned long long)key) & (S3H_SLOTS - 1);
    // (bounded: with batched loads more keys than the table has free slots can be between two abort tests, and an
    // unbounded probe loop over a full table never ends -- r03_notes.md; a closed table is nearly full: do not walk it)
    for (int probe = 0; probe < S3H_SLOTS; probe++) {
        if ((probe & 15) == 15 && L.abort_) return;
        const KR2 prev = atomicCAS(&L.u.t.key[h], EMPTY, key);
        if (prev == key) {
            atomicAdd(&L.u.t.cnt[h], 1u);
            return;
        }
        if (prev == EMPTY) {
            if (atomicAdd(&L.n_distinct, 1u) >= S3H_CAP) L.abort_ = 1;
            atomicAdd(&L.u.t.cnt[h], 1u);
            return;
        }
        h = (h + 1) & (S3H_SLOTS - 1);
    }
    L.abort_ = 1;       // the table is full
}
// the kept entries of the table: a short list, ranked by residual, written at out_*[0 ..); returns their number
template <typename KR2>
__device__ __forceinline__ uint32_t s3h_emit(s3h_lds<KR2> &L, uint32_t lower, KR2 *__restrict__ out_keys,
                                             uint32_t *__restrict__ out_cnts, unsigned long long &lsum) {
    const int tid = threadIdx.x;
    for (int i4 = tid; i4 < S3H_SLOTS / 4; i4 += S3_SORT_THREADS) {
        const uint4 c4 = reinterpret_cast<const uint4 *>(L.u.t.cnt)[i4];
        const uint32_t cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t c = cc[q];
            if (c >= lower) {
                const uint32_t at = atomicAdd(&L.n_kept, 1u);
                L.lk[at] = L.u.t.key[4 * i4 + q];
                lsum += c;
            }
        }
    }
    __syncthreads();
    const uint32_t m = L.n_kept;
    for (uint32_t i = tid; i < m; i += S3_SORT_THREADS) {
        const KR2 key = L.lk[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; j++) rank += L.lk[j] < key;
        out_keys[rank] = key;
        uint32_t h = s3h_hash2((unsigned long long)key) & (S3H_SLOTS - 1);
        while (L.u.t.key[h] != key) h = (h + 1) & (S3H_SLOTS - 1);      // (it is there)
        out_cnts[rank] = L.u.t.cnt[h];
    }
    return m;
}
__device__ __forceinline__ void s3h_clear_table_words(uint4 *zk, int nk16, uint4 *zc, int nc16) {
    for (int i = threadIdx.x; i < nk16; i += S3_SORT_THREADS) zk[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (int i = threadIdx.x; i < nc16; i += S3_SORT_THREADS) zc[i] = make_uint4(0, 0, 0, 0);
}
// One segment of at most S3_SORT_THREADS * S3H_PER residuals through both stages (all threads of the workgroup).
// Returns the number of kept (residual, count) pairs written at out_*[0 ..) in ascending order, or ~0u when more
// distinct survivors turn up than the table takes (the caller sorts the segment instead; lsum is untouched then).
template <typename KR2>
__device__ __forceinline__ uint32_t s3h_segment(const KR2 *__restrict__ seg, uint32_t n, uint32_t lower, s3h_lds<KR2> &L,
                                                KR2 *__restrict__ out_keys, uint32_t *__restrict__ out_cnts,
                                                unsigned long long &lsum) {
    const int tid = threadIdx.x;
    KR2 mine[S3H_PER];
    uint32_t alive = 0;      // bit j: mine[j] survived stage 1
    __syncthreads();         // the previous readers of L are done
    if (tid == 0) L.n_distinct = L.n_kept = L.abort_ = 0;
#pragma unroll
    for (int j = 0; j < S3H_PER; j++) {
        const uint32_t i = tid + j * S3_SORT_THREADS;
        if (i < n) mine[j] = seg[i];
    }
    {       // 16-byte LDS stores: a quarter of the instructions (the kernel is LDS-issue-bound, not latency-bound)
        uint4 *z = reinterpret_cast<uint4 *>(L.u.c1);
        for (int i = tid; i < S3H_COUNTERS / 4; i += S3_SORT_THREADS) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < S3H_PER; j++)
        if (tid + j * S3_SORT_THREADS < n) atomicAdd(&L.u.c1[s3_hash((unsigned long long)mine[j]) & (S3H_COUNTERS - 1)], 1u);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < S3H_PER; j++)
        if (tid + j * S3_SORT_THREADS < n && L.u.c1[s3_hash((unsigned long long)mine[j]) & (S3H_COUNTERS - 1)] >= lower)
            alive |= 1u << j;
    __syncthreads();     // every flag is taken: the counters' memory becomes the table
    s3h_clear_table_words(reinterpret_cast<uint4 *>(L.u.t.key), (int)(S3H_SLOTS * sizeof(KR2) / 16),
                          reinterpret_cast<uint4 *>(L.u.t.cnt), S3H_SLOTS / 4);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < S3H_PER; j++)
        if ((alive >> j) & 1u) {
            if (L.abort_) break;
            s3h_insert(L, mine[j]);
        }
    __syncthreads();
    if (L.abort_) return ~0u;          // block-uniform after the barrier
    return s3h_emit(L, lower, out_keys, out_cnts, lsum);
}

template <typename KR2>
__global__ void __launch_bounds__(S3_SORT_THREADS)
s3_final_hash(const KR2 *__restrict__ buf2, const ulonglong2 *__restrict__ span, int64_t n_fine, int R2,
              uint32_t lower, KR2 *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_cnts,
              unsigned long long *__restrict__ kept, unsigned long long *__restrict__ len_sum,
              uint32_t *__restrict__ punt_list, unsigned long long *__restrict__ n_punt) {
    __shared__ s3h_lds<KR2> L;
    const int tid = threadIdx.x;
    unsigned long long lsum = 0;
    for (int64_t bucket = blockIdx.x; bucket < n_fine; bucket += gridDim.x) {
        const ulonglong2 sp_ = span[bucket];
        const unsigned long long o = sp_.x, n64 = sp_.y;
        if (n64 == 0 || n64 > S3_BM_NMAX) {
            if (tid == 0) {
                if (n64) s3_punt(kept, bucket, punt_list, n_punt);
                else kept[bucket] = 0ULL;
            }
            continue;
        }
        const uint32_t n = (uint32_t)n64;
        const KR2 *seg = buf2 + o;
        uint32_t m;
        if (n <= S3_SORT_THREADS * S3H_PER) {      // block-uniform: both stages, the keys in registers
            m = s3h_segment<KR2>(seg, n, lower, L, tmp_keys + o, tmp_cnts + o, lsum);
        } else {                                   // a repeat family: many copies of few residuals, straight into the table
            __syncthreads();
            if (tid == 0) L.n_distinct = L.n_kept = L.abort_ = 0;
            s3h_clear_table_words(reinterpret_cast<uint4 *>(L.u.t.key), (int)(S3H_SLOTS * sizeof(KR2) / 16),
                                  reinterpret_cast<uint4 *>(L.u.t.cnt), S3H_SLOTS / 4);
            __syncthreads();
            // eight loads in flight per thread (2.55 -> 2.24 ms per 670-Mb chromosome for these buckets: most of their
            // time is not the loads but same-address LDS atomics -- half of their keys are copies of a few dozen residuals)
            for (uint32_t i0 = tid; i0 < n; i0 += 8 * S3_SORT_THREADS) {
                KR2 v[8];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (i0 + q * S3_SORT_THREADS < n) v[q] = seg[i0 + q * S3_SORT_THREADS];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (i0 + q * S3_SORT_THREADS < n && !L.abort_) s3h_insert(L, v[q]);
                if (L.abort_) break;
            }
            __syncthreads();
            m = L.abort_ ? ~0u : s3h_emit(L, lower, tmp_keys + o, tmp_cnts + o, lsum);
        }
        if (tid == 0) {
            if (m == ~0u) s3_punt(kept, bucket, punt_list, n_punt);
            else kept[bucket] = (unsigned long long)m;
        }
    }
    __syncthreads();
    const unsigned long long tot = sp_block_sum_u64(lsum, L.red);
    if (tid == 0 && tot) atomicAdd(len_sum, tot);
}

#ifndef S3_FINAL_MINW
#define S3_FINAL_MINW 3     // 32-bit residuals: <= 168 VGPRs, three workgroups per CU as the 49 KB of LDS allow (it took
                            // 181 and ran two: 1.11 -> 0.97 ms at k = 21, 8.7 -> 6.4 ms with SP_S3_FINAL=sort at k = 19);
                            // 64-bit residuals: 73 KB of LDS, two either way
#endif
#define S3_FINAL_BOUNDS __launch_bounds__(S3_SORT_THREADS, (sizeof(KR2) == 4 ? S3_FINAL_MINW : 2))
template <typename KR2>
__global__ void S3_FINAL_BOUNDS
s3_final(const KR2 *__restrict__ buf2, KR2 *__restrict__ scratch, const ulonglong2 *__restrict__ span,
         int64_t n_fine, int R2, uint32_t lower, KR2 *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_cnts,
         unsigned long long *__restrict__ kept, unsigned long long *__restrict__ big_list,
         unsigned long long *__restrict__ n_big, unsigned long long big_cap, unsigned long long *__restrict__ len_sum,
         unsigned long long light_cap /* buckets up to this size were the light kernel's; S3_BM_PUNT: the ones
                                         the bitmap kernel marked in kept[] */,
         int hash_pieces /* k >= 18: count the pieces of a split bucket like s3_final_hash does, sort only on abort */,
         const uint32_t *__restrict__ punt_list, const unsigned long long *__restrict__ n_punt,
         int force_big /* test hook: every bucket of more than one sort's worth of keys goes to the oversized path */) {
    __shared__ s3_final_lds<KR2> L;
    unsigned long long lsum = 0;
    // punt mode: the listed buckets; else (the round-2 sort kernels, SP_S3_FINAL=sort) every bucket above light_cap
    const bool listed = light_cap == S3_BM_PUNT;
    const int64_t n_work = listed ? (int64_t)*n_punt : n_fine;
    for (int64_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const int64_t bucket = listed ? (int64_t)punt_list[wi] : wi;
        const ulonglong2 sp_ = span[bucket];
        const unsigned long long o = sp_.x, n64 = sp_.y;
        if (!listed && n64 <= light_cap) continue;   // the light kernel's
        if (n64 <= S3_SORT_CAP) {
            // (a sorting network's work grows with the padded size: size classes as in the light kernel)
            uint32_t nk;
            if (n64 <= 2 * S3_SORT_THREADS)
                nk = s3_sort_segment<KR2, 2>(buf2 + o, (uint32_t)n64, R2, lower, L.q2, tmp_keys + o, tmp_cnts + o, lsum);
            else if (n64 <= 4 * S3_SORT_THREADS)
                nk = s3_sort_segment<KR2, 4>(buf2 + o, (uint32_t)n64, R2, lower, L.q4, tmp_keys + o, tmp_cnts + o, lsum);
            else
                nk = s3_sort_segment<KR2, S3_SORT_PER>(buf2 + o, (uint32_t)n64, R2, lower, L.q, tmp_keys + o, tmp_cnts + o, lsum);
            if (threadIdx.x == 0) kept[bucket] = nk;
            continue;
        }
        bool give_up = n64 >= (1ULL << 31) || force_big;
        uint32_t w = 0;                      // kept pairs written so far (block-uniform)
        unsigned long long bsum = 0;         // their counts; dropped if the bucket goes to the fallback
        if (!give_up) {
            const uint32_t n = (uint32_t)n64;
            int sb = 1;                      // pieces of ~2 K residuals: a few sorts, not 2^10 tiny ones
            while (sb < S3_SPLIT_BITS && (n >> sb) > S3_SORT_CAP / 2) sb++;
            if (sb > R2) sb = R2;
            const int rem = R2 - sb, nsub = 1 << sb;
            const KR2 rem_mask = (rem >= (int)(8 * sizeof(KR2))) ? (KR2)~(KR2)0 : (KR2)(((KR2)1 << rem) - 1);
            for (int i = threadIdx.x; i < nsub; i += S3_SORT_THREADS) L.sub_cnt[i] = 0;
            __syncthreads();
            // (eight loads in flight per thread: with one, a 64 K-key bucket was 256 serialized round trips per pass and
            // the split alone 18 of this kernel's 37 ms per wheat-like pass at k = 21)
            for (uint32_t i0 = threadIdx.x; i0 < n; i0 += 8 * S3_SORT_THREADS) {
                KR2 v[8];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (i0 + q * S3_SORT_THREADS < n) v[q] = buf2[o + i0 + q * S3_SORT_THREADS];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (i0 + q * S3_SORT_THREADS < n) atomicAdd(&L.sub_cnt[(uint32_t)(v[q] >> rem)], 1u);
            }
            __syncthreads();
            const uint32_t tot = s3_block_scan<S3_SORT_THREADS>(L.sub_cnt, L.sub_off, nsub, L.q.wsum);
            if (threadIdx.x == 0) L.sub_off[nsub] = tot;
            for (int i = threadIdx.x; i < nsub; i += S3_SORT_THREADS) L.sub_cnt[i] = 0;
            __syncthreads();
            for (uint32_t i0 = threadIdx.x; i0 < n; i0 += 8 * S3_SORT_THREADS) {
                KR2 v[8];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (i0 + q * S3_SORT_THREADS < n) v[q] = buf2[o + i0 + q * S3_SORT_THREADS];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (i0 + q * S3_SORT_THREADS < n) {
                        const uint32_t d = (uint32_t)(v[q] >> rem);
                        scratch[o + L.sub_off[d] + atomicAdd(&L.sub_cnt[d], 1u)] = v[q];
                    }
            }
            __threadfence_block();
            __syncthreads();
            for (int d = 0; d < nsub && !give_up; d++) {
                const uint32_t so = L.sub_off[d], m = L.sub_off[d + 1] - so;   // block-uniform
                if (m == 0) continue;
                const KR2 *seg = scratch + o + so;
                const KR2 hi = (rem >= (int)(8 * sizeof(KR2))) ? (KR2)0 : (KR2)((KR2)d << rem);
                uint32_t nkh = ~0u;
                if (hash_pieces && m <= S3_SORT_THREADS * S3H_PER) {
                    // round 3: a 64 K-key bucket was ~30 block radix sorts in a row inside ONE workgroup -- the long pole
                    // of this kernel; its pieces go through the hashed pre-count + exact table instead
                    nkh = s3h_segment<KR2>(seg, m, lower, L.h, tmp_keys + o + w, tmp_cnts + o + w, bsum);
                    __syncthreads();
                }
                if (nkh != ~0u) {
                    w += nkh;
                } else if (m <= S3_SORT_CAP) {
                    // the pieces hold full residuals; sort on the remaining bits only
                    if (m <= 2 * S3_SORT_THREADS)
                        w += s3_sort_segment<KR2, 2>(seg, m, rem, lower, L.q2, tmp_keys + o + w, tmp_cnts + o + w, bsum);
                    else if (m <= 4 * S3_SORT_THREADS)
                        w += s3_sort_segment<KR2, 4>(seg, m, rem, lower, L.q4, tmp_keys + o + w, tmp_cnts + o + w, bsum);
                    else
                        w += s3_sort_segment<KR2, S3_SORT_PER>(seg, m, rem, lower, L.q, tmp_keys + o + w, tmp_cnts + o + w, bsum);
                } else if (rem <= S3_DIRECT_BITS) {          // few possible values: count them all in LDS
                    const uint32_t nv = 1u << rem;
                    for (uint32_t i = threadIdx.x; i < nv; i += S3_SORT_THREADS) L.u.direct[i] = 0;
                    __syncthreads();
                    for (uint32_t i = threadIdx.x; i < m; i += S3_SORT_THREADS) atomicAdd(&L.u.direct[(uint32_t)(seg[i] & rem_mask)], 1u);
                    __syncthreads();
                    // ordered compaction of the values with count >= lower
                    const uint32_t per = (nv + S3_SORT_THREADS - 1) / S3_SORT_THREADS;
                    const uint32_t v0 = threadIdx.x * per, v1 = (v0 + per < nv) ? v0 + per : nv;
                    uint32_t nk = 0;
                    for (uint32_t v = v0; v < v1; v++) nk += L.u.direct[v] >= lower;
                    L.q.cnt1[threadIdx.x] = nk;
                    __syncthreads();
                    const uint32_t n_kept = s3_block_scan<S3_SORT_THREADS>(L.q.cnt1, L.q.cnt2, S3_SORT_THREADS, L.q.wsum);
                    uint32_t ww = w + L.q.cnt2[threadIdx.x];
                    for (uint32_t v = v0; v < v1; v++) {
                        const uint32_t c = L.u.direct[v];
                        if (c >= lower) {
                            tmp_keys[o + ww] = hi | (KR2)v;
                            tmp_cnts[o + ww] = c;
                            bsum += c;
                            ww++;
                        }
                    }
                    w += n_kept;
                    __syncthreads();
                } else {
                    // hot keys (repeat families put thousands of copies of one k-mer into a bucket): count the
                    // distinct residuals of the piece in an LDS hash table, then sort the few that are kept
                    const KR2 EMPTY = (KR2)~(KR2)0;
                    for (int i = threadIdx.x; i < S3_HASH_SLOTS; i += S3_SORT_THREADS) {
                        L.u.hash.key[i] = EMPTY;
                        L.u.hash.cnt[i] = 0;
                    }
                    if (threadIdx.x == 0) L.ones = L.distinct = 0;
                    __syncthreads();
                    for (uint32_t i = threadIdx.x; i < m; i += S3_SORT_THREADS) {
                        const KR2 v = seg[i];
                        if (v == EMPTY) {              // the table's empty marker: counted aside
                            atomicAdd(&L.ones, 1u);
                            continue;
                        }
                        uint32_t h = s3_hash((unsigned long long)v) & (S3_HASH_SLOTS - 1);
                        for (int probe = 0; probe < S3_HASH_SLOTS; probe++) {
                            const KR2 prev = atomicCAS(&L.u.hash.key[h], EMPTY, v);
                            if (prev == EMPTY) atomicAdd(&L.distinct, 1u);
                            if (prev == EMPTY || prev == v) {
                                atomicAdd(&L.u.hash.cnt[h], 1u);
                                break;
                            }
                            if (L.distinct > S3_HASH_MAX) break;   // too many distinct residuals: fallback
                            h = (h + 1) & (S3_HASH_SLOTS - 1);
                        }
                    }
                    __syncthreads();
                    if (L.distinct > S3_HASH_MAX) {
                        give_up = true;
                    } else {
                        // entries with count >= lower -> (q.s, q.heads) as (key, count), then sort the pairs
                        constexpr int SPT = S3_HASH_SLOTS / S3_SORT_THREADS;
                        uint32_t nk = 0;
                        for (int j = 0; j < SPT; j++) nk += L.u.hash.cnt[threadIdx.x * SPT + j] >= lower;
                        if (threadIdx.x == 0 && L.ones >= lower) nk++;
                        L.q.cnt1[threadIdx.x] = nk;
                        __syncthreads();
                        const uint32_t n_kept = s3_block_scan<S3_SORT_THREADS>(L.q.cnt1, L.q.cnt2, S3_SORT_THREADS, L.q.wsum);
                        uint32_t ww = L.q.cnt2[threadIdx.x];
                        for (int j = 0; j < SPT; j++) {
                            const uint32_t c = L.u.hash.cnt[threadIdx.x * SPT + j];
                            if (c >= lower) {
                                L.q.s[ww] = L.u.hash.key[threadIdx.x * SPT + j];
                                L.q.heads[ww] = c;
                                ww++;
                            }
                        }
                        if (threadIdx.x == 0 && L.ones >= lower) {
                            L.q.s[ww] = EMPTY;
                            L.q.heads[ww] = L.ones;
                        }
                        __syncthreads();
                        KR2 pk_[S3_SORT_PER];
                        uint32_t pv_[S3_SORT_PER];
#pragma unroll
                        for (int j = 0; j < S3_SORT_PER; j++) {
                            const uint32_t i = threadIdx.x * S3_SORT_PER + j;
                            pk_[j] = (i < n_kept) ? L.q.s[i] : EMPTY;
                            pv_[j] = (i < n_kept) ? L.q.heads[i] : 0u;
                        }
                        __syncthreads();
                        s3_block_sort<KR2, S3_SORT_PER, true>(pk_, pv_, L.q.s, L.q.heads);
#pragma unroll
                        for (int j = 0; j < S3_SORT_PER; j++) {
                            const uint32_t i = threadIdx.x * S3_SORT_PER + j;
                            if (i < n_kept) {          // real entries precede the padding (ties: larger count first)
                                tmp_keys[o + w + i] = pk_[j];
                                tmp_cnts[o + w + i] = pv_[j];
                                bsum += pv_[j];
                            }
                        }
                        w += n_kept;
                    }
                    __syncthreads();
                }
            }
        }
        if (threadIdx.x == 0) {
            if (give_up) {
                kept[bucket] = 0;
                const unsigned long long at = atomicAdd(n_big, 1ULL);
                if (at < big_cap) big_list[at] = (unsigned long long)bucket;
            } else {
                kept[bucket] = w;
            }
        }
        if (!give_up) lsum += bsum;
    }
    const unsigned long long tot = sp_block_sum_u64(lsum, L.red);
    if (threadIdx.x == 0 && tot) atomicAdd(len_sum, tot);
}

// All buckets too large for one workgroup at once: their residuals were sorted segment by segment by ONE
// segmented device sort (they used to be sorted one by one from a host loop: 70 buckets per wheat-sized chromosome
// at k = 21, 80 ms of launch latency per pass).  One block per bucket walks its sorted segment: a run head finds the
// end of its run by binary search, runs >= lower are kept in order.
template <typename KR2>
__global__ void __launch_bounds__(256)
s3_big_rle(const KR2 *__restrict__ sorted, const unsigned long long *__restrict__ big, const ulonglong2 *__restrict__ span,
           uint32_t lower, KR2 *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_cnts, unsigned long long *__restrict__ kept,
           unsigned long long *__restrict__ len_sum) {
    __shared__ uint32_t lds[16];
    __shared__ unsigned long long red[16];
    const unsigned long long b = big[blockIdx.x];
    const unsigned long long o = span[b].x, n = span[b].y;
    const KR2 *s = sorted + o;
    unsigned long long w = 0, lsum = 0;
    for (unsigned long long base = 0; base < n; base += 256) {
        const unsigned long long i = base + threadIdx.x;
        KR2 key = 0;
        unsigned long long c = 0;
        if (i < n) {
            key = s[i];
            if (i == 0 || s[i - 1] != key) {     // run head: upper bound of the key in (i, n)
                unsigned long long lo = i + 1, hi = n;
                while (lo < hi) {
                    const unsigned long long mid = (lo + hi) >> 1;
                    if (s[mid] == key) lo = mid + 1;
                    else hi = mid;
                }
                c = lo - i;
            }
        }
        const bool p = c >= lower && c > 0;
        uint32_t tot;
        const uint32_t my = sp_block_excl_count(p, lds, tot);
        if (p) {
            tmp_keys[o + w + my] = key;
            tmp_cnts[o + w + my] = (uint32_t)c;
            lsum += c;
        }
        w += tot;
    }
    const unsigned long long t = sp_block_sum_u64(lsum, red);
    if (threadIdx.x == 0) {
        kept[b] = w;
        if (t) atomicAdd(len_sum, t);
    }
}

// ---------------------------------------------------------------- oversized buckets without a device-wide sort (round 4)
// The handful of buckets per chromosome that s3_final gives up on (a piece with more distinct residuals than its hash
// table takes) used to go through ONE segmented library sort + s3_big_rle, with an 8-MB copy of the span table to the
// host in front of it.  They are an aggregation problem like every other bucket, only larger, so they are cut by HASH
// CLASS instead of by sorting: a bucket of n keys has P = n / 768 + 1 classes, class(key) = mix(key) mod P, and one
// workgroup streams the whole bucket once per class it owns (the bucket sits in L2 / the Infinity Cache after the first
// reader), counts the keys of that class exactly in the LDS table of s3_final_hash and appends the entries that reach
// lower_count to the bucket's kept list.  S3B_SPREAD workgroups share the classes of one bucket.  s3_big_sort then
// orders a bucket's kept list (a bitonic sort in LDS) and writes it where every finish kernel writes.  A class that
// overflows the table or a kept list beyond S3B_KCAP raises a flag and the library path below takes the chromosome's
// oversized buckets as before (wheat-like: never).
#define S3B_SPREAD 32
#define S3B_KCAP 4096
#define S3B_MAXBIG 1024         // more oversized buckets than this per chromosome: the library path
__device__ __forceinline__ uint32_t s3b_class(unsigned long long key, uint32_t P) {
    key ^= key >> 33;
    key *= 0xFF51AFD7ED558CCDULL;
    key ^= key >> 29;
    return (uint32_t)((key & 0xFFFFFFFFULL) % P);
}
template <typename KR2>
__global__ void __launch_bounds__(S3_SORT_THREADS)
s3_big_class(const KR2 *__restrict__ buf2, const unsigned long long *__restrict__ big, const ulonglong2 *__restrict__ span,
             uint32_t lower, KR2 *__restrict__ list_keys, uint32_t *__restrict__ list_cnts,
             unsigned long long *__restrict__ kcur /* per oversized bucket: kept entries so far */,
             unsigned long long *__restrict__ fail) {
    __shared__ s3h_lds<KR2> L;
    const int tid = threadIdx.x;
    const uint32_t bi = blockIdx.x / S3B_SPREAD, q = blockIdx.x % S3B_SPREAD;
    const unsigned long long b = big[bi];
    const unsigned long long o = span[b].x, n = span[b].y;
    const uint32_t P = (uint32_t)(n / 768ULL) + 1u;
    const KR2 *seg = buf2 + o;
    for (uint32_t p = q; p < P; p += S3B_SPREAD) {
        __syncthreads();
        if (tid == 0) L.n_distinct = L.n_kept = L.abort_ = 0;
        s3h_clear_table_words(reinterpret_cast<uint4 *>(L.u.t.key), (int)(S3H_SLOTS * sizeof(KR2) / 16),
                              reinterpret_cast<uint4 *>(L.u.t.cnt), S3H_SLOTS / 4);
        __syncthreads();
        for (unsigned long long i0 = tid; i0 < n; i0 += 8 * S3_SORT_THREADS) {
            KR2 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (i0 + j * S3_SORT_THREADS < n) v[j] = seg[i0 + j * S3_SORT_THREADS];
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (i0 + j * S3_SORT_THREADS < n && s3b_class((unsigned long long)v[j], P) == p && !L.abort_) s3h_insert(L, v[j]);
            if (L.abort_) break;
        }
        __syncthreads();
        if (L.abort_) {                 // block-uniform after the barrier
            if (tid == 0) atomicAdd(fail, 1ULL);
            return;
        }
        for (int s = tid; s < S3H_SLOTS; s += S3_SORT_THREADS) {
            const uint32_t c = L.u.t.cnt[s];
            if (c >= lower) {
                const unsigned long long at = atomicAdd(&kcur[bi], 1ULL);
                if (at < S3B_KCAP) {
                    list_keys[(size_t)bi * S3B_KCAP + at] = L.u.t.key[s];
                    list_cnts[(size_t)bi * S3B_KCAP + at] = c;
                }
            }
        }
    }
}
template <typename KR2>
__global__ void __launch_bounds__(S3_SORT_THREADS)
s3_big_sort(const unsigned long long *__restrict__ big, const ulonglong2 *__restrict__ span, const KR2 *__restrict__ list_keys,
            const uint32_t *__restrict__ list_cnts, const unsigned long long *__restrict__ kcur,
            unsigned long long *__restrict__ fail, KR2 *__restrict__ tmp_keys, uint32_t *__restrict__ tmp_cnts,
            unsigned long long *__restrict__ kept, unsigned long long *__restrict__ len_sum) {
    __shared__ KR2 sk[S3B_KCAP];
    __shared__ uint32_t sc[S3B_KCAP];
    __shared__ unsigned long long red[16];
    const int tid = threadIdx.x;
    const uint32_t bi = blockIdx.x;
    const unsigned long long b = big[bi], o = span[b].x, w64 = kcur[bi];
    if (w64 > S3B_KCAP) {               // block-uniform
        if (tid == 0) atomicAdd(fail, 1ULL);
        return;
    }
    const uint32_t w = (uint32_t)w64;
    uint32_t m = 1;
    while (m < w) m <<= 1;
    const KR2 PAD = (KR2)~(KR2)0;       // never a residual; sorts last
    for (uint32_t i = tid; i < m; i += S3_SORT_THREADS) {
        sk[i] = i < w ? list_keys[(size_t)bi * S3B_KCAP + i] : PAD;
        sc[i] = i < w ? list_cnts[(size_t)bi * S3B_KCAP + i] : 0u;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= m; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (m >> 1); t += S3_SORT_THREADS) {
                const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const KR2 a = sk[lo], c = sk[hi];
                if ((a > c) == up) {
                    sk[lo] = c;
                    sk[hi] = a;
                    const uint32_t x = sc[lo];
                    sc[lo] = sc[hi];
                    sc[hi] = x;
                }
            }
            __syncthreads();
        }
    unsigned long long lsum = 0;
    for (uint32_t i = tid; i < w; i += S3_SORT_THREADS) {
        tmp_keys[o + i] = sk[i];
        tmp_cnts[o + i] = sc[i];
        lsum += sc[i];
    }
    const unsigned long long t = sp_block_sum_u64(lsum, red);
    if (tid == 0) {
        kept[b] = w;
        if (t) atomicAdd(len_sum, t);
    }
}

// ---------------------------------------------------------------- s3_gather
template <typename KR2>
__global__ void __launch_bounds__(256)
s3_gather(const KR2 *__restrict__ tmp_keys, const uint32_t *__restrict__ tmp_cnts,
          const ulonglong2 *__restrict__ span, const unsigned long long *__restrict__ kept_excl,
          const unsigned long long *__restrict__ kept_tot, int64_t n_fine, int R2,
          unsigned long long *__restrict__ out_keys, uint32_t *__restrict__ out_cnts) {
    // one wave per fine bucket
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t b = wave; b < n_fine; b += n_waves) {
        const unsigned long long w0 = kept_excl[b];
        const unsigned long long n = ((b + 1 < n_fine) ? kept_excl[b + 1] : *kept_tot) - w0;
        const unsigned long long o = span[b].x;
        const unsigned long long hi = (R2 >= 64) ? 0ULL : ((unsigned long long)b << R2);
        for (unsigned long long i = lane; i < n; i += 64) {
            out_keys[w0 + i] = hi | (unsigned long long)tmp_keys[o + i];
            out_cnts[w0 + i] = tmp_cnts[o + i];
        }
    }
}

// ---------------------------------------------------------------- multi-block exclusive scan (u64)
// The single-block scan_excl_u64 walks 2^19 bucket counters in 0.75 ms; three of them per chromosome were
// 10 % of the engine.  Chunk sums -> scan of <= 1024 sums -> per-chunk scan with coalesced tiles.
__global__ void __launch_bounds__(256)
s3_scan_sum(const unsigned long long *__restrict__ a, int64_t n, int64_t chunk, unsigned long long *__restrict__ bsum) {
    __shared__ unsigned long long red[16];
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    unsigned long long v = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v += a[i];
    const unsigned long long t = sp_block_sum_u64(v, red);
    if (threadIdx.x == 0) bsum[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256)
s3_scan_apply(unsigned long long *__restrict__ a, int64_t n, int64_t chunk, const unsigned long long *__restrict__ boff) {
    __shared__ unsigned long long wtot[4];
    __shared__ unsigned long long carry_s;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = boff[blockIdx.x];
    __syncthreads();
    for (int64_t base = lo; base < hi; base += 256) {
        const int64_t i = base + threadIdx.x;
        const unsigned long long v = (i < hi) ? a[i] : 0ULL;
        unsigned long long incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long x = __shfl_up(incl, o, 64);
            if (lane >= o) incl += x;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        unsigned long long pre = carry_s, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w < wave) pre += wtot[w];
            tot += wtot[w];
        }
        if (i < hi) a[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
}

// exclusive scan of a[0..n) in place; *total (device) receives the sum.  bsum: >= 1024 u64 of scratch.
static int s3_scan(sp_ctx *ctx, unsigned long long *a, int64_t n, unsigned long long *total, unsigned long long *bsum) {
    if (n <= 4096) {
        SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, a, n, total);
        return SP_OK;
    }
    int64_t nb = (n + 2047) / 2048;
    if (nb > 1024) nb = 1024;
    const int64_t chunk = ((n + nb - 1) / nb + 255) / 256 * 256;
    nb = (n + chunk - 1) / chunk;
    SP_LAUNCH(ctx, "s3_scan_sum", s3_scan_sum, dim3((unsigned)nb), dim3(256), 0, (const unsigned long long *)a, n, chunk, bsum);
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, bsum, nb, total);
    SP_LAUNCH(ctx, "s3_scan_apply", s3_scan_apply, dim3((unsigned)nb), dim3(256), 0, a, n, chunk, (const unsigned long long *)bsum);
    return SP_OK;
}

// ================================================================== host side
#ifndef S3_P1T32
#define S3_P1T32 512
#endif
// One chromosome's chain in three phases, so that several chains can be in flight on several streams (round 4: a chain
// is ~20 launches, several of them one workgroup or latency-bound finish kernels, and used to end in three host
// synchronisations during which the chip idled):
//   A  everything up to s3_final; the overrun flag and the number of oversized buckets travel to page-locked memory
//   B  (after A's event) the oversized buckets' device sort if there are any; scan of the kept counts; totals -> host
//   C  (after B's event) the output list is sized, s3_gather writes it
// The caller issues A for the next chromosomes on other lanes before it waits for B of this one.
struct s3_job {
    int phase = 0;                  // 0 idle, 1 A issued, 2 B issued
    size_t ci = 0;
    bool exact = false, overran = false, empty = false;
    int64_t n_fine = 0;
    unsigned long long cap_tot = 0;
    size_t big_cap = 0;
    unsigned long long *d_small = nullptr, *d_kp = nullptr, *d_big = nullptr, *d_bsum = nullptr;
    ulonglong2 *d_span = nullptr;
    void *buf1 = nullptr, *buf2 = nullptr, *tmp_keys = nullptr;
    uint32_t *tmp_cnts = nullptr;
    unsigned long long *h = nullptr;    // page-locked: [0] oversized buckets, [1] overrun flag, [2] length sum, [3] kept pairs
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
};
#define S3_LANE_BUF(ctx, name) ((ctx)->lane ? (ctx)->lane->name : (ctx)->name)

template <typename KR1, typename KR2>
static int s3_chain_a(sp_ctx *ctx, sp_chrom &c, sp_sparse_chrom &out, const s3_plan &P, const sp_kparams &kp,
                      int lower, s3_job &J) {
    bool exact = J.exact;
    const int64_t len = c.len;
    const int64_t n_fine = (int64_t)P.F1 * P.F2;
    // small arrays: hist1 / off1 [F1+1], cursor1 [F1], tile_start [F1+1], hist2 -> off_fine [n_fine+1],
    // cursor2 [n_fine], kept [n_fine+1], big list, counters
    const size_t big_cap = 1 << 16;
    const size_t o_h1 = 0, o_c1 = o_h1 + (size_t)(P.F1 + 1) * 8, o_e1 = o_c1 + (size_t)P.F1 * 8, o_ts = o_e1 + (size_t)P.F1 * 8,
                 o_of = o_ts + (size_t)(P.F1 + 1) * 8, o_c2 = o_of + (size_t)(n_fine + 1) * 8,
                 o_kp = o_c2 + (size_t)n_fine * 8, o_span = o_kp + (size_t)(n_fine + 1) * 8,
                 o_pl = o_span + (size_t)n_fine * 16, o_big = o_pl + (((size_t)n_fine * 4 + 15) & ~(size_t)15),
                 o_small = o_big + big_cap * 8, o_bsum = o_small + 256, small_bytes = o_bsum + 1024 * 8;
    int rc = sp_buf_ensure(ctx, S3_LANE_BUF(ctx, b_s3_small), (int64_t)small_bytes);
    if (rc) return rc;
    char *S = (char *)S3_LANE_BUF(ctx, b_s3_small).p;
    unsigned long long *d_h1 = (unsigned long long *)(S + o_h1), *d_c1 = (unsigned long long *)(S + o_c1),
                       *d_e1 = (unsigned long long *)(S + o_e1), *d_ts = (unsigned long long *)(S + o_ts), *d_of = (unsigned long long *)(S + o_of),
                       *d_c2 = (unsigned long long *)(S + o_c2), *d_kp = (unsigned long long *)(S + o_kp),
                       *d_big = (unsigned long long *)(S + o_big), *d_small = (unsigned long long *)(S + o_small),
                       *d_bsum = (unsigned long long *)(S + o_bsum);
    ulonglong2 *d_span = (ulonglong2 *)(S + o_span);
    uint32_t *d_pl = (uint32_t *)(S + o_pl);      // buckets handed to s3_final; their number in d_small[8]
    // d_small: [0] total keys, [1] n_big, [2] length sum, [3] kept total, [5] sum of the region capacities, [6] overrun flag
    SP_HIP(ctx, hipMemsetAsync(S, 0, small_bytes, ctx->stream));
    const int64_t n_units64 = (len + S3_P1_UNIT - 1) / S3_P1_UNIT;      // (units of 32 starts since the direct-window scan)
    // level 1: regions from a 1-in-16 stripe sample (long chromosomes) or from the full histogram
    const bool sample1 = !exact && len >= S3_SAMPLE_MIN_LEN;
    const int64_t n_stripes = (n_units64 + S3_STRIPE - 1) / S3_STRIPE;
    const int64_t n_visit = sample1 ? ((n_stripes + (1 << S3_SAMPLE_SHIFT) - 1) >> S3_SAMPLE_SHIFT) * S3_STRIPE : n_units64;
    int64_t grid = (n_visit + 255) / 256;
    const int64_t hist_blocks = sample1 ? (int64_t)ctx->n_cu : (int64_t)ctx->n_cu * 8;
    if (grid > hist_blocks) grid = hist_blocks;
    if (grid < 1) grid = 1;
    SP_LAUNCH(ctx, sample1 ? "s3_hist1_sample" : "s3_hist1", s3_hist1, dim3((unsigned)grid), dim3(256), 0, c.d_pk, c.d_pm, c.d_nm,
              n_units64, n_visit, sample1 ? S3_SAMPLE_SHIFT : 0, kp, P.R1, P.F1, d_h1);
    unsigned long long slack1 = 0;
    if (sample1) {
        const char *es1 = getenv("SP_S3_SLACK1"), *em1 = getenv("SP_S3_MULT1");      // test hooks (force level-1 overruns)
        slack1 = es1 ? (unsigned long long)atoll(es1) : 8192ULL + (unsigned long long)(len >> 16);   // (wide regions slow part1 down: 35 % slack cost 40 %)
        SP_LAUNCH(ctx, "s3_caps1", s3_caps1, dim3(1), dim3(1024), 0, d_h1, P.F1, em1 ? (unsigned long long)atoll(em1) : 33ULL, slack1);
    }
    SP_LAUNCH(ctx, "scan_excl_u64", scan_excl_u64, dim3(1), dim3(1024), 0, d_h1, (int64_t)P.F1 + 1, d_small);
    SP_HIP(ctx, hipMemcpyAsync(d_c1, d_h1, (size_t)P.F1 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    // keys of the chromosome: read back when the histogram was exact; a bound (every start) when it was sampled --
    // nothing below needs more than a bound, and the chain loses a host synchronisation
    unsigned long long nv = (unsigned long long)len, cap1 = 0;
    if (!sample1) {
        SP_HIP(ctx, hipMemcpyAsync(&nv, d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        cap1 = nv;
    } else {
        // the sample sees at most n_visit * 32 k-mers
        cap1 = (((unsigned long long)n_visit * S3_P1_UNIT) << S3_SAMPLE_SHIFT) * 33ULL / 32ULL + (unsigned long long)P.F1 * slack1;
    }
    out.n = 0;
    out.length_sum = 0;
    J.overran = false;
    J.empty = nv == 0;
    J.phase = 1;
    if (nv == 0) return SP_OK;
    if (nv >= (1ULL << 32) - (1ULL << 20) || cap1 >= (1ULL << 32) - (1ULL << 20))
        return sp_fail(ctx, SP_EUNSUP, "k > 15: chromosomes of 2^32 or more k-mers are not supported");
    // keys the level-2 regions can hold: nv exactly (exact = sizes from the full histogram) or a bound on the sum of
    // the sampled capacities -- full tiles contribute exactly an eighth of their keys to the sample, the last tile of a
    // level-1 bucket at most two rows; sum of sqrt <= sqrt(n * sum)
    unsigned long long cap_tot = nv;
    if (!exact) {
        const double s_max = (double)nv / 8.0 + (double)P.F1 * 64.0 * S3_P2_PER;
        cap_tot = (unsigned long long)(10.0 * s_max + 8.0 * (7.0 * sqrt((double)n_fine * s_max) + (double)n_fine) +
                                       (double)S3_CAP_SLACK * (double)n_fine) + 4096ULL;
        if (cap_tot >= (1ULL << 32)) {      // 32-bit positions in the library sort of the hot buckets: exact sizes instead
            exact = true;
            cap_tot = nv;
        }
    }
    constexpr int P1T_ = sizeof(KR1) == 4 ? S3_P1T32 : 256;
    const size_t l1_entries = (size_t)cap1 + (size_t)P1T_ * S3_P1_UNIT + 64;      // regions + the trash area of one tile
    const size_t a_bytes = l1_entries * sizeof(KR1) > (size_t)cap_tot * sizeof(KR2) ? l1_entries * sizeof(KR1) : (size_t)cap_tot * sizeof(KR2);
    rc = sp_buf_ensure(ctx, S3_LANE_BUF(ctx, b_sp_a), (int64_t)a_bytes + 64);      // level-1 records, later the finish kernels' scratch (region-indexed)
    if (rc) return rc;
    rc = sp_buf_ensure(ctx, S3_LANE_BUF(ctx, b_sp_b), (int64_t)cap_tot * (int64_t)sizeof(KR2) + 64);
    if (rc) return rc;
    const size_t tk_bytes = ((size_t)cap_tot * sizeof(KR2) + 63) & ~(size_t)63;
    rc = sp_buf_ensure(ctx, S3_LANE_BUF(ctx, b_sp_c), (int64_t)(tk_bytes + (size_t)cap_tot * 4 + 64));
    if (rc) return rc;
    KR1 *buf1 = (KR1 *)S3_LANE_BUF(ctx, b_sp_a).p;
    KR2 *buf2 = (KR2 *)S3_LANE_BUF(ctx, b_sp_b).p;
    KR2 *tmp_keys = (KR2 *)S3_LANE_BUF(ctx, b_sp_c).p;
    uint32_t *tmp_cnts = (uint32_t *)((char *)S3_LANE_BUF(ctx, b_sp_c).p + tk_bytes);

    constexpr int P1T = sizeof(KR1) == 4 ? S3_P1T32 : 256;
    const int64_t n_units32 = (len + S3_P1_UNIT - 1) / S3_P1_UNIT;
    const int64_t n_tiles1 = (n_units32 + P1T - 1) / P1T;
    int64_t g1 = n_tiles1 < (int64_t)ctx->n_cu * 8 ? n_tiles1 : (int64_t)ctx->n_cu * 8;
    const size_t lds1 = (size_t)P1T * S3_P1_UNIT * sizeof(KR1);
    SP_HIP(ctx, hipFuncSetAttribute((const void *)s3_part1<KR1, P1T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    SP_LAUNCH(ctx, "s3_part1", (s3_part1<KR1, P1T>), dim3((unsigned)g1), dim3(P1T), lds1, c.d_pk, c.d_pm, c.d_nm, n_units32, kp,
              P.R1, P.F1, d_c1, buf1, n_tiles1, (const unsigned long long *)d_h1, (uint32_t)cap1);
    SP_LAUNCH(ctx, "s3_tiles", s3_tiles, dim3(1), dim3(1024), 0, (const unsigned long long *)d_h1, (const unsigned long long *)d_c1,
              P.F1, d_ts, d_e1, d_small + 7, d_small + 6);
    const int64_t est_tiles = (int64_t)(nv / S3_P2_KEYS) + P.F1 + 1;
    int64_t g2 = est_tiles < (int64_t)ctx->n_cu * 8 ? est_tiles : (int64_t)ctx->n_cu * 8;
    SP_LAUNCH(ctx, exact ? "s3_hist2" : "s3_hist2_sample", s3_hist2<KR1>, dim3((unsigned)g2), dim3(S3_P2_THREADS), 0, (const KR1 *)buf1,
              (const unsigned long long *)d_h1, (const unsigned long long *)d_e1, (const unsigned long long *)d_ts, P.F1, P.F2, P.R2,
              d_of, exact ? 0 : 1);
    if (!exact) {
        const char *em = getenv("SP_S3_MULT"), *es = getenv("SP_S3_SLACK");      // test hooks (force overruns)
        SP_LAUNCH(ctx, "s3_caps", s3_caps, dim3((unsigned)((n_fine + 255) / 256)), dim3(256), 0, d_of, n_fine,
                  em ? (unsigned long long)atoll(em) : S3_CAP_MULT, es ? (unsigned long long)atoll(es) : S3_CAP_SLACK);
    }
    rc = s3_scan(ctx, d_of, n_fine + 1, d_small + 5, d_bsum);
    if (rc) return rc;
    int64_t g3 = est_tiles < (int64_t)ctx->n_cu * 16 ? est_tiles : (int64_t)ctx->n_cu * 16;
    SP_LAUNCH(ctx, "s3_part2", (s3_part2<KR1, KR2>), dim3((unsigned)g3), dim3(S3_P2_THREADS), 0, (const KR1 *)buf1,
              (const unsigned long long *)d_h1, (const unsigned long long *)d_e1, (const unsigned long long *)d_ts, P.F1, P.F2, P.R2,
              (const unsigned long long *)d_of, d_c2, buf2);
    SP_LAUNCH(ctx, "s3_spans", s3_spans, dim3((unsigned)((n_fine + 255) / 256)), dim3(256), 0, (const unsigned long long *)d_of,
              (const unsigned long long *)d_c2, n_fine, d_span, d_small + 6);
    if (const char *dump = getenv("SP_S3_DUMP")) {      // dev hook (tools/s3_bucket_stats.py): spans and residuals of this chromosome -> files
        std::vector<ulonglong2> hs((size_t)n_fine);
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SP_HIP(ctx, hipMemcpy(hs.data(), d_span, (size_t)n_fine * 16, hipMemcpyDeviceToHost));
        unsigned long long hi = 0;
        for (auto &e : hs) if (e.x + e.y > hi) hi = e.x + e.y;
        std::vector<KR2> hk((size_t)hi);
        SP_HIP(ctx, hipMemcpy(hk.data(), buf2, (size_t)hi * sizeof(KR2), hipMemcpyDeviceToHost));
        const std::string base(dump);
        if (FILE *f = fopen((base + ".spans").c_str(), "wb")) { fwrite(hs.data(), 16, hs.size(), f); fclose(f); }
        if (FILE *f = fopen((base + ".keys").c_str(), "wb")) { fwrite(hk.data(), sizeof(KR2), hk.size(), f); fclose(f); }
    }
    int64_t g4 = n_fine < (int64_t)ctx->n_cu * 64 ? n_fine : (int64_t)ctx->n_cu * 64;
    const bool bitmap = P.R2 <= S3_BM_MAXBITS && sizeof(KR2) == 4;      // k = 16, 17
    // k >= 18: hashed pre-count + exact table of the survivors (SP_S3_FINAL=sort: the block radix sort of round 2)
    const char *env_fin = getenv("SP_S3_FINAL");
    const bool use_hash = !bitmap && !(env_fin && !strcmp(env_fin, "sort"));
    if (bitmap)
        SP_LAUNCH(ctx, "s3_final_bitmap", s3_final_bitmap<KR2>, dim3((unsigned)g4), dim3(S3_SORT_THREADS),
                  (size_t)(P.R2 > 5 ? 1 << (P.R2 - 5) : 1) * 6 + (size_t)S3_BM_CAP * 4,
                  (const KR2 *)buf2, (const ulonglong2 *)d_span, n_fine, P.R2, (uint32_t)lower, tmp_keys, tmp_cnts,
                  d_kp, d_small + 2, d_pl, d_small + 8);
    else if (use_hash)
        SP_LAUNCH(ctx, "s3_final_hash", s3_final_hash<KR2>, dim3((unsigned)g4), dim3(S3_SORT_THREADS), 0,
                  (const KR2 *)buf2, (const ulonglong2 *)d_span, n_fine, P.R2, (uint32_t)lower, tmp_keys, tmp_cnts,
                  d_kp, d_small + 2, d_pl, d_small + 8);
    else
        SP_LAUNCH(ctx, "s3_final_small", s3_final_small<KR2>, dim3((unsigned)g4), dim3(S3_SORT_THREADS), 0,
                  (const KR2 *)buf2, (const ulonglong2 *)d_span, n_fine, P.R2, (uint32_t)lower, tmp_keys, tmp_cnts,
                  d_kp, d_small + 2);
    const int64_t g4f = (bitmap || use_hash) ? ((int64_t)ctx->n_cu * 8 < g4 ? (int64_t)ctx->n_cu * 8 : g4) : g4;   // punt mode: the listed buckets only
    SP_LAUNCH(ctx, "s3_final", s3_final<KR2>, dim3((unsigned)g4f), dim3(S3_SORT_THREADS), 0, (const KR2 *)buf2,
              (KR2 *)buf1, (const ulonglong2 *)d_span, n_fine, P.R2, (uint32_t)lower, tmp_keys, tmp_cnts, d_kp, d_big,
              d_small + 1, (unsigned long long)big_cap, d_small + 2,
              (bitmap || use_hash) ? S3_BM_PUNT : (unsigned long long)S3_SMALL_CAP, use_hash ? 1 : 0, (const uint32_t *)d_pl,
              (const unsigned long long *)(d_small + 8), getenv("SP_S3_FORCE_BIG") ? 1 : 0);
    SP_HIP(ctx, hipMemcpyAsync(J.h, d_small + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(J.h + 1, d_small + 6, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipEventRecord(J.ev_a, ctx->stream));
    J.exact = exact;
    J.n_fine = n_fine;
    J.cap_tot = cap_tot;
    J.big_cap = big_cap;
    J.d_small = d_small;
    J.d_kp = d_kp;
    J.d_big = d_big;
    J.d_bsum = d_bsum;
    J.d_span = d_span;
    J.buf1 = buf1;
    J.buf2 = buf2;
    J.tmp_keys = tmp_keys;
    J.tmp_cnts = tmp_cnts;
    return SP_OK;
}

template <typename KR1, typename KR2>
static int s3_chain_b(sp_ctx *ctx, sp_sparse_chrom &out, const s3_plan &P, int lower, s3_job &J) {
    J.phase = 2;
    if (J.empty) return SP_OK;
    SP_HIP(ctx, hipEventSynchronize(J.ev_a));
    const unsigned long long n_big = J.h[0], h_flag = J.h[1];
    const int64_t n_fine = J.n_fine;
    const size_t big_cap = J.big_cap;
    const unsigned long long cap_tot = J.cap_tot;
    unsigned long long *d_small = J.d_small, *d_kp = J.d_kp, *d_big = J.d_big, *d_bsum = J.d_bsum;
    ulonglong2 *d_span = J.d_span;
    KR1 *buf1 = (KR1 *)J.buf1;
    KR2 *buf2 = (KR2 *)J.buf2, *tmp_keys = (KR2 *)J.tmp_keys;
    uint32_t *tmp_cnts = J.tmp_cnts;
    int rc = SP_OK;
    if (h_flag) {      // a region overran (keys were dropped): the caller counts the chromosome again with exact sizes
        J.overran = true;
        out.n = 0;
        out.length_sum = 0;
        return SP_OK;
    }
    if (n_big > big_cap) return sp_fail(ctx, SP_EUNSUP, "k > 15: %llu oversized k-mer buckets", n_big);
    bool big_done = false;
    const char *env_big = getenv("SP_S3_BIG");      // "sort": the library path always (cross-check)
    if (n_big && n_big <= S3B_MAXBIG && !(env_big && !strcmp(env_big, "sort"))) {      // round 4: by hash class, no device-wide sort
        const size_t lk_bytes = ((size_t)n_big * S3B_KCAP * sizeof(KR2) + 255) & ~(size_t)255;
        const size_t lc_bytes = ((size_t)n_big * S3B_KCAP * 4 + 255) & ~(size_t)255;
        const size_t cur_bytes = (((size_t)n_big + 1) * 8 + 255) & ~(size_t)255;
        rc = sp_buf_ensure(ctx, S3_LANE_BUF(ctx, b_sp_tmp), (int64_t)(lk_bytes + lc_bytes + cur_bytes + 256));
        if (rc) return rc;
        char *T = (char *)S3_LANE_BUF(ctx, b_sp_tmp).p;
        KR2 *lk = (KR2 *)T;
        uint32_t *lc = (uint32_t *)(T + lk_bytes);
        unsigned long long *kcur = (unsigned long long *)(T + lk_bytes + lc_bytes), *d_fail = kcur + n_big;
        SP_HIP(ctx, hipMemsetAsync(kcur, 0, ((size_t)n_big + 1) * 8, ctx->stream));
        SP_LAUNCH(ctx, "s3_big_class", s3_big_class<KR2>, dim3((unsigned)(n_big * S3B_SPREAD)), dim3(S3_SORT_THREADS), 0,
                  (const KR2 *)buf2, (const unsigned long long *)d_big, (const ulonglong2 *)d_span, (uint32_t)lower, lk, lc, kcur, d_fail);
        // (the length sum is added by s3_big_sort; a failed bucket leaves the sum short, and the library path below then
        // redoes EVERY oversized bucket -- so the sums of the buckets already written have to come out again)
        SP_HIP(ctx, hipMemcpyAsync(J.h + 4, d_small + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_LAUNCH(ctx, "s3_big_sort", s3_big_sort<KR2>, dim3((unsigned)n_big), dim3(S3_SORT_THREADS), 0,
                  (const unsigned long long *)d_big, (const ulonglong2 *)d_span, (const KR2 *)lk, (const uint32_t *)lc,
                  (const unsigned long long *)kcur, d_fail, tmp_keys, tmp_cnts, d_kp, d_small + 2);
        SP_HIP(ctx, hipMemcpyAsync(J.h + 5, d_fail, 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        big_done = J.h[5] == 0;
        if (!big_done)      // back to the length sum before s3_big_sort
            SP_HIP(ctx, hipMemcpyAsync(d_small + 2, J.h + 4, 8, hipMemcpyHostToDevice, ctx->stream));
    }
    if (n_big && !big_done) {   // the library path: ONE segmented device sort, then one block per bucket
        std::vector<unsigned long long> big((size_t)n_big), seg(2 * (size_t)n_big);
        std::vector<ulonglong2> spans((size_t)n_fine);
        SP_HIP(ctx, hipMemcpy(big.data(), d_big, (size_t)n_big * 8, hipMemcpyDeviceToHost));
        SP_HIP(ctx, hipMemcpy(spans.data(), d_span, (size_t)n_fine * 16, hipMemcpyDeviceToHost));
        for (size_t bi = 0; bi < (size_t)n_big; bi++) {
            seg[bi] = spans[(size_t)big[bi]].x;
            seg[(size_t)n_big + bi] = spans[(size_t)big[bi]].x + spans[(size_t)big[bi]].y;
        }
        KR2 *srt = (KR2 *)buf1;      // the level-1 buffer is free by now and holds every region (a_bytes above)
        size_t tb = 0;
        const unsigned long long *nul = nullptr;
        SP_HIP(ctx, rocprim::segmented_radix_sort_keys(nullptr, tb, (const KR2 *)buf2, srt, (unsigned int)cap_tot, (unsigned int)n_big,
                                                       nul, nul, 0u, (unsigned)(P.R2 > 0 ? P.R2 : 1), ctx->stream));
        const size_t seg_bytes = 2 * (size_t)n_big * 8;
        rc = sp_buf_ensure(ctx, S3_LANE_BUF(ctx, b_sp_tmp), (int64_t)(tb + seg_bytes + 512));
        if (rc) return rc;
        unsigned long long *d_seg = (unsigned long long *)((char *)S3_LANE_BUF(ctx, b_sp_tmp).p + ((tb + 255) & ~(size_t)255));
        SP_HIP(ctx, hipMemcpyAsync(d_seg, seg.data(), seg_bytes, hipMemcpyHostToDevice, ctx->stream));
        SP_HIP(ctx, rocprim::segmented_radix_sort_keys(S3_LANE_BUF(ctx, b_sp_tmp).p, tb, (const KR2 *)buf2, srt, (unsigned int)cap_tot,
                                                       (unsigned int)n_big, (const unsigned long long *)d_seg,
                                                       (const unsigned long long *)(d_seg + n_big), 0u,
                                                       (unsigned)(P.R2 > 0 ? P.R2 : 1), ctx->stream));
        SP_LAUNCH(ctx, "s3_big_rle", s3_big_rle<KR2>, dim3((unsigned)n_big), dim3(256), 0, (const KR2 *)srt,
                  (const unsigned long long *)d_big, (const ulonglong2 *)d_span, (uint32_t)lower, tmp_keys, tmp_cnts,
                  d_kp, d_small + 2);
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));      // `seg` (pageable) must outlive the copy
    }
    rc = s3_scan(ctx, d_kp, n_fine, d_small + 3, d_bsum);
    if (rc) return rc;
    SP_HIP(ctx, hipMemcpyAsync(J.h + 2, d_small + 2, 16, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipEventRecord(J.ev_b, ctx->stream));
    return SP_OK;
}

template <typename KR1, typename KR2>
static int s3_chain_c(sp_ctx *ctx, sp_sparse_chrom &out, const s3_plan &P, s3_job &J) {
    J.phase = 0;
    if (J.empty || J.overran) return SP_OK;
    SP_HIP(ctx, hipEventSynchronize(J.ev_b));
    const int64_t n_fine = J.n_fine;
    unsigned long long *d_small = J.d_small, *d_kp = J.d_kp;
    ulonglong2 *d_span = J.d_span;
    KR2 *tmp_keys = (KR2 *)J.tmp_keys;
    uint32_t *tmp_cnts = J.tmp_cnts;
    const unsigned long long *h = J.h + 2;
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
    if (keep) {
        int64_t g5 = (n_fine * 64 + 255) / 256;
        if (g5 > (int64_t)ctx->n_cu * 32) g5 = (int64_t)ctx->n_cu * 32;
        SP_LAUNCH(ctx, "s3_gather", s3_gather<KR2>, dim3((unsigned)g5), dim3(256), 0, (const KR2 *)tmp_keys,
                  (const uint32_t *)tmp_cnts, (const ulonglong2 *)d_span, (const unsigned long long *)d_kp,
                  (const unsigned long long *)(d_small + 3), n_fine, P.R2, (unsigned long long *)out.d_keys, out.d_cnts);
    }
    return SP_OK;
}

// phase dispatch on the residual widths of the plan
static int s3_chain(sp_ctx *ctx, int phase, sp_chrom &c, sp_sparse_chrom &o, const s3_plan &P, const sp_kparams &kp, int lower, s3_job &J) {
#define S3_PHASES(KR1, KR2)                                                                 \
    (phase == 0   ? s3_chain_a<KR1, KR2>(ctx, c, o, P, kp, lower, J)                        \
     : phase == 1 ? s3_chain_b<KR1, KR2>(ctx, o, P, lower, J)                               \
                  : s3_chain_c<KR1, KR2>(ctx, o, P, J))
    if (!P.wide1) return S3_PHASES(uint32_t, uint32_t);
    if (!P.wide2) return S3_PHASES(unsigned long long, uint32_t);
    return S3_PHASES(unsigned long long, unsigned long long);
#undef S3_PHASES
}

int sp_sparse_count3(sp_ctx *ctx, int k, int lower) {
    const size_t C = ctx->chroms.size();
    if (ctx->sparse.size() != C) {
        for (auto &c : ctx->sparse) sps_free_chrom(c);
        ctx->sparse.assign(C, sp_sparse_chrom());
    }
    const sp_kparams kp = sp_make_kparams(k);
    const s3_plan P = s3_make_plan(k);
    const char *env_exact = getenv("SP_S3_EXACT");      // "1": level-2 regions from the full histogram (the round-2 path)
    const bool exact0 = env_exact && env_exact[0] == '1';
    // Lanes (round 4): the chains of SP_LANES_SPARSE chromosomes (default 3; 0 = the context's stream alone) are in
    // flight at once, each on its own stream with its own buffers; a lane starts a chromosome when that chromosome's
    // pack kernel is done (sp_chrom::ev_packed).
    const char *el = getenv("SP_LANES_SPARSE");
    int n_lanes = el ? atoi(el) : 3;
    int64_t total_len = 0;
    for (const auto &c : ctx->chroms) total_len += c.len;
    if (!el && total_len < (1LL << 24)) n_lanes = 0;      // (toy inputs: see sp_count)
    if (n_lanes > SP_MAX_LANES) n_lanes = SP_MAX_LANES;
    if (n_lanes < 0 || C < 2) n_lanes = 0;
    if (n_lanes > 0) {
        // Every lane owns the partition buffers of one chain, and their floor does not shrink with the genome: the level-2
        // regions carry S3_CAP_SLACK entries of slack per fine bucket -- 2^18..2^19 buckets x 1024 x (residual + scratch +
        // count) = 3.2 GB (k <= 17), 6.4 GB (k <= 25), 10.7 GB (64-bit residuals) per lane.  Lanes whose buffers are not
        // allocated yet must fit what the device has free (round 6: several processes on one GPU with seven lanes forced ran
        // out of 288 GB); the chains then share fewer streams, down to the context's own.
        int64_t longest = 0;
        for (const auto &c : ctx->chroms) longest = c.len > longest ? c.len : longest;
        const double n_fine = (double)P.F1 * (double)P.F2;
        const double s_max = (double)longest / 8.0 + (double)P.F1 * 64.0 * S3_P2_PER;
        const double cap_tot = 10.0 * s_max + 8.0 * (7.0 * sqrt(n_fine * s_max) + n_fine) + (double)S3_CAP_SLACK * n_fine + 4096.0;
        const double r2 = P.wide2 ? 8.0 : 4.0, r1 = P.wide1 ? 8.0 : 4.0;
        const double l1 = ((double)longest * 33.0 / 32.0 + (double)P.F1 * (8192.0 + (double)(longest >> 16)) + 32768.0) * r1;
        const int64_t per_lane = (int64_t)((cap_tot * r2 > l1 ? cap_tot * r2 : l1) + cap_tot * r2 + cap_tot * (r2 + 4.0)) + (64LL << 20);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            int64_t budget = (int64_t)free_b - (int64_t)(total_b / 16);      // keep a sixteenth of the device free
            int fit = 0;
            for (int l = 0; l < n_lanes; l++) {
                const sp_ctx::lane_t &ln = ctx->lanes[l];
                const int64_t have = ln.b_sp_a.cap + ln.b_sp_b.cap + ln.b_sp_c.cap;
                const int64_t need = have >= per_lane ? 0 : per_lane - have;
                if (need > budget) break;
                budget -= need;
                fit++;
            }
            if (fit < n_lanes) {
                if (getenv("SP_DEBUG_COUNT")) fprintf(stderr, "[sp] k > 15 count: %d of %d lanes fit the free device memory\n", fit, n_lanes);
                n_lanes = fit;
            }
        }
    }
    if (!ctx->h_s3) SP_HIP(ctx, hipHostMalloc((void **)&ctx->h_s3, (size_t)(SP_MAX_LANES + 1) * 8 * sizeof(unsigned long long), hipHostMallocDefault));
    for (auto &e : ctx->s3_ev)
        if (!e) SP_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int l = 0; l < n_lanes; l++) {
        sp_ctx::lane_t &ln = ctx->lanes[l];
        if (!ln.stream) {
            SP_HIP(ctx, hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            SP_HIP(ctx, hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
        }
        if (!ln.ev_a) SP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_a, hipEventDisableTiming));
        if (!ln.ev_b) SP_HIP(ctx, hipEventCreateWithFlags(&ln.ev_b, hipEventDisableTiming));
    }
    if (n_lanes) {      // chromosomes without an event of their own: the lanes start behind what the main stream holds now
        if (!ctx->lane_go) SP_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_go, hipEventDisableTiming));
        SP_HIP(ctx, hipEventRecord(ctx->lane_go, ctx->stream));
    }
    const int n_jobs = n_lanes ? n_lanes : 1;
    std::vector<s3_job> jobs((size_t)n_jobs);
    for (int l = 0; l < n_jobs; l++) {
        jobs[(size_t)l].h = ctx->h_s3 + 8 * (size_t)l;
        jobs[(size_t)l].ev_a = n_lanes ? ctx->lanes[l].ev_a : ctx->s3_ev[0];
        jobs[(size_t)l].ev_b = n_lanes ? ctx->lanes[l].ev_b : ctx->s3_ev[1];
    }
    std::vector<size_t> redo;      // chromosomes whose sampled regions overran: counted again below, exact sizes, one by one
    hipStream_t main_stream = ctx->stream;
    // (a lambda: an error return inside must restore the context's stream and let the lanes drain)
    auto on_lane = [&](int l, int phase, s3_job &J) -> int {
        if (n_lanes) {
            ctx->lane = &ctx->lanes[l];
            ctx->stream = ctx->lanes[l].stream;
        }
        const int rc = s3_chain(ctx, phase, ctx->chroms[J.ci], ctx->sparse[J.ci], P, kp, lower, J);
        ctx->lane = nullptr;
        ctx->stream = main_stream;
        return rc;
    };
    auto finish = [&](int l) -> int {
        s3_job &J = jobs[(size_t)l];
        if (!J.phase) return SP_OK;
        int rc = on_lane(l, 1, J);
        if (!rc) rc = on_lane(l, 2, J);
        if (rc) return rc;
        if (J.overran) redo.push_back(J.ci);
        ctx->chroms[J.ci].length_sum = ctx->sparse[J.ci].length_sum;
        ctx->chroms[J.ci].n_dump = ctx->sparse[J.ci].n;
        return SP_OK;
    };
    auto run_all = [&]() -> int {
        size_t issued = 0;
        for (size_t ci = 0; ci < C; ci++) {
            sp_chrom &c = ctx->chroms[ci];
            sp_sparse_chrom &o = ctx->sparse[ci];
            o.n = 0;
            o.length_sum = 0;
            c.length_sum = 0;
            c.n_dump = 0;
            if (c.len <= 0) continue;
            const int l = (int)(issued++ % (size_t)n_jobs);
            int rc = finish(l);      // the chain this lane issued before: its buffers are free once its gather is queued
            if (rc) return rc;
            s3_job &J = jobs[(size_t)l];
            J.ci = ci;
            J.exact = exact0;
            if (n_lanes) SP_HIP(ctx, hipStreamWaitEvent(ctx->lanes[l].stream, c.ev_packed ? c.ev_packed : ctx->lane_go, 0));
            rc = on_lane(l, 0, J);
            if (rc) return rc;
        }
        for (size_t q = 0; q < (size_t)n_jobs; q++) {      // the chains still in flight, oldest first
            const int rc = finish((int)((issued + q) % (size_t)n_jobs));
            if (rc) return rc;
        }
        return SP_OK;
    };
    int rc = run_all();
    ctx->lane = nullptr;
    ctx->stream = main_stream;
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
    for (size_t ci : redo) {
        if (exact0) return sp_fail(ctx, SP_ESTATE, "k > 15: a bucket overran a region of its exact size");
        ctx->c2_recounts++;
        if (getenv("SP_DEBUG_COUNT")) fprintf(stderr, "[sp] k > 15: chromosome %zu recounted with exact bucket sizes\n", ci);
        s3_job J;
        J.ci = ci;
        J.exact = true;
        J.h = ctx->h_s3 + 8 * (size_t)SP_MAX_LANES;
        J.ev_a = ctx->s3_ev[0];
        J.ev_b = ctx->s3_ev[1];
        for (int phase = 0; phase < 3; phase++) {
            rc = s3_chain(ctx, phase, ctx->chroms[ci], ctx->sparse[ci], P, kp, lower, J);
            if (rc) return rc;
        }
        if (J.overran) return sp_fail(ctx, SP_ESTATE, "k > 15: a bucket overran a region of its exact size");
        ctx->chroms[ci].length_sum = ctx->sparse[ci].length_sum;
        ctx->chroms[ci].n_dump = ctx->sparse[ci].n;
    }
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}
