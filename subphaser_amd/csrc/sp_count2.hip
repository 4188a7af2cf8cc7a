// sp_count2.hip -- K1 engine 2: LDS radix-partition k-mer counter.
//
// Random 4-byte read-modify-writes into a 2-GiB table run at ~19 G updates/s on
// MI355X (profiles/r01_ubench_mi355x.txt), LDS atomics at ~600 G/s.  This engine
// turns the random updates into streaming traffic:
//
//   c2_hist_fine  histogram of the top B1+B2 slot bits (LDS histogram per block, one global atomic per
//             non-empty bin).  Round 3: it scans ONE STRIPE IN 16 of the chromosome and the bucket regions are
//             laid out from the estimate with slack (memory is plentiful: 288 GB); the exact sizes are what the
//             cursors of part1 / part2 hold afterwards.  A bucket that outgrows its region raises a flag and the
//             chromosome is counted again with the exact histogram (the round-2 path), so results never depend on
//             the estimate.  (The whole-genome scan this replaces was 6.3 of 136 ms per wheat-like pass.)
//   c2_part1  scan again; LDS counting sort of a 16K-key tile on the top B1
//             bits; each bucket run is written as one coalesced burst (3 bytes per key:
//             a u16 plane and a u8 plane)
//   c2_part2  per level-1 bucket: LDS counting sort of 16K-key tiles on the next
//             B2 bits; only the low 15 slot bits survive, written as u16
//   c2_count  one workgroup per fine bucket (2^15 consecutive slots): 128 KiB of
//             u32 counters in LDS, LDS atomics, then ONE streaming write of the
//             table slice, fused with K2 (sum and number of counts >= lower)
//
// All table writes and all key reads/writes are coalesced; the only random
// accesses left are LDS atomics.  Physical HBM bytes per base (k = 15):
//   0.625 x2 (scans) + 3 w + 3 r + 2 w + 2 r + 1 x slots/base (table write).
#include "sp_device.h"
#include "sp_c2batch.h"

#define C2_B3 15                      // slot bits resolved inside LDS
#define C2_FINE (1 << C2_B3)          // slots per fine bucket
#ifndef C2_P1_THREADS
#define C2_P1_THREADS 512             // part1: 512 threads x 32 k-mer starts
#endif
#define C2_P1_UNIT 32
#define C2_P1_KEYS (C2_P1_THREADS * C2_P1_UNIT)   // starts (<= keys) per part1 tile
#ifndef C2_P2_THREADS
#define C2_P2_THREADS 256             // (round 4, after the cursor fix: 256 x 24: 0.861 ms, 512 x 24: 0.897, 512 x 16: 0.990, 768 x 16: 1.20) part2 tile = threads x keys per thread.  Round 1 (software pipeline): 512 x 8 -> 1.37 ms per
                                      // 667-Mb chromosome, 768 x 12 -> 1.22.  Round 2 (rank from the histogram atomic, one table read
                                      // in the copy-out): 768 x 12 1.21, 768 x 16 1.31, 768 x 20 1.25, 1024 x 12 1.23, 1024 x 16 1.17,
                                      // 512 x 24 1.16 (adopted: 12 K-key tiles, 288-byte runs, three blocks per CU)
#endif
#ifndef C2_P2_PER
#define C2_P2_PER 24
#endif
#define C2_TILE_KEYS (C2_P2_THREADS * C2_P2_PER)  // keys per part2 tile
#ifndef C2_P2_SPREAD
#define C2_P2_SPREAD 128              // part2 tile order: concurrently processed tiles lie n_tiles / 128 apart
#endif
#define C2_HIST_THREADS 512
#define C2_MAXF 128                   // max fan-out per level (k <= 15: at most 2^29 slots = 7 + 7 + 15 bits)
#define C2_DROP (~0ULL)               // delta / gbase of a run that does not fit its region (estimate mode)
// Records travel in QUADS (round 4).  A bucket run written by one tile is padded to a multiple of four records, so that
// the copy-out of both partition kernels handles four consecutive records per lane: one 16-byte LDS read, one look-up
// of the run's base, one 8-byte (+ one 4-byte) global store -- a quarter of the vector-memory instructions of the
// one-record-per-lane form, which issued a 2-byte and a 1-byte store per key and was bound by exactly that (the
// address path takes a wave-instruction's lanes four per clock whatever their width).  A pad record is all ones: a
// level-1 record carries at most 22 slot bits in 24, a residual 15 bits in 16.
#define C2_INVALID1 0xFFFFFFu
#define C2_INVALID2 0xFFFFu
#define C2_PADS (3 * C2_MAXF)         // pad records one tile can add at most
// Cursors (round 4, tools/ubench_frag.hip).  Returning atomics on ONE cache line serialise: 128 cursors packed into
// 1 KiB held a 40 K-tile partition pass at 0.88 ms whatever it wrote; one cursor per 128-byte line 0.52 ms; eight
// sub-cursors per bucket on a line each 0.29 ms.  So every cursor owns a line, and a level-1 bucket is cut into
// C2_SPLIT sub-regions with a cursor each: tile t appends to sub-region t mod C2_SPLIT (a static, even deal of the
// tiles; with the blocks of a launch dealt round-robin to the XCDs, usually also a deal by L2).  Downstream a
// sub-region is just a level-1 "bucket" of its own: V1 = F1 * C2_SPLIT of them.
#define C2_SPLIT 8
#define C2_CSTRIDE 16                 // cursor stride in 8-byte words
#define C2_MAXV (C2_MAXF * C2_SPLIT)
// Layout (measured on one 667-Mb chain, part1 / part2 ms): every cursor on a line of its own 0.922 / 0.953; the 128
// cursors that ONE tile touches packed together -- sub-region m of all buckets in 1 KiB, the sub-regions apart --
// 0.883 / 0.899: a wave's atomics then touch 8 lines instead of 64, and only the tiles of one residue class (part1) or of
// one level-1 bucket (part2, whose concurrent tiles lie in different buckets) share them.
#ifndef C2_CSTRIDE2
#define C2_CSTRIDE2 1                 // stride of the fine (part2) cursors
#endif
// where the cursor of level-1 sub-region vb = b * C2_SPLIT + m lives (8-byte words)
#ifndef C2_CUR1_PACKED
#define C2_CUR1_PACKED 1
#endif
#if C2_CUR1_PACKED
#define C2_CUR1(vb) ((size_t)((vb) % C2_SPLIT) * C2_MAXF * C2_CUR1_PACKED + (size_t)((vb) / C2_SPLIT) * C2_CUR1_PACKED)
#else
#define C2_CUR1(vb) ((size_t)(vb) * C2_CSTRIDE)
#endif

struct c2_plan {
    int T;        // log2(nslots)
    int B1, B2;   // partition bits
    int F1, F2;
    int64_t n_fine;  // F1*F2
};

static bool c2_make_plan(int64_t nslots, c2_plan &p) {
    int T = 0;
    while ((1LL << T) < nslots) T++;
    if ((1LL << T) != nslots) return false;
    int R = T - C2_B3;
    if (R < 2) return false;
    p.T = T;
    p.B1 = (R + 1) / 2;
    p.B2 = R / 2;
    if (p.B1 > 7 || p.B2 > 7) return false;
    p.F1 = 1 << p.B1;
    p.F2 = 1 << p.B2;
    p.n_fine = (int64_t)p.F1 * p.F2;
    return true;
}

// ---------------------------------------------------------------- c2_hist_fine
// Fine histogram (top B1+B2 slot bits) over the units of 32 starts, all of them (sample_shift = 0: exact bucket
// sizes) or one stripe of C2_STRIPE units in every 2^sample_shift stripes (an estimate).  The sampled stripe of a
// group rotates with the group index so that no period of the sequence can hide from the sample.
#define C2_STRIPE 64                  // units per stripe: 2048 starts = 512 B of each packed stream, one wave
#define C2_SAMPLE_SHIFT 4             // one stripe in 16
__device__ __forceinline__ int64_t c2_sample_unit(int64_t j, int sample_shift) {
    if (sample_shift == 0) return j;
    const int64_t g = j / C2_STRIPE;                      // sampled stripe number = stripe group
    const int64_t phase = (g * 7 + (g >> 4)) & ((1 << sample_shift) - 1);
    return ((g << sample_shift) + phase) * C2_STRIPE + (j % C2_STRIPE);
}
__device__ __forceinline__ void c2_hist_fine_body(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm,
             int64_t n_units /* of 32 starts */, int64_t n_visit, int sample_shift, sp_kparams32 kp, int shift_fine,
             int n_fine, unsigned long long *__restrict__ ghist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lh[];  // n_fine
    for (int i = threadIdx.x; i < n_fine; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_visit; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = c2_sample_unit(j, sample_shift);
        if (u >= n_units) continue;
        auto scan = [&](auto parity) {
            sp_scan32_slots<decltype(parity)::value>(pk, pm, nm, u * C2_P1_UNIT, kp,
                                                     [&](uint32_t slot) { atomicAdd(&lh[slot >> shift_fine], 1u); });
        };
        if (kp.odd) scan(sp_odd_tag{});
        else scan(sp_even_tag{});
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_fine; i += blockDim.x) {
        uint32_t v = lh[i];
        if (v) atomicAdd(&ghist[i], (unsigned long long)v);
    }
}

// Region layout from the fine histogram (n_fine <= 16384).  cap(f) = ghist[f] (exact mode: mult = 1, slack = 0) or
// ghist[f] * mult / 8 + slack (estimate mode), plus the pad records the bucket can receive (three per part2 tile of its
// level-1 bucket at most), rounded up to a multiple of 4.  off_fine[n_fine + 1] = exclusive scan of the capacities =
// where every fine bucket's residuals start in buf2; off1[b] = where level-1 bucket b starts in the level-1 planes
// (its own bound + three pad records per part1 tile, a multiple of 4).  Sizes and tile starts come later, from
// part1's cursors (c2_tiles).
__device__ __forceinline__ void c2_offsets_body(const unsigned long long *__restrict__ ghist, int n_fine, int F1, int F2, unsigned long long mult8,
           unsigned long long mult8_1, unsigned long long slack, unsigned long long slack1, unsigned long long pad1,
           int split /* C2_SPLIT, or 1: everything into sub-region 0 (exact mode) */, unsigned long long sslack,
           unsigned long long *__restrict__ off_fine, unsigned long long *__restrict__ off1 /* F1 * C2_SPLIT + 1 starts */) {
    __shared__ unsigned long long wsum[16];
    const int T = 1024;
    const bool exact = !slack && mult8 == 8ULL;
    int per = (n_fine + T - 1) / T;
    int lo = threadIdx.x * per, hi = lo + per;
    if (lo > n_fine) lo = n_fine;
    if (hi > n_fine) hi = n_fine;
    // level-1 bucket boundaries are thread boundaries (F2 and the per-thread count are powers of two, F2 the larger)
    __shared__ unsigned long long cap_at[C2_MAXF + 1], raw_at[C2_MAXF + 1], bcap[C2_MAXF], padf[C2_MAXF];
    unsigned long long raw = 0;
    for (int i = lo; i < hi; i++) raw += ghist[i];
    unsigned long long total, total_raw;
    const unsigned long long raw0 = sp_block_excl_scan(raw, wsum, total_raw);
    if (lo < n_fine && lo % F2 == 0) raw_at[lo / F2] = raw0;
    if (threadIdx.x == 0) raw_at[F1] = total_raw;
    __syncthreads();
    // level-1 regions: a level-1 bucket is the union of F2 fine buckets and its relative sampling error is that much
    // smaller, so it gets its own, tighter, bound (estimate x 33/32 + slack1) instead of the sum of the fine regions
    // (the span of the level-1 planes is what part1's scattered bursts pay for: 22.7 -> 23.6 ms at 2.4x)
    if (threadIdx.x < F1) {
        const int b = threadIdx.x;
        const unsigned long long h = raw_at[b + 1] - raw_at[b];
        bcap[b] = (exact ? h : (h * mult8_1 + 7ULL) / 8ULL + slack1) + pad1;
        padf[b] = 3ULL * ((bcap[b] + C2_TILE_KEYS - 1) / C2_TILE_KEYS + 1ULL);   // part2 tiles of the bucket, at most
    }
    __syncthreads();
    auto cap = [&](int i) -> unsigned long long {
        const unsigned long long h = ghist[i];
        return ((exact ? h : (h * mult8 + 7ULL) / 8ULL + slack) + padf[i / F2] + 3ULL) & ~3ULL;
    };
    unsigned long long s = 0;
    for (int i = lo; i < hi; i++) s += cap(i);
    unsigned long long run = sp_block_excl_scan(s, wsum, total);
    if (lo < n_fine && lo % F2 == 0) cap_at[lo / F2] = run;
    if (threadIdx.x == 0) {
        off_fine[n_fine] = total;
        cap_at[F1] = total;
    }
    for (int i = lo; i < hi; i++) {
        off_fine[i] = run;
        run += cap(i);
    }
    __syncthreads();
    if (threadIdx.x < F1) {
        unsigned long long mine = bcap[threadIdx.x];
        const unsigned long long sum_fine = cap_at[threadIdx.x + 1] - cap_at[threadIdx.x];
        if (!exact && mine > sum_fine) mine = sum_fine;   // (more keys than that overrun level 2 anyway)
        bcap[threadIdx.x] = mine;
    }
    __syncthreads();
    {
        // sub-regions: an eighth of the bucket's bound each + their own slack (a local array of one repeat shorter
        // than a tile lands in ONE sub-region)
        const int V1 = F1 * C2_SPLIT;
        unsigned long long mine = 0;
        if ((int)threadIdx.x < V1) {
            const unsigned long long cb = bcap[threadIdx.x / C2_SPLIT];
            if (split > 1) mine = (cb + C2_SPLIT - 1) / C2_SPLIT + sslack;
            else mine = (threadIdx.x % C2_SPLIT) ? 0ULL : cb;
            mine = (mine + 3ULL) & ~3ULL;
        }
        unsigned long long tot1;
        const unsigned long long o = sp_block_excl_scan(mine, wsum, tot1);
        if ((int)threadIdx.x < V1) off1[threadIdx.x] = o;
        if (threadIdx.x == 0) off1[V1] = tot1;
    }
}

// After part1: the exact sizes of the level-1 sub-regions are its cursors.  off1[V1 + 1 + b] = end of the keys of
// sub-region b, tile_start[] = first part2 tile of every sub-region.  A bucket that outgrew its region (estimate mode only) had its
// surplus runs dropped by part1: the flag makes the host count the chromosome again with exact sizes.
__device__ __forceinline__ void c2_tiles_body(const unsigned long long *__restrict__ cursor1, int V1, unsigned long long *__restrict__ off1,
         unsigned long long *__restrict__ tile_start /* V1 + 1 */, unsigned long long *__restrict__ flag) {
    const int b = threadIdx.x;
    unsigned long long tc = 0;
    if (b < V1) {
        unsigned long long n = cursor1[C2_CUR1(b)];
        const unsigned long long cap = off1[b + 1] - off1[b];
        if (n > cap) {
            n = cap;
            atomicAdd(flag, 1ULL);
        }
        off1[V1 + 1 + b] = off1[b] + n;
        tc = (n + C2_TILE_KEYS - 1) / C2_TILE_KEYS;
    }
    __shared__ unsigned long long wsum[16];
    unsigned long long tot;
    const unsigned long long t0 = sp_block_excl_scan(tc, wsum, tot);
    if (b < V1) tile_start[b] = t0;
    if (b == 0) tile_start[V1] = tot;
}

// block-wide exclusive scan of hist[0..F) (F <= 256 <= blockDim) -> start[]; returns total
// LDSB: the two barriers order LDS traffic only (sp_barrier_lds): global stores of the previous tile stay in flight
template <bool LDSB = false>
__device__ __forceinline__ uint32_t c2_scan_F(const uint32_t *hist, uint32_t *start, int F,
                                              uint32_t *wsum /*>=4*/) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t v = (t < F) ? hist[t] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63 && wave < 4) wsum[wave] = incl;
    if (LDSB) sp_barrier_lds();
    else __syncthreads();
    uint32_t base = 0, total = 0;
    const int nwv = (int)(blockDim.x >> 6) < 4 ? (int)(blockDim.x >> 6) : 4;
    for (int w = 0; w < nwv; w++) {
        uint32_t s = wsum[w];
        if (w < wave) base += s;
        total += s;
    }
    if (t < F) start[t] = base + incl - v;
    if (LDSB) sp_barrier_lds();
    else __syncthreads();
    return total;
}

// ---------------------------------------------------------------- c2_part1
// LDS counting sort of a 16 K-key tile on the top B1 slot bits.  The tile's 32 slots per thread stay in registers
// (two blocks per CU leave 128 VGPRs per thread); the rank inside the bucket run is what the tile-histogram atomic
// returns; the run's global position is reserved with one global atomic per (tile, bucket) -- the order of the keys
// inside a level-1 bucket is irrelevant downstream.  Runs are padded to whole quads (C2_INVALID1 records) in LDS and
// in the bucket's region alike, and the copy-out moves a quad per lane: one 16-byte LDS read, ONE table entry
// (delta[b] = global base of the run - its LDS start, so that out index = delta[b] + i), an 8-byte store to the u16
// plane and a 4-byte store to the u8 plane.  LDS word of a record: bucket << 24 | its remaining T - B1 slot bits.
__device__ __forceinline__ uint32_t c2_pack_lo(uint32_t a, uint32_t b) {   // low halves of a and b
    return __builtin_amdgcn_perm(b, a, 0x05040100u);
}
__device__ __forceinline__ void c2_part1_body(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm,
          int64_t n_units /* of 32 starts */, sp_kparams32 kp, int shift1 /* T-B1 */, int F1, int split,
          const unsigned long long *__restrict__ off1, unsigned long long *__restrict__ cursor1, int64_t n_tiles,
          uint16_t *__restrict__ lo1, uint8_t *__restrict__ hi1) {
    __shared__ uint32_t hist[C2_MAXF], start[C2_MAXF], wsum[4];
    __shared__ unsigned long long delta[C2_MAXF];
    __shared__ __attribute__((aligned(16))) uint32_t keys[C2_P1_KEYS + C2_PADS];
    if (n_tiles <= 0 || n_units <= 0) return;      // (an empty chromosome of a batched launch: the prefetch below is unconditional)
    const int sh = 32 - 2 * kp.k;
    const uint32_t mask1 = (1u << shift1) - 1u;
    // the words of the NEXT tile's unit travel while this tile is sorted and written (copy to working registers, issue
    // the next loads, then work: loads issued after the work are waited for in full at the next use -- r03_notes.md)
    sp_words32 p_x;
    uint32_t p_nm0 = 0, p_nm1 = 0;
    auto fetch = [&](int64_t t) {
        // (unconditional, from a clamped unit: a load under a condition is merged with the old value through moves
        // that wait for it on the spot, which made the "prefetch" a plain load)
        int64_t u = (t < n_tiles ? t : n_tiles - 1) * C2_P1_THREADS + threadIdx.x;
        u = u < n_units ? u : n_units - 1;
        p_nm0 = nm[u];              // (u * 32) >> 5
        p_nm1 = nm[u + 1];
        p_x = sp_load_words32(pk, pm, u * C2_P1_UNIT);
    };
    // STORES THAT DRAIN BEHIND THE NEXT SCAN (round 5).  On this ISA loads and stores share one counter (vmcnt) and a
    // wait cannot leave the stores out, so wherever a prefetched register is first used the wave also waits for every
    // store it has in flight -- and __syncthreads() waits for them too.  With the prefetch consumed at the top of the
    // next tile, each wave sat out the round trip of its own copy-out stores once per tile, and since the blocks of a
    // CU start together and take equal steps, both were in that wait at the same time: the kernel was the SUM of its
    // scan (0.23 ms of VALU per 667-Mb chain, + 0.13 of rank atomics) and its 2.1 GB of stores, not their maximum
    // (bound variants: tools/bound_experiments.patch).  Now the words of the next tile are taken over (`take`: an empty asm that uses them,
    // which is where the compiler puts the wait) BEFORE the copy-out is issued -- the loads are a scan old by then --
    // and every barrier of the loop orders LDS traffic only: the stores drain while the next tile is scanned.
#ifndef C2_P1_ASYNC
#define C2_P1_ASYNC 1
#endif
    sp_words32 n_x;
    uint32_t n_nm0 = 0, n_nm1 = 0;
    auto take = [&]() {
#if C2_P1_ASYNC
        asm volatile("" : "+v"(p_x.l[0]), "+v"(p_x.l[1]), "+v"(p_x.l[2]), "+v"(p_x.m[0]), "+v"(p_x.m[1]), "+v"(p_x.m[2]),
                     "+v"(p_nm0), "+v"(p_nm1));
#endif
        n_x = p_x;
        n_nm0 = p_nm0;
        n_nm1 = p_nm1;
    };
    auto bar = [&]() {
#if C2_P1_ASYNC
        sp_barrier_lds();
#else
        __syncthreads();
#endif
    };
    fetch(blockIdx.x);
#if C2_P1_ASYNC
    take();
#endif
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (threadIdx.x < F1) hist[threadIdx.x] = 0;
        const int64_t u = tile * C2_P1_THREADS + threadIdx.x;
#if !C2_P1_ASYNC
        take();
#endif
        const sp_words32 x = n_x;
        const uint32_t c_nm0 = n_nm0, c_nm1 = n_nm1;
        fetch(tile + gridDim.x);
        bar();
        uint32_t slot[32], rank[32], ok = 0;
        if (u >= n_units) {      // (a thread without a unit: its slots index start[] below, unconditionally -- advisor r05)
#pragma unroll
            for (int j = 0; j < 32; j++) slot[j] = 0;
        }
        if (u < n_units) {
            ok = ~(uint32_t)sp_bad_from_words64((uint64_t)c_nm0 | ((uint64_t)c_nm1 << 32), kp.k);
            auto f = [&](int j, uint32_t V, uint32_t W) {
                slot[j] = kp.odd ? sp_slot_of32_t<true>(V >> sh, ~W & kp.kmask, kp)
                                 : sp_slot_of32_t<false>(V >> sh, ~W & kp.kmask, kp);
                if ((ok >> j) & 1u) rank[j] = atomicAdd(&hist[slot[j] >> shift1], 1u);
            };
            sp_win_loop<0, 1, decltype(f)>::run(x, f);
        }
        bar();
        unsigned long long at = 0, r_lo = 0, r_hi = 0;
        uint32_t c = 0, cpad = 0;
        if (threadIdx.x < F1) {
            c = hist[threadIdx.x];
            cpad = (c + 3u) & ~3u;
            hist[threadIdx.x] = cpad;     // (own entry: the scan below reads it from this thread)
            const int vb = (int)threadIdx.x * C2_SPLIT + (split > 1 ? (int)(tile & (C2_SPLIT - 1)) : 0);
            if (c) at = atomicAdd(&cursor1[C2_CUR1(vb)], (unsigned long long)cpad);
            r_lo = off1[vb];
            r_hi = off1[vb + 1];
            // (nothing that depends on the atomic's result before the scan below: the two waves that reserve would
            // sit out its round trip in front of a barrier the whole block waits at)
        }
        const uint32_t total = c2_scan_F<C2_P1_ASYNC != 0>(hist, start, F1, wsum);
        if (threadIdx.x < F1) {
            const unsigned long long g = r_lo + at;
            // estimate mode: a run that does not fit its bucket's region is dropped (c2_tiles sees the cursor and
            // raises the flag; the chromosome is then counted again from the exact histogram)
            const bool fits = at + cpad <= r_hi - r_lo;
            delta[threadIdx.x] = fits ? g - start[threadIdx.x] : C2_DROP;
            for (uint32_t i = c; i < ((c + 3u) & ~3u); i++) keys[start[threadIdx.x] + i] = ((uint32_t)threadIdx.x << 24) | C2_INVALID1;
        }
        // (the run starts of eight keys are read together: one LDS round trip per eight keys instead of one per key)
#pragma unroll
        for (int j0 = 0; j0 < 32; j0 += 8) {
            uint32_t st[8];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) st[jj] = start[slot[j0 + jj] >> shift1];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                const int j = j0 + jj;
                if ((ok >> j) & 1u) keys[st[jj] + rank[j]] = ((slot[j] >> shift1) << 24) | (slot[j] & mask1);
            }
        }
        bar();
#if C2_P1_ASYNC
        take();      // (the next tile's words: loaded a scan ago; no store of this wave is in flight here)
#endif
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
        // (four quads per round: their LDS reads, then their run bases, then their stores -- two LDS round trips per
        // four quads instead of two per quad)
        const uint32_t nq = total >> 2;
        for (uint32_t q0 = threadIdx.x; q0 < nq; q0 += 4u * C2_P1_THREADS) {
            uint4 v[4];
            unsigned long long d[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t q = q0 + (uint32_t)i * C2_P1_THREADS;
                v[i] = k4[q < nq ? q : nq - 1u];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) d[i] = delta[v[i].x >> 24];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t q = q0 + (uint32_t)i * C2_P1_THREADS;
                if (q >= nq || d[i] == C2_DROP) continue;
                const unsigned long long o = d[i] + 4ULL * q;      // a multiple of 4: region starts, reservations and LDS starts are
                *reinterpret_cast<uint2 *>(lo1 + o) = make_uint2(c2_pack_lo(v[i].x, v[i].y), c2_pack_lo(v[i].z, v[i].w));
                *reinterpret_cast<uint32_t *>(hi1 + o) = c2_pack_lo(__builtin_amdgcn_perm(v[i].y, v[i].x, 0x0c0c0602u),
                                                                    __builtin_amdgcn_perm(v[i].w, v[i].z, 0x0c0c0602u));
            }
        }
        bar();
    }
}

// ---------------------------------------------------------------- c2_part2
// Software-pipelined over the tiles a block processes: the keys of tile i+1 and the global cursor
// atomics of tile i are in flight while tile i is scanned / scattered / written (the kernel was
// 83 % SQ_WAIT_ANY without this).
__device__ __forceinline__ int c2_bucket_of(const unsigned long long *__restrict__ tile_start, int F1,
                                            unsigned long long tile) {
    int lo = 0, hi = F1;  // last b with tile_start[b] <= tile
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (tile_start[mid] <= tile) lo = mid;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void c2_part2_body(const uint16_t *__restrict__ lo1, const uint8_t *__restrict__ hi1, const unsigned long long *__restrict__ off1,
         const unsigned long long *__restrict__ tile_start, int F1 /* sub-regions of level 1 */, int F2, int shift2 /* B3 */,
         const unsigned long long *__restrict__ off_fine, unsigned long long *__restrict__ cursor2 /*n_fine*/,
         uint16_t *__restrict__ buf2) {
    __shared__ uint32_t hist[C2_MAXF], start[C2_MAXF], wsum[4];
    __shared__ unsigned long long gbase[C2_MAXF];
    __shared__ __attribute__((aligned(16))) uint32_t keys[C2_TILE_KEYS + C2_PADS];
    __shared__ int s_bucket[2];
    const unsigned long long n_tiles = tile_start[F1];
    const uint32_t mask2 = (uint32_t)F2 - 1u;
    // Tile ORDER: the blocks that run at the same time take tiles that lie n_tiles / C2_P2_SPREAD apart, i.e. in
    // different level-1 buckets, so that their cursor atomics go to different fine buckets' cursors (in bucket order
    // every resident block hammered the same F2 lines).  A bijection of [0, n_tiles): the last n_tiles mod SPREAD
    // tiles stay where they are.
    const unsigned long long perm_q = n_tiles / C2_P2_SPREAD, perm_n = perm_q * C2_P2_SPREAD;
    auto tile_of = [&](unsigned long long t) { return t < perm_n ? (t % C2_P2_SPREAD) * perm_q + t / C2_P2_SPREAD : t; };
    unsigned long long it = blockIdx.x;
    if (it >= n_tiles) return;
    unsigned long long tile = tile_of(it);
    if (threadIdx.x == 0) s_bucket[0] = c2_bucket_of(tile_start, F1, tile);
    __syncthreads();
    static_assert(C2_P2_PER % 4 == 0, "part2 reads its keys four at a time");
    constexpr int NG = C2_P2_PER / 4;
    uint2 nlo[NG];      // four 16-bit low parts
    uint32_t nhi[NG];   // four high bytes
    int nnext = 0;
    auto fetch = [&](int nb, unsigned long long t_in_bucket) {
        const unsigned long long base = off1[nb] + t_in_bucket * C2_TILE_KEYS, end = off1[F1 + 1 + nb];
        nnext = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const unsigned long long idx = base + ((unsigned long long)g * C2_P2_THREADS + threadIdx.x) * 4ULL;
            if (idx < end) {     // (bucket sizes are multiples of 4: whole quads)
                nlo[g] = *reinterpret_cast<const uint2 *>(lo1 + idx);
                nhi[g] = *reinterpret_cast<const uint32_t *>(hi1 + idx);
                nnext = 4 * g + 4;
            }
        }
    };
    fetch(s_bucket[0], tile - tile_start[s_bucket[0]]);
    int p = 0;
    for (; it < n_tiles; it += gridDim.x) {
        const int b1 = s_bucket[p] / C2_SPLIT;
        uint32_t my[C2_P2_PER];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            my[4 * g + 0] = (nlo[g].x & 0xFFFFu) | ((nhi[g] & 0xFFu) << 16);
            my[4 * g + 1] = (nlo[g].x >> 16) | ((nhi[g] & 0xFF00u) << 8);
            my[4 * g + 2] = (nlo[g].y & 0xFFFFu) | (nhi[g] & 0xFF0000u);
            my[4 * g + 3] = (nlo[g].y >> 16) | ((nhi[g] >> 24) << 16);
        }
        const int nmine = nnext;
        const bool more = it + gridDim.x < n_tiles;
        const unsigned long long ntile = more ? tile_of(it + gridDim.x) : n_tiles;
        if (threadIdx.x < F2) hist[threadIdx.x] = 0;
        if (threadIdx.x == 0 && ntile < n_tiles) s_bucket[p ^ 1] = c2_bucket_of(tile_start, F1, ntile);
        __syncthreads();  // (A) also: the previous tile's copy-out has finished reading keys/start/gbase
        uint32_t rank[C2_P2_PER];   // position of the key inside its bucket run = what the counting atomic returns
#pragma unroll
        for (int j = 0; j < C2_P2_PER; j++)
            if (j < nmine && my[j] != C2_INVALID1) rank[j] = atomicAdd(&hist[(my[j] >> shift2) & mask2], 1u);
        __syncthreads();  // (B)
        unsigned long long g = 0;
        bool fits = true;
        uint32_t c = 0;
        if (threadIdx.x < F2) {  // reserve the output ranges now; the result is needed only after the scan
            c = hist[threadIdx.x];
            const uint32_t cpad = (c + 3u) & ~3u;
            hist[threadIdx.x] = cpad;     // (own entry: the scan below reads it from this thread)
            const size_t fine = (size_t)b1 * F2 + threadIdx.x;
            const unsigned long long o0 = off_fine[fine];
            const unsigned long long at = c ? atomicAdd(&cursor2[fine * C2_CSTRIDE2], (unsigned long long)cpad) : 0ULL;
            g = o0 + at;
            fits = at + cpad <= off_fine[fine + 1] - o0;    // estimate mode: see part1; c2_spans raises the flag
        }
        nnext = 0;
        if (ntile < n_tiles) {  // next tile's keys: in flight across the rest of this iteration
            const int nb = s_bucket[p ^ 1];
            fetch(nb, ntile - tile_start[nb]);
        }
        const uint32_t total = c2_scan_F(hist, start, F2, wsum);
        if (threadIdx.x < F2) {
            gbase[threadIdx.x] = fits ? g - start[threadIdx.x] : C2_DROP;   // out index = gbase[b] + LDS index
            for (uint32_t i = c; i < ((c + 3u) & ~3u); i++) keys[start[threadIdx.x] + i] = ((uint32_t)threadIdx.x << 16) | C2_INVALID2;
        }
        __syncthreads();  // (C)
#pragma unroll
        for (int j = 0; j < C2_P2_PER; j++)
            if (j < nmine && my[j] != C2_INVALID1) {
                const uint32_t b = (my[j] >> shift2) & mask2;
                keys[start[b] + rank[j]] = (b << 16) | (my[j] & (C2_FINE - 1));
            }
        __syncthreads();  // (D)
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
        for (uint32_t q = threadIdx.x; q < (total >> 2); q += C2_P2_THREADS) {
            const uint4 v = k4[q];
            const unsigned long long gb = gbase[v.x >> 16];
            if (gb != C2_DROP)
                *reinterpret_cast<uint2 *>(buf2 + gb + 4ULL * q) = make_uint2(c2_pack_lo(v.x, v.y), c2_pack_lo(v.z, v.w));
        }
        p ^= 1;
    }
}

// After part2: the exact fine-bucket sizes are its cursors.  span[f] = [first, last) key of fine bucket f in buf2,
// written with plain stores so that c2_count's per-bucket look-up is one L2-resident 16-byte load (the cursors
// themselves were updated by atomics and live beyond L2: reading them inside c2_count's per-bucket prefetch
// cost 3.8 ms per wheat-like pass).  A bucket that outgrew its region (estimate mode) raises the flag.
__device__ __forceinline__ void c2_spans_body(const unsigned long long *__restrict__ off_fine, const unsigned long long *__restrict__ cursor2, int64_t n_fine,
         ulonglong2 *__restrict__ span, unsigned long long *__restrict__ flag, uint32_t list_div) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_fine) return;
    const unsigned long long lo = off_fine[f], cap = off_fine[f + 1] - lo;
    unsigned long long n = cursor2[(size_t)f * C2_CSTRIDE2];
    if (n > cap) {      // a dropped run leaves part of the region unwritten: nobody may read it (the chromosome is recounted)
        n = 0;
        atomicAdd(flag, 1ULL);
    }
    // engine 2: [first, last) key.  Engine 3 (list_div = lower_count): first key, then {number of keys, the bucket's
    // segment base floor(first / lower_count)} as two 32-bit halves -- the 64-bit division by a run-time divisor was a
    // third of c2_count_list's instructions when every wave of the workgroup did it per bucket
    span[f] = list_div ? make_ulonglong2(lo, (n & 0xffffffffULL) | ((lo / list_div) << 32)) : make_ulonglong2(lo, lo + n);
}

// ---------------------------------------------------------------- c2_count
// One workgroup per fine bucket: u32 counters in LDS, then ONE byte per slot goes to HBM (raw count,
// saturated at 255) -- a quarter of the u32 table the filter used to stream 21 times.  Counts >= 255 are
// rare; they are appended to an overflow list: the block reserves a contiguous segment for its bucket
// (one global atomic per bucket that has any) and records (segment base, count) per bucket, so that
// sp_ovf_finalize can lay the segments out in bucket order = ascending slot order without a device-wide sort.
#ifndef C2_COUNT_THREADS
#define C2_COUNT_THREADS 1024
#endif
#ifndef C2_PF
#define C2_PF 8         // prefetched 8-byte loads per thread (32 K keys per block: most of an average bucket)
#endif
#define C2_STAGE 1024   // overflow pairs staged in LDS per bucket (8 KiB next to the 128 KiB of counters)
__global__ void __launch_bounds__(C2_COUNT_THREADS)
c2_count(const uint16_t *__restrict__ buf2, const ulonglong2 *__restrict__ span /* [first, last) key of every fine bucket */,
         int64_t n_fine, const uint32_t *__restrict__ list /* the buckets to count (c2_count16 left them), or NULL: all */,
         const unsigned long long *__restrict__ n_list, uint32_t lower, uint8_t *__restrict__ tab,
         unsigned long long *__restrict__ out3 /*[0]=sum,[1]=n,[2]=overflow cursor,[3]=region-overrun flag*/,
         uint2 *__restrict__ ovf_tmp, unsigned long long ovf_cap, uint32_t *__restrict__ seg_base,
         uint32_t *__restrict__ seg_cnt) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];  // C2_FINE
    __shared__ unsigned long long red[16];
    __shared__ uint32_t s_nov, s_rank;
    __shared__ uint2 stage[C2_STAGE];
    unsigned long long s = 0, n = 0;
    const uint32_t thr = 255u;     // counts from here on go to the overflow list
    // Software pipeline over the buckets a block processes: the first C2_PF x 4 keys per thread of the NEXT
    // bucket are loaded while this bucket's counters are written out and cleared (one block per CU: nothing else
    // would hide those round trips; the kernel ran at 2.5 TB/s of its 4.6).
    uint2 pf[C2_PF];
    unsigned long long pf_lo = 0, pf_hi = 0;
    const int64_t n_it = list ? (int64_t)*n_list : n_fine;       // buckets to visit
    auto prefetch = [&](int64_t itn) {
        if (itn >= n_it) return;
        const ulonglong2 d = span[list ? (int64_t)list[itn] : itn];      // one 16-byte load: where the bucket's keys lie (c2_spans)
        pf_lo = d.x;
        pf_hi = d.y;
        unsigned long long a = (pf_lo + 3ULL) & ~3ULL;
        if (a > pf_hi) a = pf_hi;
        const unsigned long long n4 = (pf_hi - a) >> 2;
        const uint2 *p2 = reinterpret_cast<const uint2 *>(buf2 + a);
#pragma unroll
        for (int q = 0; q < C2_PF; q++) {
            const unsigned long long i = threadIdx.x + (unsigned long long)q * C2_COUNT_THREADS;
            if (i < n4) pf[q] = p2[i];
        }
    };
    prefetch(blockIdx.x);
    for (int64_t it = blockIdx.x; it < n_it; it += gridDim.x) {
        const int64_t fb = list ? (int64_t)list[it] : it;
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
        for (int i = threadIdx.x; i < C2_FINE / 4; i += C2_COUNT_THREADS) c4[i] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) s_nov = s_rank = 0;
        __syncthreads();
        const unsigned long long lo = pf_lo, hi = pf_hi;
        // 8-byte aligned body: four u16 keys per load
        unsigned long long a = (lo + 3ULL) & ~3ULL;
        if (a > hi) a = hi;
        // (a pad record, 0xFFFF, adds zero to the last counter: no branch)
        auto add = [&](uint32_t r) { atomicAdd(&cnt[r & (C2_FINE - 1)], (r >> C2_B3) ^ 1u); };
        for (unsigned long long i = lo + threadIdx.x; i < a; i += C2_COUNT_THREADS) add(buf2[i]);
        const unsigned long long n4 = (hi - a) >> 2;
        const uint2 *p2 = reinterpret_cast<const uint2 *>(buf2 + a);
#pragma unroll
        for (int q = 0; q < C2_PF; q++) {   // the prefetched part of the body
            if (threadIdx.x + (unsigned long long)q * C2_COUNT_THREADS < n4) {
                add(pf[q].x & 0xffffu);
                add(pf[q].x >> 16);
                add(pf[q].y & 0xffffu);
                add(pf[q].y >> 16);
            }
        }
        // the rest: four independent 8-byte loads in flight per lane, then the 16 LDS atomics
        unsigned long long i = threadIdx.x + (unsigned long long)C2_PF * C2_COUNT_THREADS;
        for (; i + 3ULL * C2_COUNT_THREADS < n4; i += 4ULL * C2_COUNT_THREADS) {
            uint2 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = p2[i + (unsigned long long)q * C2_COUNT_THREADS];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                add(v[q].x & 0xffffu);
                add(v[q].x >> 16);
                add(v[q].y & 0xffffu);
                add(v[q].y >> 16);
            }
        }
        for (; i < n4; i += C2_COUNT_THREADS) {
            uint2 v = p2[i];
            add(v.x & 0xffffu);
            add(v.x >> 16);
            add(v.y & 0xffffu);
            add(v.y >> 16);
        }
        for (unsigned long long t = a + (n4 << 2) + threadIdx.x; t < hi; t += C2_COUNT_THREADS) add(buf2[t]);
        prefetch(it + gridDim.x);   // in flight across the write-out below and the next clear
        __syncthreads();
        uint32_t *t32 = reinterpret_cast<uint32_t *>(tab + fb * C2_FINE);
        for (int i = threadIdx.x; i < C2_FINE / 4; i += C2_COUNT_THREADS) {
            const uint4 v = c4[i];
            const uint32_t a4[4] = {v.x, v.y, v.z, v.w};
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t c = a4[j];
                if (c >= lower) { s += c; n++; }
                if (c >= thr) {   // rare: staged in LDS; a bucket with more than C2_STAGE of them re-scans its counters
                    const uint32_t pos = atomicAdd(&s_nov, 1u);
                    if (pos < C2_STAGE) stage[pos] = make_uint2((uint32_t)(fb * C2_FINE + 4 * i + j), c);
                }
                packed |= (c < 255u ? c : 255u) << (8 * j);
            }
            t32[i] = packed;
        }
        __syncthreads();
        const uint32_t nov = s_nov;   // block-uniform
        if (nov) {
            // the bucket's segment of the staging list starts at floor(first key / 255): a slot that overflows accounts
            // for >= 255 keys, the key regions are disjoint, and floor(a / L) + floor(b / L) <= floor((a + b) / L) --
            // no cursor (one same-address global atomic per bucket: 16 K of them per launch)
            const unsigned long long base = lo / 255ULL;
            if (nov <= C2_STAGE) {
                for (uint32_t p = threadIdx.x; p < nov; p += C2_COUNT_THREADS)
                    if (base + p < ovf_cap) ovf_tmp[base + p] = stage[p];
            } else {
                for (int i = threadIdx.x; i < C2_FINE / 4; i += C2_COUNT_THREADS) {
                    const uint4 v = c4[i];
                    const uint32_t a4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (a4[j] >= thr) {
                            const unsigned long long pos = base + atomicAdd(&s_rank, 1u);
                            if (pos < ovf_cap) ovf_tmp[pos] = make_uint2((uint32_t)(fb * C2_FINE + 4 * i + j), a4[j]);
                        }
                }
            }
            if (threadIdx.x == 0) seg_base[fb] = (uint32_t)base;
        }
        if (threadIdx.x == 0) seg_cnt[fb] = nov;
        __syncthreads();
    }
    unsigned long long ts = sp_block_sum_u64(s, red);
    unsigned long long tn = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0) {
        if (ts) atomicAdd(&out3[0], ts);
        if (tn) atomicAdd(&out3[1], tn);
    }
}

// ---------------------------------------------------------------- c2_count16 (round 4)
// c2_count holds ONE 1024-thread block per CU (128 KiB of counters) whose phases -- count, write out + clear -- run one
// after the other with nothing else resident to fill the CU.  A fine bucket that receives fewer than 65536 records
// cannot push any counter past 16 bits, so its counters are packed two to a word (64 KiB): two blocks per CU, one
// counting while the other writes out.
// Round 6: the buckets with MORE records (repeat-rich ones: 10 % of the buckets of the wheat-like genome, and through c2_count
// 14 of the 38 ms the two kernels took per pass) are counted here as well -- a counter passes 16 bits only where ONE k-mer
// has 65536 copies in the chromosome.  Their adds look at what they got back: an add that finds its half at 0xFFFF wrapped it,
// the bucket's counters are thrown away and the bucket is counted again on 32-bit counters, half of its slots at a time, in the
// same 64 KiB.  c2_count stays as the cross-check (SP_C2_COUNT16=0) and for engine-3-less configurations that ask for it.
// Same outputs: byte table, overflow segments, tallies.  The write-out clears the counters it has read.
#ifndef C2_C16_THREADS
#define C2_C16_THREADS 512
#endif
#ifndef C2_C16_PF
#define C2_C16_PF 8
#endif
__global__ void __launch_bounds__(C2_C16_THREADS)
c2_count16(const uint16_t *__restrict__ buf2, const ulonglong2 *__restrict__ span, int64_t n_fine, uint32_t lower,
           uint8_t *__restrict__ tab, unsigned long long *__restrict__ out3, uint2 *__restrict__ ovf_tmp, unsigned long long ovf_cap,
           uint32_t *__restrict__ seg_base, uint32_t *__restrict__ seg_cnt) {
    __shared__ __attribute__((aligned(16))) uint32_t cnt[C2_FINE / 2];      // two 16-bit counters per word (a wrapped bucket: 2^14 32-bit counters)
    __shared__ unsigned long long red[16];
    __shared__ uint32_t s_nov, s_over;
    unsigned long long s = 0, n = 0;
    uint2 pf[C2_C16_PF];
    unsigned long long pf_lo = 0, pf_hi = 0;
    auto prefetch = [&](int64_t fbn) {
        if (fbn >= n_fine) return;
        const ulonglong2 d = span[fbn];
        pf_lo = d.x;                             // (a multiple of 4 records, like the size: whole quads)
        pf_hi = d.y;
        const unsigned long long n4 = (pf_hi - pf_lo) >> 2;
        const uint2 *p2 = reinterpret_cast<const uint2 *>(buf2 + pf_lo);
#pragma unroll
        for (int q = 0; q < C2_C16_PF; q++) {
            const unsigned long long i = threadIdx.x + (unsigned long long)q * C2_C16_THREADS;
            if (i < n4) pf[q] = p2[i];
        }
    };
    {
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
        for (int i = threadIdx.x; i < C2_FINE / 8; i += C2_C16_THREADS) c4[i] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) s_nov = s_over = 0;
    }
    prefetch(blockIdx.x);
    __syncthreads();
    for (int64_t fb = blockIdx.x; fb < n_fine; fb += gridDim.x) {
        // (round 6 audit: the tallies are cleared HERE, behind the barrier that ends the previous bucket.  s_nov used to be
        // cleared by thread 0 right after it had read `nov` below -- with no barrier between that store and the other waves' read
        // of s_nov, so a wave that left the barrier late could read 0 and skip its share of the staged overflow pairs.)
        if (threadIdx.x == 0) s_nov = s_over = 0;
        const unsigned long long lo = pf_lo, hi = pf_hi;
        const bool big = hi - lo >= 65536ULL;    // block-uniform: a counter of this bucket CAN pass 16 bits
        const unsigned long long n4 = (hi - lo) >> 2;
        const uint2 *p2 = reinterpret_cast<const uint2 *>(buf2 + lo);
        // the bucket's records: the prefetched ones, then the rest (an average bucket holds 2.5 x what the prefetch covers) in rounds
        // of C2_C16_PF loads per lane: one load per round trip -- what this loop was -- kept 8 KB in flight per CU, i.e. ~1.4 TB/s
        // for the whole chip at 1.5 us per trip, and that, not the LDS, was the kernel's rate (round 5)
        auto all_records = [&](auto add) {
#pragma unroll
            for (int q = 0; q < C2_C16_PF; q++) {
                if (threadIdx.x + (unsigned long long)q * C2_C16_THREADS < n4) {
                    add(pf[q].x & 0xffffu);
                    add(pf[q].x >> 16);
                    add(pf[q].y & 0xffffu);
                    add(pf[q].y >> 16);
                }
            }
            for (unsigned long long base = (unsigned long long)C2_C16_PF * C2_C16_THREADS; base < n4;
                 base += (unsigned long long)C2_C16_PF * C2_C16_THREADS) {
                uint2 v[C2_C16_PF];
#pragma unroll
                for (int q = 0; q < C2_C16_PF; q++) {
                    // (unconditional, from a clamped index: a load under a condition is followed by a wait of its own)
                    const unsigned long long i = base + threadIdx.x + (unsigned long long)q * C2_C16_THREADS;
                    v[q] = p2[i < n4 ? i : n4 - 1ULL];
                }
#pragma unroll
                for (int q = 0; q < C2_C16_PF; q++) {
                    if (base + threadIdx.x + (unsigned long long)q * C2_C16_THREADS < n4) {
                        add(v[q].x & 0xffffu);
                        add(v[q].x >> 16);
                        add(v[q].y & 0xffffu);
                        add(v[q].y >> 16);
                    }
                }
            }
        };
        // a record adds 1 to its half of the word; a pad record (0xFFFF) adds 0 to the last word
        if (!big) {
            all_records([&](uint32_t r) { atomicAdd(&cnt[(r & (C2_FINE - 1)) >> 1], ((r >> C2_B3) ^ 1u) << (16u * (r & 1u))); });
        } else {
            uint32_t wrapped = 0;
            all_records([&](uint32_t r) {
                const uint32_t sh = 16u * (r & 1u), inc = ((r >> C2_B3) ^ 1u) << sh;
                const uint32_t old = atomicAdd(&cnt[(r & (C2_FINE - 1)) >> 1], inc);
                wrapped |= (inc != 0u && ((old >> sh) & 0xffffu) == 0xffffu) ? 1u : 0u;
            });
            if (wrapped) s_over = 1u;
        }
        prefetch(fb + gridDim.x);   // in flight across the write-out below
        __syncthreads();
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
        const unsigned long long seg = lo / 255ULL;
        if (big && s_over) {
            // (block-uniform) a counter wrapped -- one k-mer with 65536 copies or more in this chromosome: the bucket again, on
            // 32-bit counters, HALF of its slots at a time in the same 64 KiB (nothing is pipelined here: a handful of buckets per
            // genome).  Until round 6 such buckets -- and every bucket of 65536 records or more -- went to c2_count, whose
            // 128 KiB of counters need a CU to themselves: with four chains side by side its workgroups mostly waited for one.
            for (int i = threadIdx.x; i < C2_FINE / 8; i += C2_C16_THREADS) c4[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
            for (uint32_t h = 0; h < 2u; h++) {
                for (unsigned long long i = threadIdx.x; i < n4; i += C2_C16_THREADS) {
                    const uint2 v = p2[i];
                    const uint32_t r4[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (!(r4[q] >> C2_B3) && ((r4[q] >> (C2_B3 - 1)) & 1u) == h) atomicAdd(&cnt[r4[q] & (C2_FINE / 2 - 1)], 1u);
                }
                __syncthreads();
                uint32_t *t32 = reinterpret_cast<uint32_t *>(tab + fb * C2_FINE + (int64_t)h * (C2_FINE / 2));
                for (int i = threadIdx.x; i < C2_FINE / 8; i += C2_C16_THREADS) {
                    const uint4 v = c4[i];
                    c4[i] = make_uint4(0, 0, 0, 0);
                    const uint32_t a4[4] = {v.x, v.y, v.z, v.w};
                    uint32_t packed = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t c = a4[j];
                        if (c >= lower) { s += c; n++; }
                        if (c >= 255u) {
                            const unsigned long long pos = seg + atomicAdd(&s_nov, 1u);
                            if (pos < ovf_cap) ovf_tmp[pos] = make_uint2((uint32_t)(fb * C2_FINE + h * (C2_FINE / 2) + 4 * i + j), c);
                        }
                        packed |= (c < 255u ? c : 255u) << (8 * j);
                    }
                    t32[i] = packed;
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                const uint32_t nov = s_nov;
                if (nov) seg_base[fb] = (uint32_t)seg;
                seg_cnt[fb] = nov;
            }
            __syncthreads();
            continue;
        }
        // write-out: 8 slots per 16-byte LDS read -> 8 table bytes; the words are cleared for the next bucket.  A saturated slot's
        // (slot, count) pair goes straight to the bucket's segment of the staging list (c2_count: its closed-form place
        // floor(first record / 255) -- a slot that overflows accounts for >= 255 records, so the segment holds them whatever the
        // bucket's size; ovf_place ranks a segment by slot afterwards, the order of arrival does not matter)
        uint2 *t64 = reinterpret_cast<uint2 *>(tab + fb * C2_FINE);
        for (int i = threadIdx.x; i < C2_FINE / 8; i += C2_C16_THREADS) {
            const uint4 v = c4[i];
            c4[i] = make_uint4(0, 0, 0, 0);
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
            uint32_t packed[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t c = (w4[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                if (c >= lower) { s += c; n++; }
                if (c >= 255u) {
                    const unsigned long long pos = seg + atomicAdd(&s_nov, 1u);
                    if (pos < ovf_cap) ovf_tmp[pos] = make_uint2((uint32_t)(fb * C2_FINE + 8 * i + j), c);
                }
                packed[j >> 2] |= (c < 255u ? c : 255u) << (8 * (j & 3));
            }
            t64[i] = make_uint2(packed[0], packed[1]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t nov = s_nov;
            if (nov) seg_base[fb] = (uint32_t)seg;
            seg_cnt[fb] = nov;
        }
        __syncthreads();
    }
    unsigned long long ts = sp_block_sum_u64(s, red);
    unsigned long long tn = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0) {
        if (ts) atomicAdd(&out3[0], ts);
        if (tn) atomicAdd(&out3[1], tn);
    }
}

// ---------------------------------------------------------------- c2_count_list (engine 3)
// Small genomes: a fine bucket of 2^15 slots receives a few thousand keys (1.2 K on a 20-Mb chromosome at k = 15), and
// clearing + scanning 128 KiB of LDS counters per bucket -- what the table kernel above does -- was the whole cost
// (0.33 ms per chromosome whatever its length).  Here the work per bucket is O(keys): the keys stay in registers,
// pass 1 counts them, pass 2 lets every key swap its counter for zero -- the one lane that gets the count back owns
// the slot, emits (slot, count) if count >= lower, and the counters are clean again for the next bucket.  No table is
// written.  The pairs of a bucket go to one contiguous segment of the staging list, placed at floor(first key / lower):
// a kept slot accounts for >= lower keys, the key regions are disjoint and floor(a / L) + floor(b / L) <=
// floor((a + b) / L) -- no cursor (one same-address global atomic per bucket is 16 K of them per launch: 0.2 ms, and
// it was most of this kernel).  ovf_scan / ovf_place<SPLIT> then lay the segments out in bucket order, ranked by slot
// inside a bucket.  Buckets with more keys than the registers hold (C2L_PF quads per thread) read the rest from
// memory and walk the counters in pass 2.
// What is left is one memory round trip per bucket (0.14 ms per 20-Mb chromosome): the next bucket's keys are
// requested one bucket ahead, and deeper register pipelines do not help -- the compiler cannot count the conditional
// loads and stores in flight and waits for all of them (vmcnt(0)) at the first use; four 512-thread blocks per CU on
// quarter buckets (32-KiB counters, the keys walked four times) were measured at twice the time
// (profiles/r03_notes.md).
#ifndef C2L_DEPTH
#define C2L_DEPTH 4      // buckets per group (one memory round trip per group)
#endif
#ifndef C2L_PF
#define C2L_PF 2         // 8-byte loads per thread and bucket held in registers (8 K keys per bucket)
#endif
#define C2L_MAXB 256      // fine buckets per block at most (the launch sizes the grid accordingly)
template <int C2L_DEPTH_, int C2L_PF_>
__device__ __forceinline__ void c2_count_list_body(const uint16_t *__restrict__ buf2, const ulonglong2 *__restrict__ span, int64_t n_fine, uint32_t lower,
              unsigned long long *__restrict__ out3 /*[0]=sum,[1]=n*/,
              uint2 *__restrict__ stage, unsigned long long stage_cap, uint32_t *__restrict__ seg_base,
              uint32_t *__restrict__ seg_cnt) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];  // C2_FINE
    __shared__ unsigned long long red[16];
    __shared__ uint32_t s_nov[2];
    __shared__ ulonglong2 s_span[C2L_MAXB];
    unsigned long long s = 0, n = 0;
    {
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
        for (int i = threadIdx.x; i < C2_FINE / 4; i += C2_COUNT_THREADS) c4[i] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < 2) s_nov[threadIdx.x] = 0;
        // the spans of this block's buckets: one read up front, so that a prefetch is ONE memory round trip (keys)
        for (int64_t j = threadIdx.x, fbn = blockIdx.x + (int64_t)threadIdx.x * gridDim.x; j < C2L_MAXB && fbn < n_fine;
             j += C2_COUNT_THREADS, fbn += (int64_t)C2_COUNT_THREADS * gridDim.x)
            s_span[j] = span[fbn];
    }
    __syncthreads();
    // per prefetched bucket: aligned address of its first quad, offset of the first key in that quad (0..3), number of
    // keys, segment base.  Everything inside a bucket is 32-bit arithmetic (a region holds < 2^32 keys: host check).
    uint2 pf[C2L_DEPTH_][C2L_PF_];
    unsigned long long p_a0[C2L_DEPTH_];
    uint32_t p_off[C2L_DEPTH_], p_n[C2L_DEPTH_], p_base[C2L_DEPTH_];
    auto issue = [&](int d, int64_t fbn, int64_t j) {      // d is a constant after unrolling: the arrays live in registers
        p_a0[d] = 0;
        p_off[d] = p_n[d] = p_base[d] = 0;
        if (fbn >= n_fine) return;
        const ulonglong2 sp = s_span[j];
        p_a0[d] = sp.x & ~3ULL;            // whole quads from the aligned address below the first key
        p_off[d] = (uint32_t)(sp.x & 3ULL);
        p_n[d] = (uint32_t)sp.y;
        p_base[d] = (uint32_t)(sp.y >> 32);
        const uint32_t nq = (p_off[d] + p_n[d] + 3u) >> 2;
        const uint2 *p2 = reinterpret_cast<const uint2 *>(buf2 + p_a0[d]);
        // (kept under their condition: unconditional clamped loads -- what helped c2_count16 -- cost the list counter
        // 12 % on 20-Mb chromosomes, where most lanes have nothing to load; round 5)
#pragma unroll
        for (int q = 0; q < C2L_PF_; q++) {
            const uint32_t i = threadIdx.x + (uint32_t)q * C2_COUNT_THREADS;
            if (i < nq) pf[d][q] = p2[i];
        }
    };
    // f(residual) for the valid keys of quad i of a bucket whose keys are [off, end) counted from its first quad
    auto quad = [&](const uint2 v, uint32_t i, uint32_t off, uint32_t end, auto &&f) {
        const uint32_t k0 = 4u * i;
        // (pad records, 0xFFFF, are skipped: bit 15 is set in no residual)
        if (k0 + 0 >= off && k0 + 0 < end && !(v.x & 0x8000u)) f(v.x & 0xffffu);
        if (k0 + 1 >= off && k0 + 1 < end && !(v.x & 0x80000000u)) f(v.x >> 16);
        if (k0 + 2 >= off && k0 + 2 < end && !(v.y & 0x8000u)) f(v.y & 0xffffu);
        if (k0 + 3 >= off && k0 + 3 < end && !(v.y & 0x80000000u)) f(v.y >> 16);
    };
#pragma unroll
    for (int d = 0; d < C2L_DEPTH_; d++) issue(d, (int64_t)blockIdx.x + (int64_t)d * gridDim.x, d);
    int par = 0;
    int64_t j0 = 0;
    // one bucket, full 32-bit counters (any size)
    uint2 cur[C2L_DEPTH_][C2L_PF_];
    unsigned long long c_a0[C2L_DEPTH_];
    uint32_t c_off[C2L_DEPTH_], c_n[C2L_DEPTH_], c_base[C2L_DEPTH_];
    auto single = [&](int d, int64_t fb) {
        const uint32_t off = c_off[d], end = c_off[d] + c_n[d], nq = (end + 3u) >> 2;
        const bool in_regs = nq <= (uint32_t)C2L_PF_ * C2_COUNT_THREADS;      // block-uniform
        const uint32_t slot0 = (uint32_t)(fb * C2_FINE);
        const unsigned long long base = c_base[d];
        const uint2 *p2 = reinterpret_cast<const uint2 *>(buf2 + c_a0[d]);
        auto add = [&](uint32_t r) { atomicAdd(&cnt[r], 1u); };
        // ---- pass 1: count
#pragma unroll
        for (int q = 0; q < C2L_PF_; q++) {
            const uint32_t i = threadIdx.x + (uint32_t)q * C2_COUNT_THREADS;
            if (i < nq) quad(cur[d][q], i, off, end, add);
        }
        if (!in_regs)      // (four clamped loads per round: one load per round trip held this loop at a quarter of its rate)
            for (uint32_t i0 = threadIdx.x + (uint32_t)C2L_PF_ * C2_COUNT_THREADS; i0 < nq; i0 += 4u * C2_COUNT_THREADS) {
                uint2 vv[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t i = i0 + (uint32_t)t * C2_COUNT_THREADS;
                    vv[t] = p2[i < nq ? i : nq - 1u];
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t i = i0 + (uint32_t)t * C2_COUNT_THREADS;
                    if (i < nq) quad(vv[t], i, off, end, add);
                }
            }
        sp_barrier_lds();   // (A) counts complete
        auto emit = [&](uint32_t r, uint32_t c) {
            if (c >= lower) {
                s += c;
                n++;
                const unsigned long long pos = base + atomicAdd(&s_nov[par], 1u);
                if (pos < stage_cap) stage[pos] = make_uint2(slot0 + r, c);
            }
        };
        // ---- pass 2: every key swaps its counter for zero; the lane that gets the count back owns the slot
        if (in_regs) {
            auto take = [&](uint32_t r) {
                const uint32_t c = atomicExch(&cnt[r], 0u);
                if (c) emit(r, c);
            };
#pragma unroll
            for (int q = 0; q < C2L_PF_; q++) {
                const uint32_t i = threadIdx.x + (uint32_t)q * C2_COUNT_THREADS;
                if (i < nq) quad(cur[d][q], i, off, end, take);
            }
        } else {           // a crowded bucket: walk the counters instead of the keys
            uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
            for (int i = threadIdx.x; i < C2_FINE / 4; i += C2_COUNT_THREADS) {
                const uint4 v = c4[i];
                if (v.x | v.y | v.z | v.w) {
                    c4[i] = make_uint4(0, 0, 0, 0);
                    emit(4 * i + 0, v.x);
                    emit(4 * i + 1, v.y);
                    emit(4 * i + 2, v.z);
                    emit(4 * i + 3, v.w);
                }
            }
        }
        if (threadIdx.x == 0) s_nov[par ^ 1] = 0;     // the other parity's tally was read after the previous (B)
        sp_barrier_lds();   // (B) counters clean, tally final
        if (threadIdx.x == 0) {
            seg_cnt[fb] = s_nov[par];
            seg_base[fb] = (uint32_t)base;
        }
        par ^= 1;
    };
    // TWO buckets of fewer than 65536 keys at once: they share the counters, 16 bits each (a count cannot exceed the
    // number of keys of its bucket, so neither half carries into the other; pass 2 clears its own half with an atomic
    // AND).  The chain of dependent LDS round trips and the two barriers are paid once per PAIR.
    auto pair = [&](int d, int64_t fbA, int64_t fbB) {
        const uint32_t offA = c_off[d], endA = c_off[d] + c_n[d], nqA = (endA + 3u) >> 2;
        const uint32_t offB = c_off[d + 1], endB = c_off[d + 1] + c_n[d + 1], nqB = (endB + 3u) >> 2;
        const uint32_t slotA = (uint32_t)(fbA * C2_FINE), slotB = (uint32_t)(fbB * C2_FINE);
        const unsigned long long baseA = c_base[d], baseB = c_base[d + 1];
        auto addA = [&](uint32_t r) { atomicAdd(&cnt[r], 1u); };
        auto addB = [&](uint32_t r) { atomicAdd(&cnt[r], 0x10000u); };
#pragma unroll
        for (int q = 0; q < C2L_PF_; q++) {
            const uint32_t i = threadIdx.x + (uint32_t)q * C2_COUNT_THREADS;
            if (i < nqA) quad(cur[d][q], i, offA, endA, addA);
            if (i < nqB) quad(cur[d + 1][q], i, offB, endB, addB);
        }
        sp_barrier_lds();   // (A) counts complete
        // tallies: the low half of s_nov[par] counts A's pairs, the high half B's
        auto takeA = [&](uint32_t r) {
            const uint32_t c = atomicAnd(&cnt[r], 0xffff0000u) & 0xffffu;
            if (c >= lower) {      // (c != 0: this lane owns the slot)
                s += c;
                n++;
                const unsigned long long pos = baseA + (atomicAdd(&s_nov[par], 1u) & 0xffffu);
                if (pos < stage_cap) stage[pos] = make_uint2(slotA + r, c);
            }
        };
        auto takeB = [&](uint32_t r) {
            const uint32_t c = atomicAnd(&cnt[r], 0x0000ffffu) >> 16;
            if (c >= lower) {
                s += c;
                n++;
                const unsigned long long pos = baseB + (atomicAdd(&s_nov[par], 0x10000u) >> 16);
                if (pos < stage_cap) stage[pos] = make_uint2(slotB + r, c);
            }
        };
#pragma unroll
        for (int q = 0; q < C2L_PF_; q++) {
            const uint32_t i = threadIdx.x + (uint32_t)q * C2_COUNT_THREADS;
            if (i < nqA) quad(cur[d][q], i, offA, endA, takeA);
            if (i < nqB) quad(cur[d + 1][q], i, offB, endB, takeB);
        }
        if (threadIdx.x == 0) s_nov[par ^ 1] = 0;
        sp_barrier_lds();   // (B) counters clean, tallies final
        if (threadIdx.x == 0) {
            const uint32_t t = s_nov[par];
            seg_cnt[fbA] = t & 0xffffu;
            seg_base[fbA] = (uint32_t)baseA;
            seg_cnt[fbB] = t >> 16;
            seg_base[fbB] = (uint32_t)baseB;
        }
        par ^= 1;
    };
    static_assert(C2L_DEPTH_ % 2 == 0, "buckets are taken in pairs");
    // A GROUP of C2L_DEPTH_ buckets per iteration.  Their keys move to `cur` FIRST (the compiler waits for every load in
    // flight at the first use of a loaded register -- vmcnt(0): it cannot count the conditional loads), THEN the loads
    // of the next group are issued, THEN the group is processed bucket by bucket: the wait at the top of the next
    // iteration finds loads that are a whole group old, and one memory round trip is paid per group, not per bucket
    // (per bucket, issued after the work: 1.75 ms per Arabidopsis-like pass; copy-issue-work: 1.49; groups of 4: see
    // profiles/r03_notes.md).
    for (int64_t fb0 = blockIdx.x; fb0 < n_fine; fb0 += (int64_t)C2L_DEPTH_ * gridDim.x, j0 += C2L_DEPTH_) {
#pragma unroll
        for (int d = 0; d < C2L_DEPTH_; d++) {
#pragma unroll
            for (int q = 0; q < C2L_PF_; q++) cur[d][q] = pf[d][q];
            c_a0[d] = p_a0[d];
            c_off[d] = p_off[d];
            c_n[d] = p_n[d];
            c_base[d] = p_base[d];
        }
#pragma unroll
        for (int d = 0; d < C2L_DEPTH_; d++)
            issue(d, fb0 + (int64_t)(d + C2L_DEPTH_) * gridDim.x, j0 + d + C2L_DEPTH_);
#pragma unroll
        for (int d = 0; d < C2L_DEPTH_; d += 2) {
            const int64_t fbA = fb0 + (int64_t)d * gridDim.x, fbB = fbA + gridDim.x;
            if (fbA >= n_fine) break;           // block-uniform
            const uint32_t lim = (uint32_t)C2L_PF_ * C2_COUNT_THREADS * 4u - 8u;
            if (fbB < n_fine && c_n[d] < 65536u && c_n[d + 1] < 65536u && c_n[d] <= lim && c_n[d + 1] <= lim) {
                pair(d, fbA, fbB);
            } else {
                single(d, fbA);
                if (fbB < n_fine) single(d + 1, fbB);
            }
        }
    }
    unsigned long long ts = sp_block_sum_u64(s, red);
    unsigned long long tn = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0) {
        if (ts) atomicAdd(&out3[0], ts);
        if (tn) atomicAdd(&out3[1], tn);
    }
}

// ---------------------------------------------------------------- c2_count_list16 (round 6)
// c2_count_list holds ONE 1024-thread workgroup per CU (128 KiB of counters), and on a 20-Mb chromosome a fine bucket is ~1.2 K keys:
// two barriers and a chain of dependent LDS round trips per bucket (or pair of buckets) with nothing else resident to fill
// the CU.  A bucket of fewer than 65536 keys cannot push a counter past 16 bits, so its counters are packed two to a word
// (64 KiB) and TWO 512-thread workgroups share a CU, one in its barrier while the other works -- the step c2_count16 took for
// the table engine in round 4.  One bucket at a time (no pairs: the halves of a word are two SLOTS here), groups of four
// buckets per memory round trip as before.  A bucket of 65536 keys or more (a hot repeat) raises the chromosome's overrun
// flag: the host counts that chromosome again with exact sizes, which takes the 32-bit kernel above.
#define C2L16_THREADS 512
#ifndef C2L16_DEPTH
#define C2L16_DEPTH 4
#endif
#ifndef C2L16_PF
#define C2L16_PF 2       // quads per thread and bucket held in registers: 4096 keys
#endif
#ifndef C2L16_MAXLEN
#define C2L16_MAXLEN (1LL << 26)      // longer chromosomes (~8 K keys per bucket and more): the BIG geometry
#endif
template <int C2L16_DEPTH_, int C2L16_PF_>
__device__ __forceinline__ void c2_count_list16_body(const uint16_t *__restrict__ buf2, const ulonglong2 *__restrict__ span, int64_t n_fine,
              uint32_t lower, unsigned long long *__restrict__ out3 /*[0]=sum,[1]=n,[3]=overrun flag*/,
              uint2 *__restrict__ stage, unsigned long long stage_cap, uint32_t *__restrict__ seg_base,
              uint32_t *__restrict__ seg_cnt) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];  // C2_FINE / 2 words: slot r in half (r & 1) of word r >> 1
    __shared__ unsigned long long red[16];
    __shared__ uint32_t s_nov[2];
    __shared__ ulonglong2 s_span[C2L_MAXB];
    unsigned long long s = 0, n = 0;
    {
        uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
        for (int i = threadIdx.x; i < C2_FINE / 8; i += C2L16_THREADS) c4[i] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < 2) s_nov[threadIdx.x] = 0;
        for (int64_t j = threadIdx.x, fbn = blockIdx.x + (int64_t)threadIdx.x * gridDim.x; j < C2L_MAXB && fbn < n_fine;
             j += C2L16_THREADS, fbn += (int64_t)C2L16_THREADS * gridDim.x)
            s_span[j] = span[fbn];
    }
    __syncthreads();
    uint2 pf[C2L16_DEPTH_][C2L16_PF_];
    unsigned long long p_a0[C2L16_DEPTH_];
    uint32_t p_off[C2L16_DEPTH_], p_n[C2L16_DEPTH_], p_base[C2L16_DEPTH_];
    auto issue = [&](int d, int64_t fbn, int64_t j) {
        p_a0[d] = 0;
        p_off[d] = p_n[d] = p_base[d] = 0;
        if (fbn >= n_fine) return;
        const ulonglong2 sp = s_span[j];
        p_a0[d] = sp.x & ~3ULL;
        p_off[d] = (uint32_t)(sp.x & 3ULL);
        p_n[d] = (uint32_t)sp.y;
        p_base[d] = (uint32_t)(sp.y >> 32);
        if (p_n[d] >= 65536u) return;      // (not this kernel's: no loads)
        const uint32_t nq = (p_off[d] + p_n[d] + 3u) >> 2;
        const uint2 *p2 = reinterpret_cast<const uint2 *>(buf2 + p_a0[d]);
#pragma unroll
        for (int q = 0; q < C2L16_PF_; q++) {
            const uint32_t i = threadIdx.x + (uint32_t)q * C2L16_THREADS;
            if (i < nq) pf[d][q] = p2[i];
        }
    };
    auto quad = [&](const uint2 v, uint32_t i, uint32_t off, uint32_t end, auto &&f) {
        const uint32_t k0 = 4u * i;
        if (k0 + 0 >= off && k0 + 0 < end && !(v.x & 0x8000u)) f(v.x & 0xffffu);
        if (k0 + 1 >= off && k0 + 1 < end && !(v.x & 0x80000000u)) f(v.x >> 16);
        if (k0 + 2 >= off && k0 + 2 < end && !(v.y & 0x8000u)) f(v.y & 0xffffu);
        if (k0 + 3 >= off && k0 + 3 < end && !(v.y & 0x80000000u)) f(v.y >> 16);
    };
#pragma unroll
    for (int d = 0; d < C2L16_DEPTH_; d++) issue(d, (int64_t)blockIdx.x + (int64_t)d * gridDim.x, d);
    int par = 0;
    int64_t j0 = 0;
    uint2 cur[C2L16_DEPTH_][C2L16_PF_];
    unsigned long long c_a0[C2L16_DEPTH_];
    uint32_t c_off[C2L16_DEPTH_], c_n[C2L16_DEPTH_], c_base[C2L16_DEPTH_];
    auto one = [&](int d, int64_t fb) {
        if (c_n[d] >= 65536u) {      // block-uniform: the chromosome is counted again by the 32-bit kernel
            if (threadIdx.x == 0) {
                atomicAdd(&out3[3], 1ULL);
                seg_cnt[fb] = 0;
            }
            return;
        }
        const uint32_t off = c_off[d], end = c_off[d] + c_n[d], nq = (end + 3u) >> 2;
        const bool in_regs = nq <= (uint32_t)C2L16_PF_ * C2L16_THREADS;      // block-uniform
        const uint32_t slot0 = (uint32_t)(fb * C2_FINE);
        const unsigned long long base = c_base[d];
        const uint2 *p2 = reinterpret_cast<const uint2 *>(buf2 + c_a0[d]);
        auto add = [&](uint32_t r) { atomicAdd(&cnt[r >> 1], 1u << (16u * (r & 1u))); };
        // ---- pass 1: count
#pragma unroll
        for (int q = 0; q < C2L16_PF_; q++) {
            const uint32_t i = threadIdx.x + (uint32_t)q * C2L16_THREADS;
            if (i < nq) quad(cur[d][q], i, off, end, add);
        }
        if (!in_regs)
            for (uint32_t i0 = threadIdx.x + (uint32_t)C2L16_PF_ * C2L16_THREADS; i0 < nq; i0 += 4u * C2L16_THREADS) {
                uint2 vv[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t i = i0 + (uint32_t)t * C2L16_THREADS;
                    vv[t] = p2[i < nq ? i : nq - 1u];
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t i = i0 + (uint32_t)t * C2L16_THREADS;
                    if (i < nq) quad(vv[t], i, off, end, add);
                }
            }
        sp_barrier_lds();   // (A) counts complete
        auto emit = [&](uint32_t r, uint32_t c) {
            if (c >= lower) {
                s += c;
                n++;
                const unsigned long long pos = base + atomicAdd(&s_nov[par], 1u);
                if (pos < stage_cap) stage[pos] = make_uint2(slot0 + r, c);
            }
        };
        // ---- pass 2: every key clears its half of its word; the one lane that gets the count back owns the slot
        if (in_regs) {
            auto take = [&](uint32_t r) {
                const uint32_t sh = 16u * (r & 1u);
                const uint32_t c = (atomicAnd(&cnt[r >> 1], ~(0xffffu << sh)) >> sh) & 0xffffu;
                if (c) emit(r, c);
            };
#pragma unroll
            for (int q = 0; q < C2L16_PF_; q++) {
                const uint32_t i = threadIdx.x + (uint32_t)q * C2L16_THREADS;
                if (i < nq) quad(cur[d][q], i, off, end, take);
            }
        } else {           // a crowded bucket: walk the counters instead of the keys
            uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
            for (int i = threadIdx.x; i < C2_FINE / 8; i += C2L16_THREADS) {
                const uint4 v = c4[i];
                if (v.x | v.y | v.z | v.w) {
                    c4[i] = make_uint4(0, 0, 0, 0);
                    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        emit(8 * i + 2 * q, w4[q] & 0xffffu);
                        emit(8 * i + 2 * q + 1, w4[q] >> 16);
                    }
                }
            }
        }
        if (threadIdx.x == 0) s_nov[par ^ 1] = 0;     // the other parity's tally was read after the previous (B)
        sp_barrier_lds();   // (B) counters clean, tally final
        if (threadIdx.x == 0) {
            seg_cnt[fb] = s_nov[par];
            seg_base[fb] = (uint32_t)base;
        }
        par ^= 1;
    };
    for (int64_t fb0 = blockIdx.x; fb0 < n_fine; fb0 += (int64_t)C2L16_DEPTH_ * gridDim.x, j0 += C2L16_DEPTH_) {
#pragma unroll
        for (int d = 0; d < C2L16_DEPTH_; d++) {
#pragma unroll
            for (int q = 0; q < C2L16_PF_; q++) cur[d][q] = pf[d][q];
            c_a0[d] = p_a0[d];
            c_off[d] = p_off[d];
            c_n[d] = p_n[d];
            c_base[d] = p_base[d];
        }
#pragma unroll
        for (int d = 0; d < C2L16_DEPTH_; d++)
            issue(d, fb0 + (int64_t)(d + C2L16_DEPTH_) * gridDim.x, j0 + d + C2L16_DEPTH_);
#pragma unroll
        for (int d = 0; d < C2L16_DEPTH_; d++) {
            const int64_t fb = fb0 + (int64_t)d * gridDim.x;
            if (fb >= n_fine) break;           // block-uniform
            one(d, fb);
        }
    }
    unsigned long long ts = sp_block_sum_u64(s, red);
    unsigned long long tn = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0) {
        if (ts) atomicAdd(&out3[0], ts);
        if (tn) atomicAdd(&out3[1], tn);
    }
}

// ---------------------------------------------------------------- kernels: one chromosome, or one per blockIdx.y
__global__ void __launch_bounds__(C2_P1_THREADS)
c2_hist_fine(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm,
             int64_t n_units, int64_t n_visit, int sample_shift, sp_kparams32 kp, int shift_fine, int n_fine,
             unsigned long long *__restrict__ ghist) {
    c2_hist_fine_body(pk, pm, nm, n_units, n_visit, sample_shift, kp, shift_fine, n_fine, ghist);
}
__global__ void __launch_bounds__(C2_P1_THREADS)
c2_hist_fine_b(const c2_bdesc *__restrict__ desc, sp_kparams32 kp, int shift_fine, int n_fine) {
    const c2_bdesc D = desc[blockIdx.y];
    c2_hist_fine_body(D.pk, D.pm, D.nm, D.n_units32, D.n_visit, D.sample_shift, kp, shift_fine, n_fine, D.ghist);
}
__global__ void __launch_bounds__(1024)
c2_offsets(const unsigned long long *__restrict__ ghist, int n_fine, int F1, int F2, unsigned long long mult8,
           unsigned long long mult8_1, unsigned long long slack, unsigned long long slack1, unsigned long long pad1,
           int split, unsigned long long sslack, unsigned long long *__restrict__ off_fine, unsigned long long *__restrict__ off1) {
    c2_offsets_body(ghist, n_fine, F1, F2, mult8, mult8_1, slack, slack1, pad1, split, sslack, off_fine, off1);
}
__global__ void __launch_bounds__(1024)
c2_offsets_b(const c2_bdesc *__restrict__ desc, int n_fine, int F1, int F2) {
    const c2_bdesc D = desc[blockIdx.y];
    c2_offsets_body(D.ghist, n_fine, F1, F2, D.mult8, D.mult8_1, D.slack, D.slack1, D.pad1, D.split, D.sslack, D.off_fine, D.off1);
}
__global__ void __launch_bounds__(C2_MAXV)
c2_tiles(const unsigned long long *__restrict__ cursor1, int V1, unsigned long long *__restrict__ off1,
         unsigned long long *__restrict__ tile_start, unsigned long long *__restrict__ flag) {
    c2_tiles_body(cursor1, V1, off1, tile_start, flag);
}
__global__ void __launch_bounds__(C2_MAXV)
c2_tiles_b(const c2_bdesc *__restrict__ desc, int V1) {
    const c2_bdesc D = desc[blockIdx.y];
    c2_tiles_body(D.cur1, V1, D.off1, D.tile_start, D.d_len4 + 3);
}
__global__ void __launch_bounds__(C2_P1_THREADS)
c2_part1(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm,
         int64_t n_units, sp_kparams32 kp, int shift1, int F1, int split, const unsigned long long *__restrict__ off1,
         unsigned long long *__restrict__ cursor1, int64_t n_tiles, uint16_t *__restrict__ lo1, uint8_t *__restrict__ hi1) {
    c2_part1_body(pk, pm, nm, n_units, kp, shift1, F1, split, off1, cursor1, n_tiles, lo1, hi1);
}
__global__ void __launch_bounds__(C2_P1_THREADS)
c2_part1_b(const c2_bdesc *__restrict__ desc, sp_kparams32 kp, int shift1, int F1) {
    const c2_bdesc D = desc[blockIdx.y];
    c2_part1_body(D.pk, D.pm, D.nm, D.n_units32, kp, shift1, F1, D.split, D.off1, D.cur1, D.n_tiles, D.lo1, D.hi1);
}
#ifndef C2_P2_MINW
#define C2_P2_MINW 1      // waves per SIMD the register allocation of c2_part2 must leave room for
#endif
__global__ void __launch_bounds__(C2_P2_THREADS, C2_P2_MINW)
c2_part2(const uint16_t *__restrict__ lo1, const uint8_t *__restrict__ hi1, const unsigned long long *__restrict__ off1,
         const unsigned long long *__restrict__ tile_start, int F1, int F2, int shift2, const unsigned long long *__restrict__ off_fine,
         unsigned long long *__restrict__ cursor2, uint16_t *__restrict__ buf2) {
    c2_part2_body(lo1, hi1, off1, tile_start, F1, F2, shift2, off_fine, cursor2, buf2);
}
__global__ void __launch_bounds__(C2_P2_THREADS)
c2_part2_b(const c2_bdesc *__restrict__ desc, int V1, int F2, int shift2) {
    const c2_bdesc D = desc[blockIdx.y];
    c2_part2_body(D.lo1, D.hi1, D.off1, D.tile_start, V1, F2, shift2, D.off_fine, D.cur2, D.buf2);
}
__global__ void __launch_bounds__(256)
c2_spans(const unsigned long long *__restrict__ off_fine, const unsigned long long *__restrict__ cursor2, int64_t n_fine,
         ulonglong2 *__restrict__ span, unsigned long long *__restrict__ flag, uint32_t list_div) {
    c2_spans_body(off_fine, cursor2, n_fine, span, flag, list_div);
}
__global__ void __launch_bounds__(256)
c2_spans_b(const c2_bdesc *__restrict__ desc, int64_t n_fine, uint32_t list_div) {
    const c2_bdesc D = desc[blockIdx.y];
    c2_spans_body(D.off_fine, D.cur2, n_fine, D.span, D.d_len4 + 3, list_div);
}
// BIG: chromosomes of 2^26 bases or more (peanut-like 128 Mb: ~8 K keys per fine bucket) hold FOUR quads per thread and bucket
// in registers, two buckets per group -- 16 K keys per bucket never leave the registers, where the default geometry (two quads,
// groups of four: made for 20-Mb chromosomes with ~1.2 K keys per bucket) sent every bucket above 8 K keys through the
// crowded path (re-read from memory, all 2^15 counters walked): peanut-like pass 23.23 -> 22.82 ms (round 6)
template <bool BIG>
__global__ void __launch_bounds__(C2_COUNT_THREADS)
c2_count_list(const uint16_t *__restrict__ buf2, const ulonglong2 *__restrict__ span, int64_t n_fine, uint32_t lower,
              unsigned long long *__restrict__ out3, uint2 *__restrict__ stage, unsigned long long stage_cap,
              uint32_t *__restrict__ seg_base, uint32_t *__restrict__ seg_cnt) {
    c2_count_list_body<(BIG ? 2 : C2L_DEPTH), (BIG ? 4 : C2L_PF)>(buf2, span, n_fine, lower, out3, stage, stage_cap, seg_base, seg_cnt);
}
// BIG: chromosomes of 2^26 bases or more (~8 K keys per fine bucket): four quads per thread in registers, groups of two buckets
template <bool BIG>
__global__ void __launch_bounds__(C2L16_THREADS)
c2_count_list16(const uint16_t *__restrict__ buf2, const ulonglong2 *__restrict__ span, int64_t n_fine, uint32_t lower,
                unsigned long long *__restrict__ out3, uint2 *__restrict__ stage, unsigned long long stage_cap,
                uint32_t *__restrict__ seg_base, uint32_t *__restrict__ seg_cnt) {
    c2_count_list16_body<(BIG ? 2 : C2L16_DEPTH), (BIG ? 4 : C2L16_PF)>(buf2, span, n_fine, lower, out3, stage, stage_cap, seg_base, seg_cnt);
}
__global__ void __launch_bounds__(C2L16_THREADS)
c2_count_list16_b(const c2_bdesc *__restrict__ desc, int64_t n_fine, uint32_t lower) {
    const c2_bdesc D = desc[blockIdx.y];
    c2_count_list16_body<C2L16_DEPTH, C2L16_PF>(D.buf2, D.span, n_fine, lower, D.d_len4, D.stage, D.stage_cap, D.seg_base, D.seg_cnt);
}
__global__ void __launch_bounds__(C2_COUNT_THREADS)
c2_count_list_b(const c2_bdesc *__restrict__ desc, int64_t n_fine, uint32_t lower) {
    const c2_bdesc D = desc[blockIdx.y];
    c2_count_list_body<C2L_DEPTH, C2L_PF>(D.buf2, D.span, n_fine, lower, D.d_len4, D.stage, D.stage_cap, D.seg_base, D.seg_cnt);
}

int sp_ovf_finalize(sp_ctx *ctx, sp_chrom &c, const uint2 *tmp, const uint32_t *seg_base, const uint32_t *seg_cnt,
                    uint32_t *seg_off, int64_t n_buckets, unsigned long long *d_total);   // sp_count.hip
int sp_ovf_finalize_split(sp_ctx *ctx, unsigned long long *keys, uint32_t *cnts, const uint2 *tmp, const uint32_t *seg_base,
                          const uint32_t *seg_cnt, uint32_t *seg_off, int64_t n_buckets, unsigned long long *d_total);

bool sp_engine2_supported(int64_t nslots) {
    c2_plan p;
    return c2_make_plan(nslots, p);
}

// exact = false: bucket regions from the 1-in-16 sample (the caller re-runs the chromosome with exact = true when
// d_len4[3] comes back non-zero); exact = true: regions = the exact histogram (a full scan), nothing can overrun.
// list != nullptr (engine 3): no byte table; the chromosome's (slot, count >= lower) pairs go to list->d_keys /
// d_cnts (capacity list->cap pairs, ascending slot order); their number comes back in d_len4[2].
int sp_count_engine2(sp_ctx *ctx, sp_chrom &c, const sp_kparams &kp, int lower, unsigned long long *d_len4, bool exact,
                     sp_sparse_chrom *list) {
    c2_plan P;
    if (!c2_make_plan(ctx->nslots, P))
        return sp_fail(ctx, SP_EUNSUP, "count engine 2 needs a dense table of 2^17..2^31 slots (k=%d)", kp.k);
    const int64_t n_units = (c.len + SP_UNIT - 1) / SP_UNIT;
    if (n_units == 0) {
        if (!list) SP_HIP(ctx, hipMemsetAsync(c.d_tab, 0, (size_t)ctx->nslots, ctx->stream));
        c.n_ovf = 0;
        c.ovf_idx_n = 0;
        return SP_OK;
    }
    const size_t nf = (size_t)P.n_fine;
    const int64_t n_units32 = (c.len + C2_P1_UNIT - 1) / C2_P1_UNIT;
    const int64_t n_tiles = (n_units32 + C2_P1_THREADS - 1) / C2_P1_THREADS;   // 16384 starts each
    // estimate mode: cap(f) = sample(f) * 16 * 9/8 + slack.  The slack absorbs what a 1-in-16 stripe sample cannot
    // see: a local array of one repeat shorter than 16 stripes (32 K starts) that falls between sampled stripes.
    const char *env_mult = getenv("SP_C2_MULT8"), *env_slack = getenv("SP_C2_SLACK");   // test hooks (force overruns)
    const int sample_shift = exact ? 0 : C2_SAMPLE_SHIFT;
    unsigned long long mult8 = exact ? 8ULL : (unsigned long long)(env_mult ? atoll(env_mult) : (9 << C2_SAMPLE_SHIFT));
    unsigned long long slack = 0;
    if (!exact) {
        int64_t sl = c.len / 64;
        sl = sl < 4096 ? 4096 : (sl > 32768 ? 32768 : sl);
        slack = (unsigned long long)(env_slack ? atoll(env_slack) : sl);
    }
    const unsigned long long slack1 = env_slack ? slack : 4ULL * slack + 65536ULL;
    const unsigned long long mult8_1 = (exact || env_mult) ? mult8 : (unsigned long long)(33 << (C2_SAMPLE_SHIFT - 2));   // x 1 1/32
    const int64_t n_stripes = (n_units32 + C2_STRIPE - 1) / C2_STRIPE;
    const int64_t n_visit = exact ? n_units32 : ((n_stripes + (1 << C2_SAMPLE_SHIFT) - 1) >> C2_SAMPLE_SHIFT) * C2_STRIPE;
    // keys the regions can hold at most (rigorous bounds: the sample sees at most n_visit * 32 k-mers).  Pad records:
    // three per (part1 tile, level-1 bucket) and per (part2 tile, fine bucket) at most -- c2_offsets adds the same terms.
    const unsigned long long pad1 = 3ULL * (unsigned long long)n_tiles + 3ULL * C2_SPLIT;
    // estimate mode: C2_SPLIT sub-regions per level-1 bucket, a quarter of the bucket's slack each; exact mode: one
    const int split = exact ? 1 : C2_SPLIT;
    const unsigned long long sslack = exact ? 0ULL : slack1 / 4;
    const int V1 = P.F1 * C2_SPLIT;
    const size_t cap_keys1 = (exact ? (size_t)c.len : (size_t)(((unsigned long long)n_visit * C2_P1_UNIT * mult8_1 + 7) / 8) +
                                                          (size_t)P.F1 * (size_t)slack1) + (size_t)P.F1 * (size_t)(pad1 + 4) +
                             (size_t)V1 * (size_t)(sslack + 8);
    const size_t tiles2_max = cap_keys1 / C2_TILE_KEYS + 2 * (size_t)V1;
    const size_t cap_keys2 = (exact ? (size_t)c.len + 4 * nf
                                    : (size_t)(((unsigned long long)n_visit * C2_P1_UNIT * mult8 + 7) / 8) + nf * (size_t)(slack + 4)) +
                             3 * (size_t)P.F2 * tiles2_max;
    const size_t cap_keys = cap_keys1 > cap_keys2 ? cap_keys1 : cap_keys2;
    // workspace: ghist | off_fine | off1 | tile_start | cursor1 | cursor2 | level-1 planes (3 B / key) | buf2 (u16 / key)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_ghist = 0;
    size_t o_offf = o_ghist + al(nf * 8);
    size_t o_off1 = o_offf + al((nf + 1) * 8);
    size_t o_tile = o_off1 + al((size_t)(V1 + 1) * 16);         // level-1 sub-region starts, then ends
    size_t o_cur1 = o_tile + al((size_t)(V1 + 1) * 8);
    size_t o_cur2 = o_cur1 + al((size_t)C2_MAXV * 8 * C2_CSTRIDE);   // (any layout of C2_CUR1: sub-region m of bucket b)
    size_t o_tcnt = o_cur2 + al(nf * 8 * C2_CSTRIDE2);           // end of the zeroed head of the workspace
    size_t o_span = o_tcnt;
    o_tcnt = o_span + al(nf * 16);              // (the spans are written before they are read: not zeroed)
    const size_t zero_bytes = o_span;
    const size_t lo1_bytes = al(cap_keys1 * 2 + 64 + (size_t)V1 * 8);
    size_t o_buf1 = o_tcnt;
    size_t o_buf2 = o_buf1 + lo1_bytes + al(cap_keys1 + 64 + (size_t)V1 * 4);
    size_t o_segb = o_buf2 + al(cap_keys2 * 2 + 64);               // overflow segments: base, count, offsets per fine bucket
    size_t o_segc = o_segb + al(nf * 4);
    size_t o_sego = o_segc + al(nf * 4);
    size_t o_bigl = o_sego + al((nf + 1) * 4);                    // (unused since round 6: c2_count16 keeps every bucket)
    size_t total = o_bigl + al(nf * 4);
    void *&d_ws2 = ctx->lane ? ctx->lane->d_ws2 : ctx->d_ws2;           // this lane's workspace (sp_common.h)
    int64_t &ws2_bytes = ctx->lane ? ctx->lane->ws2_bytes : ctx->ws2_bytes;
    sp_buf &b_ovfw = ctx->lane ? ctx->lane->b_ovfw : ctx->b_ovfw;
    if ((int64_t)total > ws2_bytes) {
        if (d_ws2) {
            SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            SP_HIP(ctx, hipFree(d_ws2));
            d_ws2 = nullptr;
            ws2_bytes = 0;
        }
        SP_HIP(ctx, hipMalloc(&d_ws2, total));
        ws2_bytes = (int64_t)total;
    }
    char *ws = (char *)d_ws2;
    unsigned long long *ghist = (unsigned long long *)(ws + o_ghist);
    unsigned long long *off_fine = (unsigned long long *)(ws + o_offf);
    unsigned long long *off1 = (unsigned long long *)(ws + o_off1);
    unsigned long long *tile_start = (unsigned long long *)(ws + o_tile);
    unsigned long long *cur1 = (unsigned long long *)(ws + o_cur1), *cur2 = (unsigned long long *)(ws + o_cur2);
    uint16_t *buf2 = (uint16_t *)(ws + o_buf2);
    uint32_t *seg_base = (uint32_t *)(ws + o_segb), *seg_cnt = (uint32_t *)(ws + o_segc), *seg_off = (uint32_t *)(ws + o_sego);
    uint16_t *lo1 = (uint16_t *)(ws + o_buf1);                         // level-1 records: 3 bytes per key in two planes
    uint8_t *hi1 = (uint8_t *)(ws + o_buf1) + lo1_bytes;
    // the unordered overflow pairs live in the level-1 planes (dead once part2 has run); an engine-3 (list) run stages every
    // kept slot and gets a buffer of its own
    // (a bucket's segment starts at floor(its first key / L), L = 255 or lower_count: cap_keys / L pairs at most)
    uint2 *ovf_tmp = (uint2 *)(ws + o_buf1);
    unsigned long long ovf_cap = (unsigned long long)(lo1_bytes / 8);
    if (list && c.len >= (1LL << 32) - 16)
        return sp_fail(ctx, SP_EUNSUP, "count engine 3: a chromosome of %lld bases (32-bit positions inside a bucket)", (long long)c.len);
    if (cap_keys / (size_t)(list ? lower : 255) + C2_FINE >= ((size_t)1 << 32))
        return sp_fail(ctx, SP_EUNSUP, "count engine 2: a chromosome of %lld bases needs 64-bit segment offsets", (long long)c.len);
    if (list) {
        const unsigned long long need = (unsigned long long)(cap_keys / (size_t)lower) + C2_FINE + 16;
        int rcl = sp_buf_ensure(ctx, b_ovfw, (int64_t)need * 8);
        if (rcl) return rcl;
        ovf_tmp = (uint2 *)b_ovfw.p;
        ovf_cap = need;
    }
    // zero ghist .. cursor2 in one memset (they are contiguous)
    SP_HIP(ctx, hipMemsetAsync(ws, 0, zero_bytes, ctx->stream));

    const sp_kparams32 kp32 = sp_make_kparams32(kp.k);
    int grid_scan = (int)(n_tiles < (int64_t)ctx->n_cu * 8 ? n_tiles : (int64_t)ctx->n_cu * 8);
    size_t sh_hist = nf * 4;
    const int64_t hist_tiles = (n_visit + C2_P1_THREADS - 1) / C2_P1_THREADS;
    // (every block flushes its LDS histogram with one global atomic per non-empty bin: the sample runs on fewer blocks)
    const int64_t hist_blocks = exact ? (int64_t)ctx->n_cu * 2 : (int64_t)ctx->n_cu / 2;
    int grid_hist = (int)(hist_tiles < hist_blocks ? hist_tiles : hist_blocks);
    if (sh_hist > 48 * 1024)
        SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_hist_fine, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_hist));
    SP_LAUNCH(ctx, exact ? "c2_hist_fine" : "c2_hist_sample", c2_hist_fine, dim3(grid_hist), dim3(C2_P1_THREADS), sh_hist,
              c.d_pk, c.d_pm, c.d_nm, n_units32, n_visit, sample_shift, kp32, C2_B3, (int)nf, ghist);
    SP_LAUNCH(ctx, "c2_offsets", c2_offsets, dim3(1), dim3(1024), 0, ghist, (int)nf, P.F1, P.F2, mult8, mult8_1, slack, slack1, pad1,
              split, sslack, off_fine, off1);
    SP_LAUNCH(ctx, "c2_part1", c2_part1, dim3(grid_scan), dim3(C2_P1_THREADS), 0, c.d_pk, c.d_pm, c.d_nm, n_units32,
              kp32, P.T - P.B1, P.F1, split, off1, cur1, n_tiles, lo1, hi1);
    SP_LAUNCH(ctx, "c2_tiles", c2_tiles, dim3(1), dim3(C2_MAXV), 0, (const unsigned long long *)cur1, V1, off1, tile_start,
              d_len4 + 3);
    // part2 grid: enough blocks to cover the tiles (tile count lives on the device; over-provision)
    int64_t max_tiles2 = (int64_t)tiles2_max;
    int grid2 = (int)(max_tiles2 < (int64_t)ctx->n_cu * 8 ? max_tiles2 : (int64_t)ctx->n_cu * 8);
    SP_LAUNCH(ctx, "c2_part2", c2_part2, dim3(grid2), dim3(C2_P2_THREADS), 0, (const uint16_t *)lo1, (const uint8_t *)hi1, off1,
              tile_start, V1,
              P.F2, C2_B3, off_fine, cur2, buf2);
    int gridc = (int)((int64_t)nf < (int64_t)ctx->n_cu ? (int64_t)nf : (int64_t)ctx->n_cu);
    ulonglong2 *span = (ulonglong2 *)(ws + o_span);
    SP_LAUNCH(ctx, "c2_spans", c2_spans, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, (const unsigned long long *)off_fine,
              (const unsigned long long *)cur2, (int64_t)nf, span, d_len4 + 3, (uint32_t)(list ? lower : 0));
    if (list) {
        if ((int64_t)gridc * C2L_MAXB < (int64_t)nf) gridc = (int)(((int64_t)nf + C2L_MAXB - 1) / C2L_MAXB);
        const char *env_l16 = getenv("SP_C2_LIST16");      // "0": the 32-bit list counter everywhere (cross-check)
        if (!exact && !(env_l16 && env_l16[0] == '0')) {
            // (estimate mode only: a bucket of 65536 keys or more raises the overrun flag and the exact recount takes the kernels below)
            int g16 = (int)((int64_t)nf < (int64_t)ctx->n_cu * 2 ? (int64_t)nf : (int64_t)ctx->n_cu * 2);
            if ((int64_t)g16 * C2L_MAXB < (int64_t)nf) g16 = (int)(((int64_t)nf + C2L_MAXB - 1) / C2L_MAXB);
            if (c.len >= C2L16_MAXLEN) {
                SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_count_list16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_FINE * 2));
                SP_LAUNCH(ctx, "c2_count_list", c2_count_list16<true>, dim3(g16), dim3(C2L16_THREADS), C2_FINE * 2, buf2,
                          (const ulonglong2 *)span, (int64_t)nf, (uint32_t)lower, d_len4, ovf_tmp, ovf_cap, seg_base, seg_cnt);
            } else {
                SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_count_list16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_FINE * 2));
                SP_LAUNCH(ctx, "c2_count_list", c2_count_list16<false>, dim3(g16), dim3(C2L16_THREADS), C2_FINE * 2, buf2,
                          (const ulonglong2 *)span, (int64_t)nf, (uint32_t)lower, d_len4, ovf_tmp, ovf_cap, seg_base, seg_cnt);
            }
        } else if (c.len >= (1LL << 26)) {
            SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_count_list<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_FINE * 4));
            SP_LAUNCH(ctx, "c2_count_list", c2_count_list<true>, dim3(gridc), dim3(C2_COUNT_THREADS), C2_FINE * 4, buf2,
                      (const ulonglong2 *)span, (int64_t)nf, (uint32_t)lower, d_len4, ovf_tmp, ovf_cap, seg_base, seg_cnt);
        } else {
            SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_count_list<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_FINE * 4));
            SP_LAUNCH(ctx, "c2_count_list", c2_count_list<false>, dim3(gridc), dim3(C2_COUNT_THREADS), C2_FINE * 4, buf2,
                      (const ulonglong2 *)span, (int64_t)nf, (uint32_t)lower, d_len4, ovf_tmp, ovf_cap, seg_base, seg_cnt);
        }
        return sp_ovf_finalize_split(ctx, (unsigned long long *)list->d_keys, list->d_cnts, ovf_tmp, seg_base, seg_cnt, seg_off,
                                     (int64_t)nf, d_len4 + 2);
    }
    SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_count, hipFuncAttributeMaxDynamicSharedMemorySize, C2_FINE * 4));
    const char *env16 = getenv("SP_C2_COUNT16");      // "0": every bucket through the 32-bit counters (cross-check)
    if (env16 && env16[0] == '0') {
        SP_LAUNCH(ctx, "c2_count", c2_count, dim3(gridc), dim3(C2_COUNT_THREADS), C2_FINE * 4, buf2, (const ulonglong2 *)span,
                  (int64_t)nf, (const uint32_t *)nullptr, (const unsigned long long *)nullptr, (uint32_t)lower, c.d_tab, d_len4,
                  ovf_tmp, ovf_cap, seg_base, seg_cnt);
    } else {
        int grid16 = (int)((int64_t)nf < (int64_t)ctx->n_cu * 2 ? (int64_t)nf : (int64_t)ctx->n_cu * 2);
        SP_LAUNCH(ctx, "c2_count16", c2_count16, dim3(grid16), dim3(C2_C16_THREADS), 0, buf2, (const ulonglong2 *)span, (int64_t)nf,
                  (uint32_t)lower, c.d_tab, d_len4, ovf_tmp, ovf_cap, seg_base, seg_cnt);
    }
    return sp_ovf_finalize(ctx, c, ovf_tmp, seg_base, seg_cnt, seg_off, (int64_t)nf, d_len4 + 2);
}

// ---------------------------------------------------------------- engine 3, batched: one launch per kernel type (sp_c2batch.h)
// Same chain, same layout arithmetic as sp_count_engine2 (estimate mode only: a chromosome whose flag comes back up is
// counted again, alone, from the exact histogram by the caller).  Workspace: the zeroed heads (histogram .. cursors) of all
// chromosomes back to back -- one memset -- then their bodies.
int sp_ovf_finalize_split_batch(sp_ctx *ctx, const c2_bdesc *d_desc, int n_chrom, int64_t n_buckets);   // sp_count.hip
int sp_count_engine3_batch(sp_ctx *ctx, const int *chrom_idx, int n, const sp_kparams &kp, int lower, unsigned long long *d_len /* 4 per chromosome */) {
    c2_plan P;
    if (!c2_make_plan(ctx->nslots, P))
        return sp_fail(ctx, SP_EUNSUP, "count engine 3 needs a dense slot space of 2^17..2^31 slots (k=%d)", kp.k);
    const size_t nf = (size_t)P.n_fine;
    const int V1 = P.F1 * C2_SPLIT;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // head (zeroed): ghist | off_fine | off1 | tile_start | cursor1 | cursor2
    const size_t h_ghist = 0, h_offf = h_ghist + al(nf * 8), h_off1 = h_offf + al((nf + 1) * 8),
                 h_tile = h_off1 + al((size_t)(V1 + 1) * 16), h_cur1 = h_tile + al((size_t)(V1 + 1) * 8),
                 h_cur2 = h_cur1 + al((size_t)C2_MAXV * 8 * C2_CSTRIDE), head_bytes = h_cur2 + al(nf * 8 * C2_CSTRIDE2);
    const char *env_mult = getenv("SP_C2_MULT8"), *env_slack = getenv("SP_C2_SLACK");   // test hooks (force overruns)
    std::vector<c2_bdesc> hd((size_t)n);
    std::vector<size_t> body_off((size_t)n), lo1_bytes_v((size_t)n), k1((size_t)n), k2((size_t)n), stage_n((size_t)n);
    size_t body_total = 0;
    int64_t max_tiles = 0, max_hist_tiles = 0, max_tiles2 = 0;
    for (int i = 0; i < n; i++) {
        const sp_chrom &c = ctx->chroms[(size_t)chrom_idx[i]];
        c2_bdesc &D = hd[(size_t)i];
        D.pk = c.d_pk;
        D.pm = c.d_pm;
        D.nm = c.d_nm;
        D.n_units32 = (c.len + C2_P1_UNIT - 1) / C2_P1_UNIT;
        D.n_tiles = (D.n_units32 + C2_P1_THREADS - 1) / C2_P1_THREADS;
        D.sample_shift = C2_SAMPLE_SHIFT;
        D.mult8 = (unsigned long long)(env_mult ? atoll(env_mult) : (9 << C2_SAMPLE_SHIFT));
        int64_t sl = c.len / 64;
        sl = sl < 4096 ? 4096 : (sl > 32768 ? 32768 : sl);
        D.slack = (unsigned long long)(env_slack ? atoll(env_slack) : sl);
        D.slack1 = env_slack ? D.slack : 4ULL * D.slack + 65536ULL;
        D.mult8_1 = env_mult ? D.mult8 : (unsigned long long)(33 << (C2_SAMPLE_SHIFT - 2));
        const int64_t n_stripes = (D.n_units32 + C2_STRIPE - 1) / C2_STRIPE;
        D.n_visit = ((n_stripes + (1 << C2_SAMPLE_SHIFT) - 1) >> C2_SAMPLE_SHIFT) * C2_STRIPE;
        D.pad1 = 3ULL * (unsigned long long)D.n_tiles + 3ULL * C2_SPLIT;
        D.split = C2_SPLIT;
        D.sslack = D.slack1 / 4;
        const size_t cap_keys1 = (size_t)(((unsigned long long)D.n_visit * C2_P1_UNIT * D.mult8_1 + 7) / 8) + (size_t)P.F1 * (size_t)D.slack1 +
                                 (size_t)P.F1 * (size_t)(D.pad1 + 4) + (size_t)V1 * (size_t)(D.sslack + 8);
        const size_t tiles2_max = cap_keys1 / C2_TILE_KEYS + 2 * (size_t)V1;
        const size_t cap_keys2 = (size_t)(((unsigned long long)D.n_visit * C2_P1_UNIT * D.mult8 + 7) / 8) + nf * (size_t)(D.slack + 4) +
                                 3 * (size_t)P.F2 * tiles2_max;
        const size_t cap_keys = cap_keys1 > cap_keys2 ? cap_keys1 : cap_keys2;
        if (c.len >= (1LL << 32) - 16 || cap_keys / (size_t)lower + C2_FINE >= ((size_t)1 << 32))
            return sp_fail(ctx, SP_EUNSUP, "count engine 3: a chromosome of %lld bases", (long long)c.len);
        k1[(size_t)i] = cap_keys1;
        k2[(size_t)i] = cap_keys2;
        stage_n[(size_t)i] = cap_keys / (size_t)lower + C2_FINE + 16;
        lo1_bytes_v[(size_t)i] = al(cap_keys1 * 2 + 64 + (size_t)V1 * 8);
        // body: span | level-1 planes | buf2 | seg_base | seg_cnt | seg_off | stage
        body_off[(size_t)i] = body_total;
        body_total += al(nf * 16) + lo1_bytes_v[(size_t)i] + al(cap_keys1 + 64 + (size_t)V1 * 4) + al(cap_keys2 * 2 + 64) + 2 * al(nf * 4) +
                      al((nf + 1) * 4) + al(stage_n[(size_t)i] * 8);
        if (D.n_tiles > max_tiles) max_tiles = D.n_tiles;
        const int64_t ht = (D.n_visit + C2_P1_THREADS - 1) / C2_P1_THREADS;
        if (ht > max_hist_tiles) max_hist_tiles = ht;
        if ((int64_t)tiles2_max > max_tiles2) max_tiles2 = (int64_t)tiles2_max;
    }
    const size_t total = al((size_t)n * sizeof(c2_bdesc)) + (size_t)n * head_bytes + body_total;
    void *&d_ws2 = ctx->lane ? ctx->lane->d_ws2 : ctx->d_ws2;           // this lane's workspace (sp_common.h)
    int64_t &ws2_bytes = ctx->lane ? ctx->lane->ws2_bytes : ctx->ws2_bytes;
    if ((int64_t)total > ws2_bytes) {
        if (d_ws2) {
            SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            SP_HIP(ctx, hipFree(d_ws2));
            d_ws2 = nullptr;
            ws2_bytes = 0;
        }
        SP_HIP(ctx, hipMalloc(&d_ws2, total));
        ws2_bytes = (int64_t)total;
    }
    char *ws = (char *)d_ws2;
    c2_bdesc *d_desc = (c2_bdesc *)ws;
    char *heads = ws + al((size_t)n * sizeof(c2_bdesc)), *bodies = heads + (size_t)n * head_bytes;
    for (int i = 0; i < n; i++) {
        c2_bdesc &D = hd[(size_t)i];
        char *h = heads + (size_t)i * head_bytes, *b = bodies + body_off[(size_t)i];
        D.ghist = (unsigned long long *)(h + h_ghist);
        D.off_fine = (unsigned long long *)(h + h_offf);
        D.off1 = (unsigned long long *)(h + h_off1);
        D.tile_start = (unsigned long long *)(h + h_tile);
        D.cur1 = (unsigned long long *)(h + h_cur1);
        D.cur2 = (unsigned long long *)(h + h_cur2);
        D.span = (ulonglong2 *)b;
        b += al(nf * 16);
        D.lo1 = (uint16_t *)b;
        D.hi1 = (uint8_t *)b + lo1_bytes_v[(size_t)i];
        b += lo1_bytes_v[(size_t)i] + al(k1[(size_t)i] + 64 + (size_t)V1 * 4);
        D.buf2 = (uint16_t *)b;
        b += al(k2[(size_t)i] * 2 + 64);
        D.seg_base = (uint32_t *)b;
        b += al(nf * 4);
        D.seg_cnt = (uint32_t *)b;
        b += al(nf * 4);
        D.seg_off = (uint32_t *)b;
        b += al((nf + 1) * 4);
        D.stage = (uint2 *)b;
        D.stage_cap = (unsigned long long)stage_n[(size_t)i];
        D.d_len4 = d_len + 4 * (size_t)chrom_idx[i];
        D.out_keys = (unsigned long long *)ctx->sparse[(size_t)chrom_idx[i]].d_keys;
        D.out_cnts = ctx->sparse[(size_t)chrom_idx[i]].d_cnts;
    }
    // the table travels from a page-locked buffer of this lane's (advisor r05: a pageable source is only safe because the
    // runtime stages it before hipMemcpyAsync returns -- and that staging made the host wait on the lane's stream).  The
    // buffer is free: a lane issues one batched count per sp_count call, and sp_count ends with a host synchronisation.
    void *&h_desc = ctx->lane ? ctx->lane->h_desc : ctx->h_desc;
    int64_t &h_desc_cap = ctx->lane ? ctx->lane->h_desc_cap : ctx->h_desc_cap;
    if ((int64_t)((size_t)n * sizeof(c2_bdesc)) > h_desc_cap) {
        if (h_desc) {
            SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            SP_HIP(ctx, hipHostFree(h_desc));
            h_desc = nullptr;
            h_desc_cap = 0;
        }
        const size_t cap = (size_t)(n < 64 ? 64 : n) * sizeof(c2_bdesc);
        SP_HIP(ctx, hipHostMalloc(&h_desc, cap, hipHostMallocDefault));
        h_desc_cap = (int64_t)cap;
    }
    memcpy(h_desc, hd.data(), (size_t)n * sizeof(c2_bdesc));
    SP_HIP(ctx, hipMemcpyAsync(d_desc, h_desc, (size_t)n * sizeof(c2_bdesc), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemsetAsync(heads, 0, (size_t)n * head_bytes, ctx->stream));
    const sp_kparams32 kp32 = sp_make_kparams32(kp.k);
    const unsigned un = (unsigned)n;
    // grids: what the largest chromosome needs, capped so that all chromosomes together fill the chip a few times over
    auto cap_grid = [&](int64_t need, int64_t per_cu) {
        int64_t g = ((int64_t)ctx->n_cu * per_cu + n - 1) / n;
        if (g > need) g = need;
        return (unsigned)(g < 1 ? 1 : g);
    };
    const size_t sh_hist = nf * 4;
    if (sh_hist > 48 * 1024)
        SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_hist_fine_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_hist));
    SP_LAUNCH(ctx, "c2_hist_sample", c2_hist_fine_b, dim3(cap_grid(max_hist_tiles, 2), un), dim3(C2_P1_THREADS), sh_hist, (const c2_bdesc *)d_desc,
              kp32, C2_B3, (int)nf);
    SP_LAUNCH(ctx, "c2_offsets", c2_offsets_b, dim3(1, un), dim3(1024), 0, (const c2_bdesc *)d_desc, (int)nf, P.F1, P.F2);
    SP_LAUNCH(ctx, "c2_part1", c2_part1_b, dim3(cap_grid(max_tiles, 8), un), dim3(C2_P1_THREADS), 0, (const c2_bdesc *)d_desc, kp32, P.T - P.B1,
              P.F1);
    SP_LAUNCH(ctx, "c2_tiles", c2_tiles_b, dim3(1, un), dim3(C2_MAXV), 0, (const c2_bdesc *)d_desc, V1);
    SP_LAUNCH(ctx, "c2_part2", c2_part2_b, dim3(cap_grid(max_tiles2, 8), un), dim3(C2_P2_THREADS), 0, (const c2_bdesc *)d_desc, V1, P.F2, C2_B3);
    SP_LAUNCH(ctx, "c2_spans", c2_spans_b, dim3((unsigned)((nf + 255) / 256), un), dim3(256), 0, (const c2_bdesc *)d_desc, (int64_t)nf,
              (uint32_t)lower);
    const char *env_l16 = getenv("SP_C2_LIST16");      // "0": the 32-bit list counter (cross-check)
    if (!(env_l16 && env_l16[0] == '0')) {
        // two 512-thread workgroups per CU on 16-bit counters (a bucket of >= 65536 keys raises the chromosome's overrun flag: the
        // caller counts it again, alone, with exact sizes and the 32-bit kernel)
        unsigned g16 = cap_grid((int64_t)nf, 2);
        if ((int64_t)g16 * C2L_MAXB < (int64_t)nf) g16 = (unsigned)(((int64_t)nf + C2L_MAXB - 1) / C2L_MAXB);
        SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_count_list16_b, hipFuncAttributeMaxDynamicSharedMemorySize, C2_FINE * 2));
        SP_LAUNCH(ctx, "c2_count_list", c2_count_list16_b, dim3(g16, un), dim3(C2L16_THREADS), C2_FINE * 2, (const c2_bdesc *)d_desc, (int64_t)nf,
                  (uint32_t)lower);
    } else {
        unsigned gridc = cap_grid((int64_t)nf, 1);
        if ((int64_t)gridc * C2L_MAXB < (int64_t)nf) gridc = (unsigned)(((int64_t)nf + C2L_MAXB - 1) / C2L_MAXB);
        SP_HIP(ctx, hipFuncSetAttribute((const void *)c2_count_list_b, hipFuncAttributeMaxDynamicSharedMemorySize, C2_FINE * 4));
        SP_LAUNCH(ctx, "c2_count_list", c2_count_list_b, dim3(gridc, un), dim3(C2_COUNT_THREADS), C2_FINE * 4, (const c2_bdesc *)d_desc, (int64_t)nf,
                  (uint32_t)lower);
    }
    return sp_ovf_finalize_split_batch(ctx, d_desc, n, (int64_t)nf);
}
