// sp_filter.hip -- K3: chromosome x k-mer matrix (outer join of the per-
// chromosome tables) + SubPhaser's differential-k-mer filter.
//
// Replaces JellyfishDumps.to_matrix (Jellyfish.py:439-460) and
// JellyfishDumps.filter / _filter_kmer (Jellyfish.py:462-512, 611-648).
//
// Pass A (k3_eval) streams all C byte tables once (C bytes per slot, 16-B loads)
// into an LDS tile, finds the slots that hold any count >= lower_count with a
// few SWAR instructions per 4 slots, and evaluates the filter -- fp64, the
// reference's operation order -- only for those (about one slot in ten), every
// lane busy.  It writes two slot bitmaps (row = differential k-mer, hist =
// fold-passing) plus per-block popcounts.
// Pass B (k3_emit) walks the bitmap (64 MiB at k=15) and gathers the M
// surviving rows, in ascending slot order, so the output is deterministic.
#include "sp_device.h"
#include "sp_filter.h"

#ifndef F_BLOCK
#define F_BLOCK 256
#endif
#ifndef F_GROUPS_PER_WAVE
#define F_GROUPS_PER_WAVE 64
#endif
#define F_WAVES (F_BLOCK / 64)
#define F_SLOTS_PER_BLOCK (F_WAVES * F_GROUPS_PER_WAVE * 64)  // 16384
#ifndef F_TILE_BYTES
#define F_TILE_BYTES (24 * 1024)   // LDS budget of the staged byte rows (measured: 48 K 13.0 ms, 24 K 9.3, 12 K 11.9)
#endif

#define F3_CHROM_MASK 0xfffff
#define F3_UNIT_END (1 << 20)
#define F3_SET_END (1 << 21)
#define F3_TOT (1 << 22)       // first row of its chromosome: counts towards tot
#define F3_BI1 (1 << 23)       // the set's baseline is the second largest frequency (else the smallest)

struct sp_filter_params {
    int C;
    int n_sets;
    int baseline;
    uint32_t lower;
    double min_fold, min_freq, max_freq, ratio;
    int64_t nslots;
    int64_t slot_base;   // absolute slot of local index 0 (slot-range views)
    int TS;              // slots per LDS tile (power of two, 256 .. F_SLOTS_PER_BLOCK)
    int n_multi;         // sets with more than one unit
    int need_active;     // fewest non-singleton sets a k-mer must occur in to reach include / _all >= ratio (0: no screen)
    int need_hist;       // smallest `include` with !(include / _all < ratio); n_multi + 1 if none
    int fast;            // every set uses baseline 1 or -1: the row walk below applies (else: generic decision)
    int R, R_sets;       // LDS rows: [0, R_sets) = the chromosomes of the non-singleton sets in config order (a
                         // chromosome listed twice is staged twice), [R_sets, R) = the remaining chromosomes
    const int32_t *rowdesc;   // per row: chromosome | F3_UNIT_END | F3_SET_END | F3_TOT | F3_BI1
    const float *rowinv;      // per row that ends a unit: 1 / (sum of the unit's lengths), fp32
    const int32_t *row_of_chrom;   // first row of every chromosome (generic / exact decisions)
    uint32_t *gq;             // global slow queue: entries of 1 + (R + 3) / 4 words (local slot, the column's bytes)
    unsigned long long *gq_n; // entries pushed (may exceed gq_cap: the excess was decided inline)
    unsigned long long gq_cap;
};

// bytes >= L (1 <= L <= 128) -> 0x80 in that byte, else 0
__device__ __forceinline__ uint32_t k3_ge_mask(uint32_t x, uint32_t addL /* (0x80 - L) * 0x01010101 */) {
    return (((x & 0x7f7f7f7fu) + addL) | x) & 0x80808080u;
}

// NCH > 0: the column has at most 32 rows: the row descriptors (set / unit boundaries) are bit masks in scalar
// registers, the reciprocal lengths one per lane.
// NCH = 0: any number of rows; every qualifying slot takes the generic decision.
#ifndef K3_STAGE
#define K3_STAGE 6      // 16-byte loads a thread keeps in flight while a tile is staged: 6 x F_BLOCK x 16 B = F_TILE_BYTES, one batch
                        // per tile (8: 3.96 ms, 6: 3.76 -- pieces past the tile's end are loaded again from a clamped index; a uniform
                        // branch around them instead: 4.56 ms, the loads no longer go out together)
#endif
#ifndef K3_SCAN_CH
#define K3_SCAN_CH 4    // rows whose words a thread of the scan reads from LDS before it looks at any
#endif
#ifndef K3_WALK_CH
#define K3_WALK_CH 8    // ... and bytes per chunk of the decision walk
#endif
template <bool SWAR, int NCH>
__global__ void __launch_bounds__(F_BLOCK)
k3_eval(const sp_tabref *__restrict__ tabs, sp_filter_params P,
        const int32_t *__restrict__ set_off, const int32_t *__restrict__ unit_off,
        const int32_t *__restrict__ unit_chrom, const double *__restrict__ unit_den,
        unsigned long long *__restrict__ bm_row, unsigned long long *__restrict__ bm_hist,
        unsigned long long *__restrict__ blk_row, unsigned long long *__restrict__ blk_hist,
        unsigned long long *__restrict__ n_union) {
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];   // [R][TS] bytes | queue u16[TS] | slow queue u16[TS] | bitmaps u32[2][TS/32] | lists
    __shared__ unsigned long long red[16];
    __shared__ uint32_t s_qn, s_q2n, s_q3n;
    __shared__ unsigned long long s_gq0;
    const int TS = P.TS, R = P.R;
    const int32_t *__restrict__ rowdesc = P.rowdesc;   // padded with zeros to a multiple of 8 (and at least 32)
    const float *__restrict__ rowinv = P.rowinv;
    uint16_t *queue = reinterpret_cast<uint16_t *>(tile + (size_t)R * TS), *queue2 = queue + TS;
    uint32_t *bmr = reinterpret_cast<uint32_t *>(queue2 + TS), *bmh = bmr + TS / 32;
    // overflow lists per ROW, kept in LDS (slow path only)
    const uint2 **l_ovf = reinterpret_cast<const uint2 **>(bmh + TS / 32);
    unsigned long long *l_novf = reinterpret_cast<unsigned long long *>(l_ovf + R);
    uint32_t *resolved = reinterpret_cast<uint32_t *>(l_novf + R);   // [waves][R] exact counts of the slot a wave is deciding
    for (int r = threadIdx.x; r < R; r += F_BLOCK) {
        const int c = rowdesc[r] & F3_CHROM_MASK;
        l_ovf[r] = tabs[c].ovf;
        l_novf[r] = (unsigned long long)tabs[c].n_ovf;
    }
    // row descriptors of the fast walk: loaded ONCE, uniform, packed into four bit masks (one bit per row, R <= 32) in scalar
    // registers; the reciprocal lengths sit one per LANE and are fetched with v_readlane where a unit ends.
    // Round 6: the scan and the walk were unrolled over all 8 * NCH rows.  Every one of the 4 x 24 bit tests is loop-invariant:
    // the compiler hoisted them out of the tile loop as lane masks, ran out of scalar registers, spilled them to VGPR lanes and
    // paid two v_readlane per test -- 563 VALU instructions per scan and thread, every LDS read waited for on its own.  Both are
    // loops over chunks of rows now (the row index is a loop variable: s_bitcmp + s_cbranch_scc where a flag is used).
    uint32_t mk_unit = 0, mk_set = 0, mk_bi1 = 0, mk_tot = 0;
    int inv_bits = 0;
    if (NCH > 0) {
        const int d = rowdesc[threadIdx.x & 31];
        const unsigned long long b_unit = __ballot((d & F3_UNIT_END) != 0), b_set = __ballot((d & F3_SET_END) != 0);
        const unsigned long long b_bi1 = __ballot((d & F3_BI1) != 0), b_tot = __ballot((d & F3_TOT) != 0);
        mk_unit = (uint32_t)b_unit; mk_set = (uint32_t)b_set; mk_bi1 = (uint32_t)b_bi1; mk_tot = (uint32_t)b_tot;
        inv_bits = __float_as_int(rowinv[threadIdx.x & 31]);
    }
    mk_unit = __builtin_amdgcn_readfirstlane(mk_unit);
    mk_set = __builtin_amdgcn_readfirstlane(mk_set);
    mk_bi1 = __builtin_amdgcn_readfirstlane(mk_bi1);
    mk_tot = __builtin_amdgcn_readfirstlane(mk_tot);
    const int lane = threadIdx.x & 63;
    unsigned long long nrow = 0, nhist = 0, nuni = 0;
    const int64_t blk_base = (int64_t)blockIdx.x * F_SLOTS_PER_BLOCK;
    const uint32_t addL = (0x80u - P.lower) * 0x01010101u;
    const float fold32 = (float)P.min_fold;
    sp_fsets F;
    F.n_sets = P.n_sets;
    F.n_multi = P.n_multi;
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
    for (int64_t base = blk_base; base < blk_base + F_SLOTS_PER_BLOCK && base < P.nslots; base += TS) {
        // ---- stage the tile: 16 slots per load, K3_STAGE loads in flight per thread
        const int n16 = TS / 16, total16 = R * n16;
        if (threadIdx.x == 0) s_qn = s_q2n = 0;
        for (int i = threadIdx.x; i < TS / 16; i += F_BLOCK) reinterpret_cast<uint32_t *>(bmr)[i] = 0;   // both bitmaps: 2*TS/32 words
        // Round 5: the "eight loads in flight" below were eight loads ONE AFTER THE OTHER -- every one sat behind two
        // dependent loads of its own (rowdesc[r], then tabs[c].tab) and a branch, and the compiler waits for everything
        // in flight (vmcnt(0)) before it uses a loaded pointer: three serial round trips per 16 bytes, in a kernel that
        // streams 11 GB per pass.  A tile that lies inside the table -- every tile but the last of a tiny table -- is now
        // loaded without a condition, from a clamped index: eight independent 16-byte loads per thread and one wait.
        // A wave's 64 consecutive 16-byte pieces lie in ONE row when a row is a multiple of 64 pieces (TS = 1024: exactly
        // one): the row, its table and the piece offset are wave-uniform -- scalar loads and a scalar base for the load.
        const bool whole = base + TS <= P.nslots && (n16 & 63) == 0;     // block-uniform
        if (whole)
        for (int i0 = threadIdx.x; i0 < total16; i0 += K3_STAGE * F_BLOCK) {
            uint32_t vx[K3_STAGE], vy[K3_STAGE], vz[K3_STAGE], vw[K3_STAGE];
#pragma unroll
            for (int q = 0; q < K3_STAGE; q++) {
                int iw = __builtin_amdgcn_readfirstlane(i0 - lane + q * F_BLOCK);     // the wave's first piece
                iw = iw < total16 ? iw : total16 - 64;
                const int r = iw / n16, j0 = iw - r * n16;
                const uint8_t *__restrict__ tab = tabs[rowdesc[r] & F3_CHROM_MASK].tab;
                const uint4 t = *reinterpret_cast<const uint4 *>(tab + base + (int64_t)(j0 + lane) * 16);
                vx[q] = t.x; vy[q] = t.y; vz[q] = t.z; vw[q] = t.w;
            }
#pragma unroll
            for (int q = 0; q < K3_STAGE; q++) {
                const int i = i0 + q * F_BLOCK;
                if (i < total16) reinterpret_cast<uint4 *>(tile)[i] = make_uint4(vx[q], vy[q], vz[q], vw[q]);
            }
        }
        if (!whole)
        for (int i0 = threadIdx.x; i0 < total16; i0 += 8 * F_BLOCK) {
            uint4 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = i0 + q * F_BLOCK;
                v[q] = make_uint4(0, 0, 0, 0);
                if (i < total16) {
                    const int r = i / n16, j = i - r * n16;
                    const uint8_t *__restrict__ tab = tabs[rowdesc[r] & F3_CHROM_MASK].tab;
                    const int64_t s0 = base + (int64_t)j * 16;
                    if (s0 + 16 <= P.nslots) {
                        v[q] = *reinterpret_cast<const uint4 *>(tab + s0);
                    } else if (s0 < P.nslots) {   // ragged end of a tiny table
                        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
                        for (int b = 0; b < 16 && s0 + b < P.nslots; b++) {
                            const uint32_t y = (uint32_t)tab[s0 + b] << (8 * (b & 3));
                            if (b < 4) w0 |= y; else if (b < 8) w1 |= y; else if (b < 12) w2 |= y; else w3 |= y;
                        }
                        v[q] = make_uint4(w0, w1, w2, w3);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = i0 + q * F_BLOCK;
                if (i < total16) reinterpret_cast<uint4 *>(tile)[i] = v[q];
            }
        }
        __syncthreads();
        // ---- which slots hold a count >= lower in any chromosome (4 slots per word), and in how many
        // non-singleton sets: a set without any count has hi = 0 and fails the fold test (0 / 1e-20 < min_fold),
        // so a slot active in fewer than `need_active` sets cannot reach include / _all >= ratio -- it is
        // counted for the union and never evaluated
        const uint32_t *tw = reinterpret_cast<const uint32_t *>(tile);
        const int nw = TS / 4;
        for (int g = threadIdx.x; g < nw; g += F_BLOCK) {
            uint32_t any = 0, is255 = 0, qual;
            if (SWAR) {
                uint32_t act = 0, m = 0;
                if (NCH > 0) {
                    const uint32_t *col = tw + g;
                    for (int r0 = 0; r0 < R; r0 += K3_SCAN_CH) {         // R is uniform: a scalar loop
                        uint32_t x[K3_SCAN_CH];
#pragma unroll
                        for (int i = 0; i < K3_SCAN_CH; i++) x[i] = col[(r0 + i < R ? r0 + i : R - 1) * nw];     // independent LDS reads
#pragma unroll
                        for (int i = 0; i < K3_SCAN_CH; i++) {
                            if (r0 + i < R) {
                                const uint32_t lo = x[i] & 0x7f7f7f7fu;
                                m |= (lo + addL) | x[i];        // bit 7 of a byte: >= lower (the other bits are masked at the set's end)
                                is255 |= (lo + 0x01010101u) & x[i];
                                if ((mk_set >> (r0 + i)) & 1u) {
                                    asm volatile("");       // a real (scalar) branch: if-converted, every row pays the six instructions of a set's end
                                    m &= 0x80808080u;
                                    any |= m;
                                    act += m >> 7;
                                    m = 0;
                                }
                            }
                        }
                    }
                    any |= m & 0x80808080u;                 // rows behind the last set: chromosomes of no non-singleton set
                } else {
                    for (int r = 0; r < R; r++) {
                        const uint32_t x = tw[r * nw + g];
                        const uint32_t ge = k3_ge_mask(x, addL);
                        any |= ge;
                        is255 |= ((x & 0x7f7f7f7fu) + 0x01010101u) & x;
                        m |= ge;
                        if (rowdesc[r] & F3_SET_END) {
                            act += m >> 7;
                            m = 0;
                        }
                    }
                }
                is255 &= 0x80808080u;
                nuni += __popc(any);
                qual = P.need_active > 0 ? any & k3_ge_mask(act, (0x80u - (uint32_t)P.need_active) * 0x01010101u) : any;
            } else {
                for (int r = 0; r < R; r++) {
                    const uint32_t x = tw[r * nw + g];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t y = (x >> (8 * b)) & 255u;
                        if (y == 255u) is255 |= 0x80u << (8 * b);
                        if (y == 255u || y >= P.lower) any |= 0x80u << (8 * b);   // 255: decided exactly below
                    }
                }
                qual = any;
            }
            // columns with a saturated byte (or every column when the rows are not unrolled) take the generic decision
            const uint32_t slow = (SWAR && NCH > 0 && P.fast) ? is255 : 0x80808080u;
            // queue the qualifying slots: ONE reservation per wave and queue (it was one per byte lane of the word -- four returning
            // LDS atomics and four cross-lane broadcasts per wave, one after the other).  A wave of this loop is whole (TS / 4 is a
            // multiple of 64), so lane 0 is there to reserve.
            const uint32_t q_fast = qual & ~slow & 0x80808080u, q_slow = qual & slow & 0x80808080u;
            {
                unsigned long long bal[4];
                uint32_t n = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    bal[b] = __ballot((q_fast >> (8 * b + 7)) & 1u);
                    n += (uint32_t)__popcll(bal[b]);
                }
                if (n) {
                    uint32_t qb = 0;
                    if (lane == 0) qb = atomicAdd(&s_qn, n);
                    qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if ((q_fast >> (8 * b + 7)) & 1u)
                            queue[qb + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[b] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[b], 0u))] = (uint16_t)(4 * g + b);
                        qb += (uint32_t)__popcll(bal[b]);
                    }
                }
            }
            if (__ballot(q_slow != 0u)) {
                unsigned long long bal[4];
                uint32_t n = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    bal[b] = __ballot((q_slow >> (8 * b + 7)) & 1u);
                    n += (uint32_t)__popcll(bal[b]);
                }
                uint32_t qb = 0;
                if (lane == 0) qb = atomicAdd(&s_q2n, n);
                qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if ((q_slow >> (8 * b + 7)) & 1u)
                        queue2[qb + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[b] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[b], 0u))] = (uint16_t)(4 * g + b);
                    qb += (uint32_t)__popcll(bal[b]);
                }
            }
        }
        __syncthreads();
        // ---- decide the qualifying slots, one per lane.  The rows lie in config order, so the decision is
        // ONE walk down the column: bytes are fetched eight at a time (independent LDS reads), unit and set
        // boundaries are uniform flags in scalar registers -- no pointer chasing, no divergence between lanes.
        // _filter_kmer (Jellyfish.py:611-648) for baseline 1 / -1: running max, second max and min of the
        // unit frequencies, in fp32 on reciprocal products.  fp32 moves hi and lo by a relative 1e-6 at
        // most: outside a 1e-5 band around the threshold this and the reference's fp64 quotient test
        // agree; a k-mer with any set inside the band goes to the exact fp64 code (slow queue).
        if (NCH > 0) {
            const uint32_t qn = s_qn;
            for (uint32_t q0 = 0; q0 < qn; q0 += F_BLOCK) {
                // a tile queues ~1 slot in 10: the last waves of the block have nothing to decide (the kernel is bound by
                // instruction issue: 245 instructions per wave and walk)
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(q0 + (threadIdx.x & ~63u))) >= qn) continue;
                const uint32_t q = q0 + threadIdx.x;
                const bool live = q < qn;
                const int sl = live ? (int)queue[q] : 0;
                int include = 0;
                bool exact = false;
                uint32_t tot = 0, num = 0;
                float m1 = -1.0f, m2 = -1.0f, mn = 3e38f;
                for (int r0 = 0; r0 < R; r0 += K3_WALK_CH) {
                    uint32_t y[K3_WALK_CH];
#pragma unroll
                    for (int i = 0; i < K3_WALK_CH; i++) y[i] = (uint32_t)tile[(size_t)(r0 + i < R ? r0 + i : R - 1) * TS + sl];
#pragma unroll
                    for (int i = 0; i < K3_WALK_CH; i++) {
                        const int r = r0 + i;
                        if (r < R) {
                            const uint32_t c = y[i] >= P.lower ? y[i] : 0u;
                            if ((mk_tot >> r) & 1u) tot += c;            // uniform branches on register bits
                            num += c;
                            if ((mk_unit >> r) & 1u) {
                                asm volatile("");
                                const float x = (float)num * __int_as_float(__builtin_amdgcn_readlane(inv_bits, r));
                                m2 = fmaxf(m2, fminf(m1, x));            // running max / second max / min, branch-free
                                m1 = fmaxf(m1, x);
                                mn = fminf(mn, x);
                                num = 0;
                            }
                            if ((mk_set >> r) & 1u) {
                                asm volatile("");
                                const float thr = fold32 * ((((mk_bi1 >> r) & 1u) ? m2 : mn) + 1e-20f);
                                const bool pass = m1 > thr * (1.0f + 1e-5f);
                                include += pass ? 1 : 0;
                                exact = exact || (!pass && !(m1 < thr * (1.0f - 1e-5f)));
                                m1 = -1.0f; m2 = -1.0f; mn = 3e38f;
                            }
                        }
                    }
                }
                if (live && tot) {
                    if (exact) {
                        queue2[atomicAdd(&s_q2n, 1u)] = (uint16_t)sl;
                    } else if (include >= P.need_hist) {   // == !(include / _all < ratio), :642-644
                        atomicOr(&bmh[sl >> 5], 1u << (sl & 31));
                        nhist++;
                        const double t = (double)tot;
                        if (!(t < P.min_freq || t > P.max_freq)) {  // :645-646
                            atomicOr(&bmr[sl >> 5], 1u << (sl & 31));
                            nrow++;
                        }
                    }
                }
            }
            __syncthreads();
        }
        // ---- the generic decision: saturated columns (counts >= 255 come from the overflow lists), k-mers on
        // the fp32 band, baselines other than 1 / -1, more rows than the unrolled walk holds
        // Slow slots leave the streaming kernel: the column's bytes go to a global queue that k3_slow works off with
        // the whole machine (their decisions start with binary searches in the overflow lists -- chains of ~17
        // dependent global loads; decided here, a handful of them per tile kept every wave of the block waiting at the
        // barrier and tripled the kernel's time).  Only when the queue is full are they decided in place.
        const uint32_t q2n = s_q2n;      // (block-uniform: read behind a barrier, reset behind the tile's last one)
        if (q2n) {
            const int EW = 1 + (R + 3) / 4;
            // ONE reservation per tile, not one returning same-address atomic per slow slot
            if (threadIdx.x == 0) {
                s_q3n = 0;
                s_gq0 = (P.gq && q2n) ? atomicAdd(P.gq_n, (unsigned long long)q2n) : ~0ULL;
            }
            __syncthreads();
            const unsigned long long gq0 = s_gq0;
            for (uint32_t q = threadIdx.x; q < q2n; q += F_BLOCK) {
                const int sl = (int)queue2[q];
                const unsigned long long pos = gq0 == ~0ULL ? ~0ULL : gq0 + q;
                if (pos < P.gq_cap) {
                    uint32_t *e = P.gq + pos * EW;
                    e[0] = (uint32_t)(base + sl);
                    for (int r0 = 0; r0 < R; r0 += 4) {
                        uint32_t w = 0;
                        for (int i = 0; i < 4 && r0 + i < R; i++) w |= (uint32_t)tile[(size_t)(r0 + i) * TS + sl] << (8 * i);
                        e[1 + r0 / 4] = w;
                    }
                } else {
                    queue[atomicAdd(&s_q3n, 1u)] = (uint16_t)sl;   // the fast queue has been consumed: reuse it
                }
            }
            __syncthreads();
            const uint32_t q3n = s_q3n;
            const int wave = threadIdx.x >> 6;
            uint32_t *mine = resolved + (size_t)wave * R;
            for (uint32_t q = wave; q < q3n; q += F_BLOCK / 64) {   // one WAVE per slot: rows resolved in parallel
                const int sl = (int)queue[q];
                for (int r = lane; r < R; r += 64) {
                    uint32_t y = tile[(size_t)r * TS + sl];
                    if (y == 255u) y = sp_ovf_lookup(l_ovf[r], (int64_t)l_novf[r], (uint32_t)(P.slot_base + base + sl));
                    mine[r] = y >= P.lower ? y : 0u;
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
                if (lane == 0) {
                    auto cnt_chrom = [&](int c) -> uint32_t { return mine[P.row_of_chrom[c]]; };
                    unsigned long long tot = 0;
                    for (int r = 0; r < R; r++)
                        if (rowdesc[r] & F3_TOT) tot += mine[r];
                    if (tot) {
                        if (!SWAR) nuni++;
                        bool is_row = false, is_hist = false;
                        sp_filter_decide(cnt_chrom, tot, F, is_row, is_hist);
                        if (is_row) { atomicOr(&bmr[sl >> 5], 1u << (sl & 31)); nrow++; }
                        if (is_hist) { atomicOr(&bmh[sl >> 5], 1u << (sl & 31)); nhist++; }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TS / 64; i += F_BLOCK) {
            const int64_t g = (base >> 6) + i;
            if ((g << 6) < P.nslots) {
                bm_row[g] = (unsigned long long)bmr[2 * i] | ((unsigned long long)bmr[2 * i + 1] << 32);
                bm_hist[g] = (unsigned long long)bmh[2 * i] | ((unsigned long long)bmh[2 * i + 1] << 32);
            }
        }
        __syncthreads();
    }
    unsigned long long t_row = sp_block_sum_u64(nrow, red);
    unsigned long long t_hist = sp_block_sum_u64(nhist, red);
    unsigned long long t_uni = sp_block_sum_u64(nuni, red);
    if (threadIdx.x == 0) {
        blk_row[blockIdx.x] = t_row;
        blk_hist[blockIdx.x] = t_hist;
        if (t_uni) atomicAdd(n_union, t_uni);
    }
}

// The slow queue of k3_eval: one wave per entry.  Lanes resolve the rows in parallel (overflow-list searches),
// lane 0 takes the generic decision and sets the bitmap bits / per-block tallies k3_eval left for it.
__global__ void __launch_bounds__(256)
k3_slow(const sp_tabref *__restrict__ tabs, sp_filter_params P, unsigned long long n_entries,
        const int32_t *__restrict__ set_off, const int32_t *__restrict__ unit_off,
        const int32_t *__restrict__ unit_chrom, const double *__restrict__ unit_den,
        unsigned long long *__restrict__ bm_row, unsigned long long *__restrict__ bm_hist,
        unsigned long long *__restrict__ blk_row, unsigned long long *__restrict__ blk_hist,
        unsigned long long *__restrict__ n_union, int count_union) {
    extern __shared__ uint32_t res[];   // [4 waves][R]
    const int R = P.R, EW = 1 + (R + 3) / 4, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *mine = res + (size_t)wave * R;
    sp_fsets F;
    F.n_sets = P.n_sets;
    F.n_multi = P.n_multi;
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
    for (unsigned long long q = (unsigned long long)blockIdx.x * 4 + wave; q < n_entries; q += (unsigned long long)gridDim.x * 4) {
        const uint32_t *e = P.gq + q * EW;
        const uint32_t sl = e[0];   // local slot of the view
        for (int r = lane; r < R; r += 64) {
            uint32_t y = (e[1 + r / 4] >> (8 * (r & 3))) & 255u;
            if (y == 255u) {
                const sp_tabref t = tabs[P.rowdesc[r] & F3_CHROM_MASK];
                y = sp_ovf_lookup_t(t, (uint32_t)(P.slot_base + sl));
            }
            mine[r] = y >= P.lower ? y : 0u;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (lane == 0) {
            auto cnt_chrom = [&](int c) -> uint32_t { return mine[P.row_of_chrom[c]]; };
            unsigned long long tot = 0;
            for (int r = 0; r < R; r++)
                if (P.rowdesc[r] & F3_TOT) tot += mine[r];
            if (tot) {
                if (count_union) atomicAdd(n_union, 1ULL);
                bool is_row = false, is_hist = false;
                sp_filter_decide(cnt_chrom, tot, F, is_row, is_hist);
                if (is_row) {
                    atomicOr(&bm_row[sl >> 6], 1ULL << (sl & 63));
                    atomicAdd(&blk_row[sl / F_SLOTS_PER_BLOCK], 1ULL);
                }
                if (is_hist) {
                    atomicOr(&bm_hist[sl >> 6], 1ULL << (sl & 63));
                    atomicAdd(&blk_hist[sl / F_SLOTS_PER_BLOCK], 1ULL);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
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
k3_emit(const sp_tabref *__restrict__ tabs, int C, uint32_t lower, int64_t M, int64_t slot_base, sp_kparams kp,
        const uint32_t *__restrict__ slots, const double *__restrict__ chrom_len,
        unsigned long long *__restrict__ keys, uint32_t *__restrict__ counts,
        double *__restrict__ freqs, unsigned long long *__restrict__ tots) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const int64_t slot = slots[r];
    unsigned long long tot = 0;
    for (int c = 0; c < C; c++) {
        uint32_t v = sp_tab_count(tabs[c], slot, slot_base);
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

static int filter_C(sp_ctx *ctx) {
    return ctx->sv_on ? (int)ctx->sv_keys.size() : ctx->fv_on ? (int)ctx->fv_tabs.size() : (int)ctx->chroms.size();
}
static sp_tabref filter_tab(sp_ctx *ctx, int i) {
    if (ctx->fv_on) return ctx->fv_tabs[(size_t)i];
    const sp_chrom &c = ctx->chroms[(size_t)i];
    sp_tabref t;
    t.tab = c.d_tab;
    t.ovf = c.d_ovf;
    t.n_ovf = c.n_ovf;
    t.ovf_idx = (c.ovf_idx_n > 0 && c.ovf_idx_n == ((ctx->nslots + (1LL << SP_OVF_SHIFT) - 1) >> SP_OVF_SHIFT) + 1) ? c.d_ovf_idx : nullptr;
    return t;
}
static int64_t filter_len(sp_ctx *ctx, int i) {
    return (ctx->fv_on || ctx->sv_on) ? ctx->fv_lengths[(size_t)i] : ctx->chroms[(size_t)i].length_sum;
}
static int64_t filter_nslots(sp_ctx *ctx) { return ctx->fv_on ? ctx->fv_nslots : ctx->nslots; }
static int64_t filter_base(sp_ctx *ctx) { return ctx->fv_on ? ctx->fv_slot_base : 0; }

static int upload_tabs(sp_ctx *ctx, const sp_tabref **d_tabs, double **d_len) {
    const size_t C = (size_t)filter_C(ctx);
    size_t bytes = C * sizeof(sp_tabref) + C * sizeof(double);
    void *scr = nullptr;
    int rc = sp_scratch(ctx, (int64_t)bytes + 4096, &scr);
    if (rc) return rc;
    std::vector<sp_tabref> h(C);
    std::vector<double> hl(C);
    for (size_t i = 0; i < C; i++) {
        h[i] = filter_tab(ctx, (int)i);
        hl[i] = (double)filter_len(ctx, (int)i);
    }
    SP_HIP(ctx, hipMemcpyAsync(scr, h.data(), C * sizeof(sp_tabref), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync((char *)scr + C * sizeof(sp_tabref), hl.data(), C * sizeof(double),
                              hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // h/hl go out of scope
    *d_tabs = (const sp_tabref *)scr;
    *d_len = (double *)((char *)scr + C * sizeof(sp_tabref));
    return SP_OK;
}

int sp_sparse_filter(sp_ctx *ctx, int n_sets, const int32_t *set_off, const int32_t *unit_off,
                     const int32_t *unit_chrom, const std::vector<double> &den, double min_fold, int baseline,
                     double min_freq, double max_freq, double ratio);                      // sp_sparse.hip
int sp_sparse_fetch(sp_ctx *ctx, bool hist, uint64_t *keys, uint32_t *counts, double *freqs, uint64_t *tot, bool async);

extern "C" {

int sp_filter_view(sp_ctx *ctx, int C, const void *const *d_tabs, int64_t slot_base, int64_t nslots_view,
                   const int64_t *lengths, int k, int lower_count, const void *const *d_ovf, const int64_t *n_ovf) {
    if (!ctx) return SP_EINVAL;
    if (d_tabs && (ctx->sparse_mode || ctx->list_mode)) return sp_fail(ctx, SP_EUNSUP, "sp_filter_view: byte-table engines (k <= 15) only");
    if (!d_tabs) {   // back to the local chromosomes
        ctx->fv_on = false;
        ctx->fv_tabs.clear();
        ctx->fv_lengths.clear();
        ctx->filtered = false;
        return SP_OK;
    }
    if (C <= 0 || !lengths || slot_base < 0 || nslots_view <= 0 || (slot_base % 64) != 0 || k < 1 || k > 15)
        return sp_fail(ctx, SP_EINVAL, "sp_filter_view: bad arguments (slot_base must be a multiple of 64)");
    ctx->fv_tabs.assign((size_t)C, sp_tabref());
    ctx->fv_lengths.assign((size_t)C, 0);
    for (int i = 0; i < C; i++) {
        if (!d_tabs[i]) return sp_fail(ctx, SP_EINVAL, "sp_filter_view: table %d is NULL", i);
        if ((uintptr_t)d_tabs[i] & 15) return sp_fail(ctx, SP_EINVAL, "sp_filter_view: table %d is not 16-byte aligned", i);
        sp_tabref t;
        t.tab = (const uint8_t *)d_tabs[i];
        t.ovf = (d_ovf && n_ovf && n_ovf[i] > 0) ? (const uint2 *)d_ovf[i] : nullptr;
        t.n_ovf = (d_ovf && n_ovf && n_ovf[i] > 0) ? n_ovf[i] : 0;
        if (t.n_ovf && !t.ovf) return sp_fail(ctx, SP_EINVAL, "sp_filter_view: overflow list %d is NULL", i);
        ctx->fv_tabs[(size_t)i] = t;
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
    // rows of the previous filter call still on their way to the host (sp_filter_fetch_async without its wait): this
    // call rewrites the buffers they are read from
    if (ctx->copy_stream) SP_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
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

    if (ctx->sparse_mode || ctx->list_mode) {
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
    size_t tot_b = al(b_set) + al(b_uo) + al(b_uc) + al(b_den) + al(C * sizeof(sp_tabref)) + 256;
    int rcb = sp_buf_ensure(ctx, ctx->b_fpar, (int64_t)tot_b);
    if (rcb) return rcb;
    char *d_par = (char *)ctx->b_fpar.p;
    char *p = d_par;
    int32_t *d_set = (int32_t *)p; p += al(b_set);
    int32_t *d_uo = (int32_t *)p; p += al(b_uo);
    int32_t *d_uc = (int32_t *)p; p += al(b_uc);
    double *d_den = (double *)p; p += al(b_den);
    sp_tabref *d_tabs = (sp_tabref *)p; p += al(C * sizeof(sp_tabref));
    unsigned long long *d_nuni = (unsigned long long *)p;   // [0] union count, [1..2] scan totals
    std::vector<sp_tabref> htabs((size_t)C);
    for (int i = 0; i < C; i++) htabs[(size_t)i] = filter_tab(ctx, i);
    hipError_t e = hipSuccess;
    auto cp = [&](void *d, const void *h, size_t n) {
        if (e == hipSuccess && n) e = hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->stream);
    };
    cp(d_set, set_off, b_set);
    cp(d_uo, unit_off, b_uo);
    cp(d_uc, unit_chrom, (size_t)n_uc * 4);
    cp(d_den, den.data(), b_den);
    cp(d_tabs, htabs.data(), C * sizeof(sp_tabref));
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
    P.slot_base = filter_base(ctx);
    // LDS tile: the largest power of two of slots whose C byte rows fit 48 KiB (three blocks per CU)
    int TS = F_SLOTS_PER_BLOCK;
    while (TS > 256 && (size_t)C * TS > F_TILE_BYTES) TS >>= 1;
    if ((size_t)C * TS > 140 * 1024)
        return sp_fail(ctx, SP_EUNSUP, "sp_filter: %d chromosomes exceed the LDS staging budget (at most 560)", C);
    P.TS = TS;
    P.n_multi = n_sets - n_single;
    P.need_hist = 0;    // smallest include with !(include / _all < ratio): the quotient is monotone in include
    while (P.need_hist <= P.n_multi && (double)P.need_hist / (double)P.n_multi < ratio) P.need_hist++;
    P.need_active = (min_fold > 0 && P.n_multi <= 126) ? P.need_hist : 0;
    // LDS rows in config order: the chromosomes of the non-singleton sets, then the rest
    std::vector<int32_t> rowdesc, row_of_chrom((size_t)C, -1);
    std::vector<float> rowinv;
    P.fast = 1;
    for (int st = 0; st < n_sets; st++) {
        const int nu = set_off[st + 1] - set_off[st];
        if (nu == 1) continue;
        const int bi = baseline < 0 ? nu + baseline : baseline;
        if (!(bi == 1 || bi == nu - 1)) P.fast = 0;
        for (int u = set_off[st]; u < set_off[st + 1]; u++) {
            if (unit_off[u + 1] == unit_off[u]) P.fast = 0;   // a unit without chromosomes: generic code only
            for (int j = unit_off[u]; j < unit_off[u + 1]; j++) {
                const int c = unit_chrom[j];
                int d = c;
                if (row_of_chrom[(size_t)c] < 0) {
                    row_of_chrom[(size_t)c] = (int32_t)rowdesc.size();
                    d |= F3_TOT;
                }
                if (j == unit_off[u + 1] - 1) d |= F3_UNIT_END;
                if (j == unit_off[u + 1] - 1 && u == set_off[st + 1] - 1) d |= F3_SET_END;
                if (bi == 1) d |= F3_BI1;
                rowdesc.push_back(d);
                rowinv.push_back((float)den[(size_t)(n_units + u)]);
            }
        }
    }
    P.R_sets = (int)rowdesc.size();
    for (int c = 0; c < C; c++)
        if (row_of_chrom[(size_t)c] < 0) {
            row_of_chrom[(size_t)c] = (int32_t)rowdesc.size();
            rowdesc.push_back(c | F3_TOT);
            rowinv.push_back(0.0f);
        }
    P.R = (int)rowdesc.size();
    if (C > F3_CHROM_MASK) return sp_fail(ctx, SP_EUNSUP, "sp_filter: too many chromosomes");
    // LDS tile: the largest power of two of slots whose R byte rows fit the budget (several blocks per CU)
    TS = F_SLOTS_PER_BLOCK;
    while (TS > 256 && (size_t)P.R * TS > F_TILE_BYTES) TS >>= 1;
    if ((size_t)P.R * TS > 140 * 1024)
        return sp_fail(ctx, SP_EUNSUP, "sp_filter: %d chromosome rows exceed the LDS staging budget (at most 560)", P.R);
    P.TS = TS;
    {
        size_t nR = ((size_t)P.R + 7) & ~(size_t)7;   // padded with neutral rows: the unrolled walk reads whole chunks
        if (nR < 32) nR = 32;
        std::vector<int32_t> blob(2 * nR + (size_t)C, 0);
        memcpy(blob.data(), rowdesc.data(), (size_t)P.R * 4);
        memcpy(blob.data() + nR, rowinv.data(), (size_t)P.R * 4);
        memcpy(blob.data() + 2 * nR, row_of_chrom.data(), (size_t)C * 4);
        int rcf = sp_buf_ensure(ctx, ctx->b_fflat, (int64_t)blob.size() * 4 + 64);
        if (rcf) return rcf;
        SP_HIP(ctx, hipMemcpyAsync(ctx->b_fflat.p, blob.data(), blob.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `blob` goes out of scope
        P.rowdesc = (const int32_t *)ctx->b_fflat.p;
        P.rowinv = (const float *)(P.rowdesc + nR);
        P.row_of_chrom = P.rowdesc + 2 * nR;
    }
    // global slow queue (saturated columns, fp32-band k-mers): generously sized, never required (overflow is decided inline)
    {
        const int EW = 1 + (P.R + 3) / 4;
        unsigned long long cap = (unsigned long long)(nslots / 64 + 4096);
        int rq = sp_buf_ensure(ctx, ctx->b_fq, (int64_t)(cap * EW * 4 + 64));
        if (rq) return rq;
        P.gq = (uint32_t *)((char *)ctx->b_fq.p + 64);
        P.gq_n = (unsigned long long *)ctx->b_fq.p;
        P.gq_cap = cap;
        SP_HIP(ctx, hipMemsetAsync(ctx->b_fq.p, 0, 64, ctx->stream));
    }
    const size_t shmem = (size_t)P.R * TS + (size_t)TS * 4 + (size_t)TS / 4 + (size_t)P.R * 16 + (size_t)P.R * 4 * F_WAVES;
    const bool swar = ctx->lower >= 1 && ctx->lower <= 128;
    const int nch = (swar && P.fast && P.R <= 32) ? 1 : 0;
#define K3_LAUNCH(SW, N)                                                                                          \
    do {                                                                                                          \
        SP_HIP(ctx, hipFuncSetAttribute((const void *)k3_eval<SW, N>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)shmem));                                                             \
        SP_LAUNCH(ctx, "k3_eval", (k3_eval<SW, N>), dim3((unsigned)nblk), dim3(F_BLOCK), shmem,                   \
                  (const sp_tabref *)d_tabs, P, d_set, d_uo, d_uc, d_den, (unsigned long long *)ctx->d_flag_row,  \
                  (unsigned long long *)ctx->d_flag_hist, (unsigned long long *)ctx->d_blk_row,                   \
                  (unsigned long long *)ctx->d_blk_hist, d_nuni);                                                 \
    } while (0)
    if (!swar) K3_LAUNCH(false, 0);
    else if (nch == 1) K3_LAUNCH(true, 1);
    else K3_LAUNCH(true, 0);
#undef K3_LAUNCH
    {   // the slow queue: its length comes back with the other totals; launched over the capacity-bounded count
        unsigned long long hq = 0;
        SP_HIP(ctx, hipMemcpyAsync(&hq, P.gq_n, 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hq > P.gq_cap) hq = P.gq_cap;
        if (hq > 0) {
            unsigned long long nb = (hq + 3) / 4;
            if (nb > (unsigned long long)ctx->n_cu * 32) nb = (unsigned long long)ctx->n_cu * 32;
            SP_LAUNCH(ctx, "k3_slow", k3_slow, dim3((unsigned)nb), dim3(256), (size_t)P.R * 16, (const sp_tabref *)d_tabs, P, hq,
                      d_set, d_uo, d_uc, d_den, (unsigned long long *)ctx->d_flag_row, (unsigned long long *)ctx->d_flag_hist,
                      (unsigned long long *)ctx->d_blk_row, (unsigned long long *)ctx->d_blk_hist, d_nuni, swar ? 0 : 1);
        }
    }
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
                       uint64_t *tot, int64_t cap, bool async = false) {
    if (!ctx->filtered) return sp_fail(ctx, SP_EINVAL, "call sp_filter first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t M = hist ? ctx->n_hist : ctx->n_rows;
    if (cap < M) return sp_fail(ctx, SP_EINVAL, "capacity %lld < %lld rows", (long long)cap, (long long)M);
    if (M == 0) return SP_OK;
    if (ctx->sparse_mode || ctx->list_mode) return sp_sparse_fetch(ctx, hist, keys, counts, freqs, tot, async);
    const int C = filter_C(ctx);
    const sp_tabref *d_tabs = nullptr;
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
    // async: the rows are gathered on the compute stream, then a copy stream takes them to the (page-locked) host
    // buffers while the compute stream goes on with the map stage; sp_filter_fetch_wait joins the two
    hipStream_t cs = ctx->stream;
    if (async) {
        if (!ctx->copy_stream) {
            SP_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            SP_HIP(ctx, hipEventCreateWithFlags(&ctx->copy_event, hipEventDisableTiming));
        }
        SP_HIP(ctx, hipEventRecord(ctx->copy_event, ctx->stream));
        SP_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->copy_event, 0));
        cs = ctx->copy_stream;
    }
    if (keys) SP_HIP(ctx, hipMemcpyAsync(keys, d_keys, (size_t)M * 8, hipMemcpyDeviceToHost, cs));
    if (tot) SP_HIP(ctx, hipMemcpyAsync(tot, d_tot, (size_t)M * 8, hipMemcpyDeviceToHost, cs));
    if (counts) SP_HIP(ctx, hipMemcpyAsync(counts, d_counts, (size_t)M * C * 4, hipMemcpyDeviceToHost, cs));
    if (freqs) SP_HIP(ctx, hipMemcpyAsync(freqs, d_freqs, (size_t)M * C * 8, hipMemcpyDeviceToHost, cs));
    if (!async) SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_filter_fetch(sp_ctx *ctx, uint64_t *keys, uint32_t *counts, double *freqs, uint64_t *tot,
                    int64_t cap_rows) {
    if (!ctx) return SP_EINVAL;
    return emit_common(ctx, false, keys, counts, freqs, tot, cap_rows);
}

int sp_filter_fetch_async(sp_ctx *ctx, uint64_t *keys, uint32_t *counts, uint64_t *tot, int64_t cap_rows) {
    if (!ctx) return SP_EINVAL;
    return emit_common(ctx, false, keys, counts, nullptr, tot, cap_rows, true);
}

int sp_filter_fetch_wait(sp_ctx *ctx) {
    if (!ctx) return SP_EINVAL;
    if (ctx->copy_stream) SP_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    return SP_OK;
}

int sp_filter_fetch_device(sp_ctx *ctx, void *d_keys, void *d_counts, void *d_tot, int64_t cap_rows) {
    if (!ctx) return SP_EINVAL;
    if (!ctx->filtered) return sp_fail(ctx, SP_EINVAL, "call sp_filter first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t M = ctx->n_rows;
    if (cap_rows < M) return sp_fail(ctx, SP_EINVAL, "capacity %lld < %lld rows", (long long)cap_rows, (long long)M);
    if (M == 0) return SP_OK;
    const int C = filter_C(ctx);
    if (ctx->sparse_mode || ctx->list_mode) {   // the list filter leaves its rows in device buffers already
        if (d_keys) SP_HIP(ctx, hipMemcpyAsync(d_keys, ctx->b_sf_keys.p, (size_t)M * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_counts) SP_HIP(ctx, hipMemcpyAsync(d_counts, ctx->b_sf_counts.p, (size_t)M * C * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_tot) SP_HIP(ctx, hipMemcpyAsync(d_tot, ctx->b_sf_tot.p, (size_t)M * 8, hipMemcpyDeviceToDevice, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return SP_OK;
    }
    const sp_tabref *d_tabs = nullptr;
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
