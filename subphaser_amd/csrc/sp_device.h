// sp_device.h -- device-side helpers shared by the counting and mapping kernels.
#pragma once
#include "sp_common.h"

// Per-thread scan unit: 64 consecutive k-mer START positions [s0, s0+64),
// s0 a multiple of 64 (so the unit is 4 code words = one 16-B load, and 2 mask
// words).  The unit also reads the k-1 bases to its right (halo), which live
// in at most 2 more code words for k <= 32.  Starting with run = 0 at s0 makes
// the first emitted window the one that STARTS at s0, so no left halo is
// needed and every start position is visited exactly once -- the same
// exactly-once rule the reference obtains with its k-1 chunk overlap
// (Seqs.py:121-139).
#define SP_UNIT 64

// 32-bit k-mer arithmetic for the dense-table kernels (2k <= 32 bits)
struct sp_kparams32 {
    int k, odd, rcshift;
    uint32_t kmask;
};
__host__ __device__ inline sp_kparams32 sp_make_kparams32(int k) {
    sp_kparams32 p;
    p.k = k;
    p.odd = k & 1;
    p.rcshift = 2 * (k - 1);
    p.kmask = (k >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    return p;
}
__device__ __forceinline__ uint32_t sp_slot_of32(uint32_t fwd, uint32_t rc, const sp_kparams32 &p) {
    if (p.odd) {
        const uint32_t rep = ((fwd >> p.k) & 1u) ? rc : fwd;
        const uint32_t lowmask = (1u << p.k) - 1u;
        return (rep & lowmask) | ((rep >> (p.k + 1)) << p.k);
    }
    return fwd < rc ? fwd : rc;
}

// bit b of the result is set iff an invalid base lies in the window of `w` bases that ENDS at base b
// (bases b-w+1 .. b) -- the validity of every window of a unit from a handful of shifts instead of a
// run counter updated at every base (3 VALU instructions per base in the scan-bound kernels)
__device__ __forceinline__ unsigned __int128 sp_bad_windows(unsigned __int128 m, int w) {
    if (w <= 0) return 0;
    unsigned __int128 e = m;
    int have = 1;
    while (have * 2 <= w) {
        e |= e << have;
        have *= 2;
    }
    if (have < w) e |= e << (w - have);
    return e;
}

// compile-time twin of sp_slot_of32: the kernels branch once per unit on the (uniform) parity of k instead of
// computing both forms and selecting at every k-mer
template <bool ODD>
__device__ __forceinline__ uint32_t sp_slot_of32_t(uint32_t fwd, uint32_t rc, const sp_kparams32 &p) {
    if (ODD) {
        const uint32_t rep = ((fwd >> p.k) & 1u) ? rc : fwd;
        const uint32_t lowmask = (1u << p.k) - 1u;
        return (rep & lowmask) | ((rep >> (p.k + 1)) << p.k);
    }
    return fwd < rc ? fwd : rc;
}
struct sp_odd_tag { static constexpr bool value = true; };
struct sp_even_tag { static constexpr bool value = false; };

// UNIT consecutive k-mer START positions [s0, s0+UNIT), s0 a multiple of UNIT (UNIT = 32 or 64),
// emit(start, fwd, rc) for every start whose k bases are all valid.
template <int UNIT, typename KeyT, typename KP, typename F>
__device__ __forceinline__ void sp_scan_unit_t(const uint32_t *__restrict__ pk,
                                               const uint32_t *__restrict__ nm, int64_t s0,
                                               const KP &kp, F &&emit) {
    constexpr int MW = UNIT / 16;  // code words owned by the unit
    const int64_t w0 = s0 >> 4;
    uint32_t words[MW + 2];
    if (MW == 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(pk + w0);
        words[0] = v.x; words[1] = v.y; words[2] = v.z; words[3] = v.w;
    } else {
        const uint2 v = *reinterpret_cast<const uint2 *>(pk + w0);
        words[0] = v.x; words[1] = v.y;
    }
    words[MW] = pk[w0 + MW];
    words[MW + 1] = pk[w0 + MW + 1];
    const uint64_t mlo = (uint64_t)nm[s0 >> 5] | ((uint64_t)nm[(s0 >> 5) + 1] << 32);
    const uint32_t mhi = (MW == 4) ? nm[(s0 >> 5) + 2] : 0u;
    KeyT fwd = 0, rc = 0;
    const int k = kp.k;
    const int total = UNIT + k - 1;  // bases to consume (k - 1 <= 31 halo bases: two extra words)
    // emit mask: bit b set iff the k-mer ENDING at base b starts inside the unit and has no invalid base
    const unsigned __int128 m128 = (unsigned __int128)mlo | ((unsigned __int128)mhi << 64);
    const unsigned __int128 range = ((((unsigned __int128)1 << total) - 1) >> (k - 1)) << (k - 1);
    const unsigned __int128 em = ~sp_bad_windows(m128, k) & range;
    const uint64_t em_lo = (uint64_t)em, em_hi = (uint64_t)(em >> 64);
#pragma unroll
    for (int w = 0; w < MW + 2; w++) {
        const uint32_t cw = words[w];
        if (w * 16 >= total) break;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int b = w * 16 + j;
            const uint32_t c = (cw >> (2 * j)) & 3u;
            fwd = ((fwd << 2) | c) & kp.kmask;
            rc = (rc >> 2) | ((KeyT)(3u - c) << kp.rcshift);
            if (((b < 64 ? em_lo >> b : em_hi >> (b - 64)) & 1ULL)) emit(s0 + b - (k - 1), fwd, rc);
        }
    }
}

// Same walk, but step(start, fwd, rc, valid_k, valid_k1) is called for EVERY start of the unit:
// valid_k = the k-mer is valid, valid_k1 = the (k-1)-mer ending at the same base is valid (the
// mapping kernel tests the (k-1)-mer two neighbouring k-mers share).
template <int UNIT, typename KeyT, typename KP, typename F>
__device__ __forceinline__ void sp_scan_unit_all(const uint32_t *__restrict__ pk,
                                                 const uint32_t *__restrict__ nm, int64_t s0,
                                                 const KP &kp, F &&step) {
    constexpr int MW = UNIT / 16;
    const int64_t w0 = s0 >> 4;
    uint32_t words[MW + 2];
    if (MW == 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(pk + w0);
        words[0] = v.x; words[1] = v.y; words[2] = v.z; words[3] = v.w;
    } else {
        const uint2 v = *reinterpret_cast<const uint2 *>(pk + w0);
        words[0] = v.x; words[1] = v.y;
    }
    words[MW] = pk[w0 + MW];
    words[MW + 1] = pk[w0 + MW + 1];
    const uint64_t mlo = (uint64_t)nm[s0 >> 5] | ((uint64_t)nm[(s0 >> 5) + 1] << 32);
    const uint32_t mhi = (MW == 4) ? nm[(s0 >> 5) + 2] : 0u;
    KeyT fwd = 0, rc = 0;
    const int k = kp.k;
    const int total = UNIT + k - 1;
    const unsigned __int128 m128 = (unsigned __int128)mlo | ((unsigned __int128)mhi << 64);
    const unsigned __int128 ok_k = ~sp_bad_windows(m128, k), ok_k1 = ~sp_bad_windows(m128, k - 1);
    const uint64_t k_lo = (uint64_t)ok_k, k_hi = (uint64_t)(ok_k >> 64);
    const uint64_t k1_lo = (uint64_t)ok_k1, k1_hi = (uint64_t)(ok_k1 >> 64);
#pragma unroll
    for (int w = 0; w < MW + 2; w++) {
        const uint32_t cw = words[w];
        if (w * 16 >= total) break;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int b = w * 16 + j;
            const uint32_t c = (cw >> (2 * j)) & 3u;
            fwd = ((fwd << 2) | c) & kp.kmask;
            rc = (rc >> 2) | ((KeyT)(3u - c) << kp.rcshift);
            if (b >= k - 1 && b < total)
                step(s0 + b - (k - 1), fwd, rc, (bool)((b < 64 ? k_lo >> b : k_hi >> (b - 64)) & 1ULL),
                     (bool)((b < 64 ? k1_lo >> b : k1_hi >> (b - 64)) & 1ULL));
        }
    }
}

// 32-bit keys (k <= 16: the dense-table kernels) and 64-bit keys (k <= 32: the sparse engine)
template <int UNIT, typename F>
__device__ __forceinline__ void sp_scan_unit32(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm,
                                               int64_t s0, const sp_kparams32 &kp, F &&emit) {
    sp_scan_unit_t<UNIT, uint32_t>(pk, nm, s0, kp, emit);
}
template <int UNIT, typename F>
__device__ __forceinline__ void sp_scan_unit64(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm,
                                               int64_t s0, const sp_kparams &kp, F &&emit) {
    sp_scan_unit_t<UNIT, uint64_t>(pk, nm, s0, kp, emit);
}

// ------------------------------------------------------------------ direct-window scan (k <= 15/16)
// The rolling scan above spends ~40 VALU instructions per k-mer (two shift/or chains per base, the
// run mask test and an exec-mask branch per emitted k-mer).  The dense-table kernels are bound by
// exactly that, so they read the genome from TWO packed streams instead:
//   pk  LSB-first  (base i at bits 2*(i%16))      -- also what the 64-bit engines use
//   pm  MSB-first  (base i at bits 30 - 2*(i%16))
// A 16-base window starting at ANY base is one v_alignbit_b32 per stream.  From the MSB-first window V
// the forward k-mer in key order (first base most significant) is V >> (32 - 2k); from the LSB-first
// window W its reverse complement in key order is ~W & kmask (complementing reverses nothing: the last
// base of the k-mer already sits in the most significant used bits).  ~6 instructions for both strands.
#define SP_UNIT32 32

// bit j of the result: the w bases s0+j .. s0+j+w-1 are all valid (w <= 32; s0 a multiple of 32).
// `shift1`: answer for windows starting one base later (s0+j+1 ..), used for the shared (k-1)-mer.
__device__ __forceinline__ uint64_t sp_bad_from_words64(uint64_t inv, int w) {
    if (w <= 0) return 0;
    uint64_t e = inv;
    int have = 1;
    while (have * 2 <= w) {
        e |= e >> have;
        have *= 2;
    }
    if (have < w) e |= e >> (w - have);
    return e;   // bit j: an invalid base in [s0+j, s0+j+w)
}
__device__ __forceinline__ uint64_t sp_bad_starts64(const uint32_t *__restrict__ nm, int64_t s0, int w) {
    const uint64_t inv = (uint64_t)nm[s0 >> 5] | ((uint64_t)nm[(s0 >> 5) + 1] << 32);
    if (w <= 0) return 0;
    uint64_t e = inv;
    int have = 1;
    while (have * 2 <= w) {
        e |= e >> have;
        have *= 2;
    }
    if (have < w) e |= e >> (w - have);
    return e;   // bit j: an invalid base in [s0+j, s0+j+w)
}

// Round 5: the MSB-first stream is no longer read (nor written): the MSB-first twin of a 16-base word is its bit
// reversal with the two bits of every code swapped back -- four VALU operations per word, i.e. 12 per 32 k-mers in the
// kernels that scan, against a second 0.25-B/base stream written by k0_pack and read by every scan (SP_DERIVE_PM=0
// restores the loads; the kernels keep their `pm` argument)
__device__ __forceinline__ uint32_t sp_msb_of_lsb(uint32_t w) {
    const uint32_t r = __builtin_bitreverse32(w);
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}
struct sp_words32 {
    uint32_t l[3], m[3];
};
__device__ __forceinline__ sp_words32 sp_load_words32(const uint32_t *__restrict__ pk,
                                                      const uint32_t *__restrict__ pm, int64_t s0) {
    const int64_t w0 = s0 >> 4;   // even: 8-byte aligned
    sp_words32 r;
    const uint2 a = *reinterpret_cast<const uint2 *>(pk + w0);
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = pk[w0 + 2];
#if SP_DERIVE_PM
    (void)pm;
    r.m[0] = sp_msb_of_lsb(r.l[0]); r.m[1] = sp_msb_of_lsb(r.l[1]); r.m[2] = sp_msb_of_lsb(r.l[2]);
#else
    const uint2 b = *reinterpret_cast<const uint2 *>(pm + w0);
    r.m[0] = b.x; r.m[1] = b.y; r.m[2] = pm[w0 + 2];
#endif
    return r;
}
// 16-base windows starting at base s0 + J (J a compile-time constant after unrolling)
template <int J>
__device__ __forceinline__ uint32_t sp_win_lsb(const sp_words32 &x) {
    constexpr int q = J >> 4, r = J & 15;
    if (r == 0) return x.l[q];
    return __builtin_amdgcn_alignbit(x.l[q + 1], x.l[q], 2 * r);
}
template <int J>
__device__ __forceinline__ uint32_t sp_win_msb(const sp_words32 &x) {
    constexpr int q = J >> 4, r = J & 15;
    if (r == 0) return x.m[q];
    return __builtin_amdgcn_alignbit(x.m[q], x.m[q + 1], 32 - 2 * r);
}

// the same for an index that is constant only after loop unrolling
__device__ __forceinline__ uint32_t sp_win_lsb_at(const sp_words32 &x, int j) {
    const int q = j >> 4, r = j & 15;
    return r == 0 ? x.l[q] : __builtin_amdgcn_alignbit(x.l[q + 1], x.l[q], 2 * r);
}
__device__ __forceinline__ uint32_t sp_win_msb_at(const sp_words32 &x, int j) {
    const int q = j >> 4, r = j & 15;
    return r == 0 ? x.m[q] : __builtin_amdgcn_alignbit(x.m[q], x.m[q + 1], 32 - 2 * r);
}

template <int J, int STEP, typename F>
struct sp_win_loop {
    static __device__ __forceinline__ void run(const sp_words32 &x, F &f) {
        f(J, sp_win_msb<J>(x), sp_win_lsb<J>(x));
        sp_win_loop<J + STEP, STEP, F>::run(x, f);
    }
};
template <int STEP, typename F>
struct sp_win_loop<32, STEP, F> {
    static __device__ __forceinline__ void run(const sp_words32 &, F &) {}
};

// emit(slot) for every valid k-mer start of the unit [s0, s0+32); k <= 15 (dense-table kernels)
template <bool ODD, typename F>
__device__ __forceinline__ void sp_scan32_slots(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm,
                                                const uint32_t *__restrict__ nm, int64_t s0,
                                                const sp_kparams32 &kp, F &&emit) {
    const uint32_t ok = ~(uint32_t)sp_bad_starts64(nm, s0, kp.k);
    const sp_words32 x = sp_load_words32(pk, pm, s0);
    const int sh = 32 - 2 * kp.k;
    if (__all(ok == 0xffffffffu)) {   // wave-uniform: no N / chromosome end in any lane's unit
        auto f = [&](int, uint32_t V, uint32_t W) { emit(sp_slot_of32_t<ODD>(V >> sh, ~W & kp.kmask, kp)); };
        sp_win_loop<0, 1, decltype(f)>::run(x, f);
    } else {
        auto f = [&](int j, uint32_t V, uint32_t W) {
            if ((ok >> j) & 1u) emit(sp_slot_of32_t<ODD>(V >> sh, ~W & kp.kmask, kp));
        };
        sp_win_loop<0, 1, decltype(f)>::run(x, f);
    }
}

// ------------------------------------------------------------------ direct-window scan, 64-bit keys (k <= 32)
// The same two-stream trick for the sparse engines: a 32-base window starting at ANY base of a 32-start unit is two
// v_alignbit_b32 per stream (bases s0+j .. s0+j+31 live in words 0..3 of the unit's streams); the forward k-mer in
// key order is the MSB-first window >> (64 - 2k), its reverse complement ~(LSB-first window) & kmask.  The rolling
// scan these kernels used costs ~55 VALU instructions per k-mer in 64-bit arithmetic, this ~20.
struct sp_words64 {
    uint32_t l[4], m[4];
};
__device__ __forceinline__ sp_words64 sp_load_words64(const uint32_t *__restrict__ pk,
                                                      const uint32_t *__restrict__ pm, int64_t s0) {
    const int64_t w0 = s0 >> 4;   // even: 8-byte aligned
    sp_words64 r;
    const uint2 a = *reinterpret_cast<const uint2 *>(pk + w0), a2 = *reinterpret_cast<const uint2 *>(pk + w0 + 2);
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a2.x; r.l[3] = a2.y;
#if SP_DERIVE_PM
    (void)pm;
#pragma unroll
    for (int i = 0; i < 4; i++) r.m[i] = sp_msb_of_lsb(r.l[i]);
#else
    const uint2 b = *reinterpret_cast<const uint2 *>(pm + w0), b2 = *reinterpret_cast<const uint2 *>(pm + w0 + 2);
    r.m[0] = b.x; r.m[1] = b.y; r.m[2] = b2.x; r.m[3] = b2.y;
#endif
    return r;
}
template <int J>
__device__ __forceinline__ uint32_t sp_w64_lsb16(const sp_words64 &x) {   // 16 bases from s0 + J, LSB-first
    constexpr int q = J >> 4, r = J & 15;
    if (r == 0) return x.l[q];
    return __builtin_amdgcn_alignbit(x.l[q + 1], x.l[q], 2 * r);
}
template <int J>
__device__ __forceinline__ uint32_t sp_w64_msb16(const sp_words64 &x) {   // 16 bases from s0 + J, MSB-first
    constexpr int q = J >> 4, r = J & 15;
    if (r == 0) return x.m[q];
    return __builtin_amdgcn_alignbit(x.m[q], x.m[q + 1], 32 - 2 * r);
}
template <int J, typename F>
struct sp_win64_loop {
    static __device__ __forceinline__ void run(const sp_words64 &x, F &f) {
        // J = 31, r = 15: the second half-window starts at base 47 and ends at base 62 -> words 2 and 3
        const uint64_t V = ((uint64_t)sp_w64_msb16<J>(x) << 32) | sp_w64_msb16<J + 16>(x);
        const uint64_t W = (uint64_t)sp_w64_lsb16<J>(x) | ((uint64_t)sp_w64_lsb16<J + 16>(x) << 32);
        f(J, V, W);
        sp_win64_loop<J + 1, F>::run(x, f);
    }
};
template <typename F>
struct sp_win64_loop<32, F> {
    static __device__ __forceinline__ void run(const sp_words64 &, F &) {}
};
// J + 16 = 32 .. 47 with r == 0 only at J = 16 (q = 2): every index stays within words 0..3

// step(j, fwd, rc) for EVERY start s0 + j of the 32-start unit (validity is the caller's: sp_bad_starts64)
template <typename F>
__device__ __forceinline__ void sp_scan32_keys64(const sp_words64 &x, const sp_kparams &kp, F &&step) {
    const int sh = 64 - 2 * kp.k;
    auto f = [&](int j, uint64_t V, uint64_t W) { step(j, V >> sh, ~W & kp.kmask); };
    sp_win64_loop<0, decltype(f)>::run(x, f);
}
// emit(start, fwd, rc) for every VALID k-mer start of the unit [s0, s0 + 32), s0 a multiple of 32; 16 <= k <= 32
template <typename F>
__device__ __forceinline__ void sp_scan32_valid64(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm,
                                                  const uint32_t *__restrict__ nm, int64_t s0, const sp_kparams &kp,
                                                  F &&emit) {
    const uint32_t ok = ~(uint32_t)sp_bad_starts64(nm, s0, kp.k);
    if (__all(ok == 0u)) return;
    const sp_words64 x = sp_load_words64(pk, pm, s0);
    if (__all(ok == 0xffffffffu)) {
        sp_scan32_keys64(x, kp, [&](int j, uint64_t fwd, uint64_t rc) { emit(s0 + j, fwd, rc); });
    } else {
        sp_scan32_keys64(x, kp, [&](int j, uint64_t fwd, uint64_t rc) {
            if ((ok >> j) & 1u) emit(s0 + j, fwd, rc);
        });
    }
}

// ------------------------------------------------------------------ byte tables
// raw count of `slot` (absolute) given its table byte: bytes below 255 are exact, 255 defers to the
// chromosome's overflow list (ascending slots; binary search -- counts >= 255 are rare)
__device__ __forceinline__ uint32_t sp_ovf_lookup(const uint2 *__restrict__ ovf, int64_t n, uint32_t slot) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ovf[mid].x < slot) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && ovf[lo].x == slot) ? ovf[lo].y : 255u;
}
// the same through the per-bucket index when the list has one: ~4 steps inside the bucket instead of ~17 over the list
#define SP_OVF_SHIFT_ 15
__device__ __forceinline__ uint32_t sp_ovf_lookup_t(const sp_tabref &t, uint32_t slot) {
    if (!t.ovf_idx) return sp_ovf_lookup(t.ovf, t.n_ovf, slot);
    const uint32_t b = slot >> SP_OVF_SHIFT_;
    uint32_t lo = t.ovf_idx[b], hi = t.ovf_idx[b + 1];
    const uint32_t end = hi;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (t.ovf[mid].x < slot) lo = mid + 1;
        else hi = mid;
    }
    return (lo < end && t.ovf[lo].x == slot) ? t.ovf[lo].y : 255u;
}
__device__ __forceinline__ uint32_t sp_tab_count(const sp_tabref &t, int64_t local, int64_t slot_base) {
    const uint32_t b = t.tab[local];
    return b < 255u ? b : sp_ovf_lookup_t(t, (uint32_t)(slot_base + local));
}
// overflow lists are produced bucket by bucket (2^15 consecutive slots), see sp_count2.hip / sp_count.hip
#define SP_OVF_SHIFT 15
static_assert(SP_OVF_SHIFT == SP_OVF_SHIFT_, "bucket width of the overflow index");

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: it
// also drains every global load in flight, so a software pipeline that prefetches the next tile's keys across a
// barrier pays the full memory round trip at that barrier anyway (the list counter of engine 3 spent 2 us per bucket
// there).  Use where the threads of a block exchange data through LDS only; the compiler still waits (vmcnt) for a
// prefetched register before its first use.
__device__ __forceinline__ void sp_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// wave-level inclusive/exclusive helpers (64 lanes)
__device__ __forceinline__ int sp_lane() { return threadIdx.x & 63; }

// Exclusive prefix count of `pred` over the 256..1024-thread block; returns
// this thread's offset and the block total.  lds must hold >= 16 ints.
__device__ __forceinline__ uint32_t sp_block_excl_count(bool pred, uint32_t *lds, uint32_t &total) {
    const unsigned long long bal = __ballot(pred);
    const int lane = sp_lane();
    const int wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    uint32_t in_wave = __popcll(bal & ((1ULL << lane) - 1ULL));
    if (lane == 0) lds[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < nw; w++) {
        uint32_t v = lds[w];
        if (w < wave) base += v;
        tot += v;
    }
    __syncthreads();
    total = tot;
    return base + in_wave;
}

// Exclusive prefix sum of one value per thread over the block (<= 1024 threads); lds: >= 16 elements.  Wave scans by
// shuffles, the (<= 16) wave totals through LDS -- a single thread walking 1024 partial sums in LDS took 20-25 us in
// the one-block offset kernels that run once per chromosome.
template <typename T>
__device__ __forceinline__ T sp_block_excl_scan(T v, T *lds, T &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    T incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < nw; w++) {
        const T x = lds[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

__device__ __forceinline__ unsigned long long sp_block_sum_u64(unsigned long long v,
                                                               unsigned long long *lds) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = sp_lane(), wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    unsigned long long t = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; w++) t += lds[w];
    __syncthreads();
    return t;  // valid on thread 0
}
