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

template <typename F>
__device__ __forceinline__ void sp_scan_unit(const uint32_t *__restrict__ pk,
                                             const uint32_t *__restrict__ nm, int64_t s0,
                                             const sp_kparams &kp, F &&emit) {
    const int64_t w0 = s0 >> 4;
    const uint4 main4 = *reinterpret_cast<const uint4 *>(pk + w0);
    const uint32_t h0 = pk[w0 + 4], h1 = pk[w0 + 5];
    const uint32_t m0 = nm[(s0 >> 5)], m1 = nm[(s0 >> 5) + 1], m2 = nm[(s0 >> 5) + 2];
    const uint64_t mlo = (uint64_t)m0 | ((uint64_t)m1 << 32);
    uint32_t words[6] = {main4.x, main4.y, main4.z, main4.w, h0, h1};
    uint64_t fwd = 0, rc = 0;
    int run = 0;
    const int k = kp.k;
    const int total = SP_UNIT + k - 1;  // bases to consume
#pragma unroll
    for (int w = 0; w < 6; w++) {
        uint32_t cw = words[w];
        uint32_t mw = (w < 4) ? (uint32_t)((mlo >> (16 * w)) & 0xffffu)
                              : (uint32_t)((m2 >> (16 * (w - 4))) & 0xffffu);
        if (w * 16 >= total) break;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int b = w * 16 + j;
            uint32_t c = (cw >> (2 * j)) & 3u;
            fwd = ((fwd << 2) | c) & kp.kmask;
            rc = (rc >> 2) | ((uint64_t)(3u - c) << kp.rcshift);
            run = ((mw >> j) & 1u) ? 0 : run + 1;
            if (run >= k && b < total) emit(s0 + b - (k - 1), fwd, rc);
        }
    }
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
