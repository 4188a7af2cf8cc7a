// ubench_gather.hip -- what a locality-addressed pair filter would buy the map stage (round 6, review item 2).
// k5_map2 issues ONE 4-byte filter gather per PAIR of starts (7.03 G per wheat-like pass, 21.4 of its 40.7 ms) and one
// 16-byte bucket load per candidate QUAD (0.96 G, 12.7 ms).  A filter whose block is addressed by the (k-3)-mer the two
// pairs of a quad share answers both pairs with ONE gather of 8 (or 16) bytes -- but every (k-1)-mer is then entered
// under two cores, so the filter doubles (2 -> 4 MiB: the size of one XCD's L2) for the same false-positive rate.
// This benchmark prices exactly that trade with the access pattern of the kernel's inner loop:
//   per "quad" and lane:  G gathers of W bytes from a filter of F MiB (random words), then, with probability p, one
//   16-byte load from a 128-MiB table (random buckets) -- p = the candidate-quad rate of the wheat-like pass (0.27).
// 768-thread workgroups, 16 per CU, 16 quads per lane and unit as in k5_map2; 3.5 G quads per run.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    return x;
}

// G gathers of W bytes per quad (W = 4: uint32, 8: uint2, 16: uint4), QI quads in flight per lane; NT: the table's bucket loads
// carry the non-temporal hint (streaming lines are the first the L2 gives up: does a 4-MiB filter then stay resident?)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int W, int G, int QI, int NT = 0, int PRED = 256 /* a lane issues each of its gathers with probability PRED / 256 */>
__global__ void __launch_bounds__(768)
k_quads(const uint32_t *__restrict__ filt, uint32_t fmask /* W-byte blocks - 1 */, const uint4 *__restrict__ tab, uint32_t tmask,
        uint32_t p16 /* candidate rate x 65536 */, long long n_units, unsigned long long *__restrict__ sink) {
    uint32_t acc = 0;
    for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += (long long)gridDim.x * blockDim.x) {
        uint32_t s = mix((uint32_t)u * 2654435761u + 12345u);
#pragma unroll 1
        for (int q0 = 0; q0 < 16; q0 += QI) {
            uint32_t w[QI][G][W / 4];
            uint32_t h[QI];
#pragma unroll
            for (int q = 0; q < QI; q++) {
                h[q] = s = mix(s + 0x632BE5ABu);
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const uint32_t idx = (uint32_t)(((unsigned long long)mix(h[q] + 77u * g) * (unsigned long long)(fmask + 1u)) >> 32);
                    if (PRED < 256) {
                        w[q][g][0] = 0;
                        if (((mix(h[q] ^ (0x51ED27u * (g + 1))) >> 11) & 255u) < (uint32_t)PRED) w[q][g][0] = filt[idx];
                        continue;
                    }
                    if (W == 4) w[q][g][0] = filt[idx];
                    else if (W == 8) {
                        const uint2 v = reinterpret_cast<const uint2 *>(filt)[idx];
                        w[q][g][0] = v.x; w[q][g][1] = v.y;
                    } else {
                        const uint4 v = reinterpret_cast<const uint4 *>(filt)[idx];
                        w[q][g][0] = v.x; w[q][g][1] = v.y; w[q][g][2] = v.z; w[q][g][3] = v.w;
                    }
                }
            }
            uint4 B[QI];
#pragma unroll
            for (int q = 0; q < QI; q++) {
                uint32_t x = 0;
#pragma unroll
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int i = 0; i < W / 4; i++) x ^= w[q][g][i];
                // candidate: decided by the hash (rate p), but only once the filter words are here (the dependency of the real loop)
                const bool cand = ((h[q] >> 8) & 0xFFFFu) < p16 + (x == 0xDEADBEEFu ? 1u : 0u);
                B[q] = make_uint4(0, 0, 0, 0);
                if (cand) {
                    if (NT) {
                        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(tab) + (mix(h[q] ^ 0x5bd1e995u) & tmask));
                        B[q] = make_uint4(t.x, t.y, t.z, t.w);
                    } else {
                        B[q] = tab[mix(h[q] ^ 0x5bd1e995u) & tmask];
                    }
                }
                acc += x;
            }
#pragma unroll
            for (int q = 0; q < QI; q++) acc += B[q].x ^ B[q].y ^ B[q].z ^ B[q].w;
        }
    }
    if (acc == 0x1234567u) atomicAdd(sink, 1ULL);
}

template <int W, int G, int QI, int NT = 0, int PRED = 256>
static void run(const char *what, const uint32_t *filt, size_t fbytes, const uint4 *tab, size_t tbytes, double p, long long n_units,
                unsigned long long *sink, int cus) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int t = 0; t < 3; t++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_quads<W, G, QI, NT, PRED>), dim3(cus * 16), dim3(768), 0, 0, filt, (uint32_t)(fbytes / W - 1), tab, (uint32_t)(tbytes / 16 - 1),
                           (uint32_t)(p * 65536.0), n_units, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double quads = (double)n_units * 16.0;
    printf("%-34s filter %5.1f MiB  p = %.2f  QI = %d: %7.2f ms  (%.0f G gathers/s, %.0f G quads/s)\n", what, fbytes / 1048576.0, p, QI, best,
           quads * G / best / 1e6, quads / best / 1e6);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t fmax = 32ull << 20, tbytes = 128ull << 20;
    uint32_t *filt;
    uint4 *tab;
    unsigned long long *sink;
    CK(hipMalloc(&filt, fmax));
    CK(hipMalloc(&tab, tbytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(filt, 0x5a, fmax));
    CK(hipMemset(tab, 0x3c, tbytes));
    const long long n_units = 14070000000LL / 64;      // the wheat-like genome: 3.5 G quads
    for (double p : {0.0, 0.27, 0.33}) {
        for (size_t mb : {1, 2, 3, 4, 8, 16}) {
            const size_t fb = mb << 20;
            if (mb <= 4 && mb != 3) run<4, 2, 1>("2 x 4 B per quad (today)", filt, fb, tab, tbytes, p, n_units, sink, cus);
            run<8, 1, 1>("1 x 8 B per quad", filt, fb, tab, tbytes, p, n_units, sink, cus);
            if (mb >= 2) run<16, 1, 1>("1 x 16 B per quad", filt, fb, tab, tbytes, p, n_units, sink, cus);
        }
        run<8, 1, 2>("1 x 8 B per quad", filt, 4u << 20, tab, tbytes, p, n_units, sink, cus);
        run<4, 2, 2>("2 x 4 B per quad (today)", filt, 2u << 20, tab, tbytes, p, n_units, sink, cus);
    }
    printf("-- is a gather priced per ACTIVE lane?  2 x 4 B per quad, each issued by a lane with probability 2/3, 1/2 (2-MiB filter)\n");
    for (double p : {0.0, 0.27}) {
        run<4, 2, 1, 0, 171>("2 x 4 B, 2/3 of the lanes each", filt, 2u << 20, tab, tbytes, p, n_units, sink, cus);
        run<4, 2, 1, 0, 128>("2 x 4 B, 1/2 of the lanes each", filt, 2u << 20, tab, tbytes, p, n_units, sink, cus);
    }
    printf("-- the table's bucket loads non-temporal\n");
    for (double p : {0.27, 0.33}) {
        run<4, 2, 1, 1>("2 x 4 B per quad (today), NT table", filt, 2u << 20, tab, tbytes, p, n_units, sink, cus);
        for (size_t mb : {2, 3, 4, 6, 8}) run<8, 1, 1, 1>("1 x 8 B per quad, NT table", filt, mb << 20, tab, tbytes, p, n_units, sink, cus);
    }
    // one gather per SIX starts (a (k-5)-mer core, 16-byte blocks): 2/3 of the quads' count of gathers, three insertions per (k-1)-mer
    printf("-- one 16-byte gather per six starts: the same kernel over 2/3 of the units\n");
    for (size_t mb : {4, 8, 16}) run<16, 1, 1>("1 x 16 B per 6 starts", filt, mb << 20, tab, tbytes, 0.27 * 1.5, n_units * 2 / 3, sink, cus);
    return 0;
}
