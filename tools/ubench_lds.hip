// LDS micro-benchmarks behind the counting engine's design (MI355X / gfx950).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_lds tools/ubench_lds.hip && tools/ubench_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: ds_add (no return)  1: ds_add_rtn  2: plain write  3: plain read  4: private RMW (read+add+write, thread-own column)
template <int MODE>
__global__ void __launch_bounds__(1024) k_lds(int log2bins, int iters, uint32_t *out) {
    extern __shared__ uint32_t sm[];
    const uint32_t nb = 1u << log2bins;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    uint32_t acc = 0, x = blockIdx.x * 1024u + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t b = (x >> 7) & (nb - 1);
            if (MODE == 0) atomicAdd(&sm[b], 1u);
            if (MODE == 1) acc += atomicAdd(&sm[b], 1u);
            if (MODE == 2) sm[b] = x;
            if (MODE == 3) acc += sm[b];
            if (MODE == 4) { const uint32_t a = (b & ~1023u & (nb - 1)) | threadIdx.x; uint32_t v = sm[a & (nb - 1)]; sm[a & (nb - 1)] = v + 1; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc + sm[5];
}

// wave-level multisplit on NB bits: rank of every key among the lanes of its wave that hold the same bucket,
// from NB ballots; one LDS add per DISTINCT bucket per wave.  Returns ranks via acc to keep them live.
template <int NB>
__global__ void __launch_bounds__(1024) k_ballot(int iters, uint32_t *out) {
    __shared__ uint32_t cnt[1 << NB];
    for (uint32_t i = threadIdx.x; i < (1u << NB); i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    uint32_t acc = 0, x = blockIdx.x * 1024u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t b = (x >> 7) & ((1u << NB) - 1);
            unsigned long long m = ~0ULL;
#pragma unroll
            for (int bit = 0; bit < NB; bit++) {
                const unsigned long long bal = __ballot((b >> bit) & 1u);
                m &= ((b >> bit) & 1u) ? bal : ~bal;
            }
            const uint32_t rank = __popcll(m & ((1ULL << lane) - 1ULL));
            uint32_t base = 0;
            if (rank == 0) base = atomicAdd(&cnt[b], (uint32_t)__popcll(m));   // leader of its group
            base = __shfl(base, __ffsll((long long)m) - 1, 64);
            acc += base + rank;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc + cnt[3];
}

template <typename K>
static double run(K kern, int grid, size_t shm, int iters, uint32_t *d_out, const char *name, int log2bins, bool two_args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (two_args) hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), shm, 0, log2bins, iters, d_out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)grid * 1024.0 * iters * 8.0;
    printf("%-44s bins 2^%-2d : %.3f ms  %.1f G/s\n", name, log2bins, ms, ops / ms / 1e6);
    return ms;
}

int main() {
    uint32_t *d_out; CK(hipMalloc(&d_out, 1 << 20));
    const int grid = 256, iters = 2048;
    hipFuncSetAttribute((const void *)k_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k_lds<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int lb : {7, 10, 13, 15}) {
        run(k_lds<0>, grid, 4u << lb, iters, d_out, "ds_add_u32 (no return), random bins", lb, true);
        run(k_lds<1>, grid, 4u << lb, iters, d_out, "ds_add_rtn_u32, random bins", lb, true);
        run(k_lds<2>, grid, 4u << lb, iters, d_out, "ds_write_b32, random addresses", lb, true);
        run(k_lds<3>, grid, 4u << lb, iters, d_out, "ds_read_b32, random addresses", lb, true);
    }
    run(k_lds<4>, grid, 4u << 15, iters, d_out, "private read+add+write (own column)", 15, true);
    for (int g : {256, 512, 1024})   // more than one block per CU
        run(k_lds<0>, g, 4u << 13, iters, d_out, g == 256 ? "ds_add_u32 no-rtn, 1 block/CU" : g == 512 ? "ds_add_u32 no-rtn, 2 blocks/CU" : "ds_add_u32 no-rtn, 4 blocks/CU", 13, true);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); hipLaunchKernelGGL(k_ballot<7>, dim3(grid), dim3(1024), 0, 0, iters, d_out); hipEventRecord(e1); hipEventSynchronize(e1); }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s bins 2^7  : %.3f ms  %.1f G/s\n", "wave multisplit (7 ballots) + 1 add / group", ms, (double)grid * 1024.0 * iters * 8.0 / ms / 1e6);
        for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); hipLaunchKernelGGL(k_ballot<6>, dim3(grid), dim3(1024), 0, 0, iters, d_out); hipEventRecord(e1); hipEventSynchronize(e1); }
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s bins 2^6  : %.3f ms  %.1f G/s\n", "wave multisplit (6 ballots) + 1 add / group", ms, (double)grid * 1024.0 * iters * 8.0 / ms / 1e6);
    }
    return 0;
}
