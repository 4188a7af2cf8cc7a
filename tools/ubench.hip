// tools/ubench.hip -- micro-benchmarks that drive the K1/K5 design decisions on MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256) k_atomic(uint32_t *tab, uint64_t mask, int per_thread) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; i++) {
        uint64_t idx = mix(t * 1315423911ULL + i) & mask;
        atomicAdd(&tab[idx], 1u);
    }
}

__global__ void __launch_bounds__(256) k_gather8(const uint8_t *tab, uint64_t mask, int per_thread, uint32_t *out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; i++) {
        uint64_t idx = mix(t * 1315423911ULL + i) & mask;
        acc += tab[idx];
    }
    if (acc == 0xffffffffu) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_gatherbit(const uint32_t *tab, uint64_t mask_bits, int per_thread, uint32_t *out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; i++) {
        uint64_t idx = mix(t * 1315423911ULL + i) & mask_bits;
        acc += (tab[idx >> 5] >> (idx & 31)) & 1u;
    }
    if (acc == 0xffffffffu) out[0] = acc;
}

// grouped gather: G adjacent lanes read (different words of) the same 64-B line
__global__ void __launch_bounds__(256) k_gather_grouped(const uint32_t *tab, uint64_t mask_lines, int G, int per_thread, uint32_t *out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; i++) {
        uint64_t line = mix((t / G) * 1315423911ULL + i) & mask_lines;
        acc += tab[line * 16 + (t % 16)];
    }
    if (acc == 0xffffffffu) out[0] = acc;
}

// LDS histogram: random increments into `bins` u32 counters, then flush (store) to global
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_ldshist(uint32_t *out, int bins, int per_thread) {
    extern __shared__ uint32_t h[];
    for (int i = threadIdx.x; i < bins; i += THREADS) h[i] = 0;
    __syncthreads();
    uint64_t t = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
    for (int i = 0; i < per_thread; i++) {
        uint32_t idx = (uint32_t)(mix(t * 1315423911ULL + i) % (uint32_t)bins);
        atomicAdd(&h[idx], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += THREADS) out[(size_t)blockIdx.x * bins + i] = h[i];
}

// scatter write: each thread writes u32 keys into one of F buckets through per-block LDS staging
// (counting-sort of a tile, then coalesced copy-out).  Measures the partition pass.
template <int THREADS, int PER_THREAD>
__global__ void __launch_bounds__(THREADS)
k_partition(uint32_t *out, unsigned long long *cursors, int log2F, int key_bits, size_t cap_per_bucket, int tiles_per_block) {
    extern __shared__ uint32_t sm[];
    const int F = 1 << log2F;
    uint32_t *hist = sm;                 // F
    uint32_t *start = sm + F;            // F
    uint32_t *gbase_lo = sm + 2 * F;     // F  (global offset within bucket, low 32 bits are enough per tile)
    uint32_t *keys = sm + 3 * F;         // THREADS*PER_THREAD
    unsigned long long *gb = (unsigned long long *)(sm + 3 * F + THREADS * PER_THREAD);  // F
    const int T = THREADS * PER_THREAD;
    for (int tile = 0; tile < tiles_per_block; tile++) {
        for (int i = threadIdx.x; i < F; i += THREADS) hist[i] = 0;
        __syncthreads();
        uint32_t my[PER_THREAD];
        uint64_t t = ((uint64_t)blockIdx.x * tiles_per_block + tile) * T + (uint64_t)threadIdx.x * PER_THREAD;
#pragma unroll
        for (int j = 0; j < PER_THREAD; j++) {
            my[j] = (uint32_t)(mix(t + j) & ((1ULL << key_bits) - 1));
            atomicAdd(&hist[my[j] >> (key_bits - log2F)], 1u);
        }
        __syncthreads();
        // exclusive scan of hist (single wave, F <= 4096)
        if (threadIdx.x < 64) {
            uint32_t run = 0;
            for (int base = 0; base < F; base += 64) {
                uint32_t v = hist[base + threadIdx.x];
                uint32_t incl = v;
                for (int o = 1; o < 64; o <<= 1) {
                    uint32_t n = __shfl_up(incl, o, 64);
                    if ((int)threadIdx.x >= o) incl += n;
                }
                start[base + threadIdx.x] = run + incl - v;
                run += __shfl(incl, 63, 64);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < F; i += THREADS) {
            uint32_t c = hist[i];
            gb[i] = c ? atomicAdd(&cursors[i], (unsigned long long)c) : 0ULL;
            hist[i] = 0;  // reuse as running local cursor
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER_THREAD; j++) {
            uint32_t b = my[j] >> (key_bits - log2F);
            uint32_t pos = start[b] + atomicAdd(&hist[b], 1u);
            keys[pos] = my[j];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += THREADS) {
            uint32_t kx = keys[i];
            uint32_t b = kx >> (key_bits - log2F);
            size_t dst = (size_t)b * cap_per_bucket + (size_t)(gb[b] + (i - start[b]));
            out[dst] = kx;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_copy(const uint4 *a, uint4 *b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}

static double timeit(hipEvent_t e0, hipEvent_t e1) {
    float ms;
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d\n", prop.name, prop.multiProcessorCount);
    const size_t GiB = 1ULL << 30;
    uint8_t *buf;
    CK(hipMalloc(&buf, 8 * GiB));
    CK(hipMemset(buf, 0, 8 * GiB));
    uint32_t *out;
    CK(hipMalloc(&out, 512 << 20));
    const int blocks = 256 * 16, threads = 256;
    // streaming copy
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        k_copy<<<256 * 8, 256>>>((const uint4 *)buf, (uint4 *)(buf + 4 * GiB), 4 * GiB / 16);
        CK(hipEventRecord(e1));
        double ms = timeit(e0, e1);
        printf("copy 4GiB->4GiB: %.3f ms  %.1f GB/s (r+w)\n", ms, 8.0 * GiB / ms / 1e6);
    }
    // random atomics
    for (unsigned long long sz : {2ULL << 30, 512ULL << 20, 128ULL << 20, 16ULL << 20, 2ULL << 20}) {
        int per = 64;
        double n = (double)blocks * threads * per;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            k_atomic<<<blocks, threads>>>((uint32_t *)buf, sz / 4 - 1, per);
            CK(hipEventRecord(e1));
            double ms = timeit(e0, e1);
            if (rep) printf("atomicAdd u32 random over %6llu MiB: %.3f ms  %.2f G/s\n", sz >> 20, ms, n / ms / 1e6);
        }
    }
    for (unsigned long long sz : {512ULL << 20, 128ULL << 20, 32ULL << 20, 4ULL << 20, 1ULL << 20}) {
        int per = 64;
        double n = (double)blocks * threads * per;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            k_gather8<<<blocks, threads>>>(buf, sz - 1, per, out);
            CK(hipEventRecord(e1));
            double ms = timeit(e0, e1);
            if (rep) printf("gather u8 random over %6llu MiB: %.3f ms  %.2f G/s\n", sz >> 20, ms, n / ms / 1e6);
        }
    }
    for (unsigned long long sz : {64ULL << 20, 16ULL << 20, 4ULL << 20, 1ULL << 20, 256ULL << 10}) {
        int per = 64;
        double n = (double)blocks * threads * per;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            k_gatherbit<<<blocks, threads>>>((const uint32_t *)buf, sz * 8 - 1, per, out);
            CK(hipEventRecord(e1));
            double ms = timeit(e0, e1);
            if (rep) printf("gather bit random over %6llu KiB bitmap: %.3f ms  %.2f G/s\n", sz >> 10, ms, n / ms / 1e6);
        }
    }
    for (unsigned long long sz : {4ULL << 20, 32ULL << 20, 512ULL << 20}) {
        for (int G : {1, 2, 4, 8, 16}) {
            int per = 64;
            double n = (double)blocks * threads * per;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                k_gather_grouped<<<blocks, threads>>>((const uint32_t *)buf, sz / 64 - 1, G, per, out);
                CK(hipEventRecord(e1));
                double ms = timeit(e0, e1);
                if (rep) printf("grouped gather (G=%2d lanes per 64-B line) over %4llu MiB: %.3f ms  %.2f G lanes/s  %.2f G lines/s\n", G, sz >> 20, ms, n / ms / 1e6, n / G / ms / 1e6);
            }
        }
    }
    // LDS histogram
    for (int bins : {8192, 32768}) {
        int per = 256;
        int nb = 256 * 4;
        double n = (double)nb * 1024 * per;
        CK(hipFuncSetAttribute((const void *)k_ldshist<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            k_ldshist<1024><<<nb, 1024, bins * 4>>>(out, bins, per);
            CK(hipEventRecord(e1));
            double ms = timeit(e0, e1);
            if (rep) printf("LDS atomicAdd random, %d bins, 1024 thr: %.3f ms  %.2f G/s (incl. flush %d KiB/block)\n", bins, ms, n / ms / 1e6, bins * 4 / 1024);
        }
    }
    // partition pass: 2^30 keys of 29 bits into F buckets
    unsigned long long *cursors;
    CK(hipMalloc(&cursors, 4096 * 8));
    for (int log2F : {6, 8, 10, 11}) {
        const int THREADS = 1024, PER = 16;
        const int T = THREADS * PER;
        const size_t nkeys = 1ULL << 29;
        const int tiles_per_block = 8;
        const int nb = (int)(nkeys / T / tiles_per_block);
        const int F = 1 << log2F;
        size_t cap = (nkeys / F) * 5 / 4 + 65536;
        if (cap * F * 4 > 7 * GiB) { printf("skip F=%d\n", F); continue; }
        size_t sh = (3 * F + T) * 4 + F * 8;
        CK(hipFuncSetAttribute((const void *)k_partition<THREADS, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(cursors, 0, 4096 * 8));
            CK(hipEventRecord(e0));
            k_partition<THREADS, PER><<<nb, THREADS, sh>>>((uint32_t *)buf, cursors, log2F, 29, cap, tiles_per_block);
            CK(hipEventRecord(e1));
            double ms = timeit(e0, e1);
            if (rep) printf("partition 2^29 keys -> %4d buckets (tile %d): %.3f ms  %.2f Gkeys/s  write %.1f GB/s\n", F, T, ms,
                            nkeys / ms / 1e6, nkeys * 4.0 / ms / 1e6);
        }
    }
    CK(hipFree(buf));
    printf("done\n");
    return 0;
}
