// ubench_xcd.hip -- bound experiment for the per-XCD split of the map stage's exact look-ups (round 4; the round-3
// review's "k5b on synthetic queues").  Today every candidate pair of k5_map gathers one 8-byte bucket of a 64-MB
// table: 1.4 G gathers beyond L2 per wheat-like pass, 18 of the kernel's 46 ms.  The proposal: route the candidates
// into queues by table slice, drain queue q only from workgroups of XCD q mod 8 so that the 2-MB slice stays in that
// XCD's 4-MiB L2.  This program measures the drain: N records of 8 bytes in 16 queues, each record gathers one 8-byte
// bucket, and a few ALU operations stand in for the hit bookkeeping.
//   mode 0  pinned: a workgroup asks HW_REG_XCC_ID which XCD it runs on and drains that XCD's two queues
//   mode 1  unpinned: queue = blockIdx mod 16 (every slice is touched from every XCD)
//   mode 2  no slices: the gathers go to the whole 64-MB table (what k5_map does today, minus the scan and the probes)
//   mode 3  stream only: the records are read, nothing is gathered
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_xcd tools/ubench_xcd.hip && tools/ubench_xcd [records in millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
#define NQ 16
#define SLICE_BITS 18                 // buckets of 8 bytes per slice: 2 MB
#define CHUNK 4096
#define THREADS 256

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDULL; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ULL; x ^= x >> 33;
    return x;
}
__global__ void k_fill(uint2 *__restrict__ a, size_t n, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(i + seed);
        a[i] = make_uint2((uint32_t)h, (uint32_t)(h >> 32));
    }
}
__global__ void __launch_bounds__(THREADS)
k_drain(const uint2 *__restrict__ queues, size_t per_queue, const uint2 *__restrict__ table, unsigned long long *__restrict__ cursor,
        int mode, unsigned long long *__restrict__ out, unsigned int *__restrict__ xcd_seen) {
    __shared__ unsigned long long s_chunk;
    const uint32_t xcd = xcc_id();
    if (threadIdx.x == 0) atomicOr(&xcd_seen[blockIdx.x & 7], 1u << xcd);
    const size_t n_chunks = (per_queue + CHUNK - 1) / CHUNK;
    unsigned long long acc = 0;
    for (int round = 0; round < NQ / 8; round++) {
        const int q = (mode == 1) ? (int)((blockIdx.x + 8u * round) % NQ) : (int)(xcd + 8u * round);
        const uint2 *Q = queues + (size_t)q * per_queue;
        const uint2 *T = table + ((size_t)q << SLICE_BITS);
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) s_chunk = atomicAdd(&cursor[q * 16], 1ULL);
            __syncthreads();
            const size_t c = s_chunk;
            if (c >= n_chunks) break;
            uint2 r[CHUNK / THREADS];
#pragma unroll
            for (int j = 0; j < CHUNK / THREADS; j++) {
                const size_t i = c * CHUNK + (size_t)j * THREADS + threadIdx.x;
                r[j] = i < per_queue ? Q[i] : make_uint2(0, 0);
            }
            if (mode == 3) {
#pragma unroll
                for (int j = 0; j < CHUNK / THREADS; j++) acc += r[j].x ^ r[j].y;
                continue;
            }
            uint2 e[CHUNK / THREADS];
#pragma unroll
            for (int j = 0; j < CHUNK / THREADS; j++)
                e[j] = (mode == 2) ? table[r[j].x & ((NQ << SLICE_BITS) - 1)] : T[r[j].x & ((1u << SLICE_BITS) - 1)];
#pragma unroll
            for (int j = 0; j < CHUNK / THREADS; j++) {       // stand-in for: tag compare, field extraction, two hits
                const uint32_t f = (e[j].x ^ r[j].y) >> 7;
                acc += ((e[j].y >> (f & 28u)) & 7u) + ((e[j].x >> ((f >> 5) & 28u)) & 7u);
            }
        }
    }
    if (acc == 0x123456789ULL) out[0] = acc;      // (keeps the work alive)
    atomicAdd(&out[1], acc & 1ULL);
}
int main(int argc, char **argv) {
    const size_t n_total = (size_t)(argc > 1 ? atof(argv[1]) : 1400.0) * 1000000ULL;
    const size_t per_queue = n_total / NQ;
    uint2 *queues, *table;
    unsigned long long *cursor, *out;
    unsigned int *seen;
    CK(hipMalloc(&queues, per_queue * NQ * 8));
    CK(hipMalloc(&table, ((size_t)NQ << SLICE_BITS) * 8));
    CK(hipMalloc(&cursor, NQ * 16 * 8));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&seen, 64));
    k_fill<<<4096, 256>>>(queues, per_queue * NQ, 1);
    k_fill<<<1024, 256>>>(table, (size_t)NQ << SLICE_BITS, 77);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[4] = {"pinned to the XCD (HW_REG_XCC_ID)", "unpinned (queue = block mod 16)", "one 32-MB table, no slices", "stream only"};
    printf("%zu M records of 8 B in %d queues (%.1f GB), slices of %d KB\n", n_total / 1000000, NQ, n_total * 8 / 1e9, (8 << SLICE_BITS) / 1024);
    for (int grid_mult = 8; grid_mult <= 32; grid_mult *= 2)
        for (int mode = 0; mode < 4; mode++) {
            float best = 1e9f;
            unsigned int hs[8];
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(cursor, 0, NQ * 16 * 8));
                CK(hipMemset(out, 0, 64));
                CK(hipMemset(seen, 0, 64));
                CK(hipEventRecord(e0));
                k_drain<<<256 * grid_mult, THREADS>>>(queues, per_queue, table, cursor, mode, out, seen);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            CK(hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost));
            printf("blocks/CU %2d  %-36s %7.2f ms  %6.1f G records/s", grid_mult, names[mode], best, n_total / best / 1e6);
            if (mode == 0 && grid_mult == 8) {
                printf("   [XCDs seen by blockIdx mod 8 = 0..7:");
                for (int i = 0; i < 8; i++) printf(" %02x", hs[i]);
                printf("]");
            }
            printf("\n");
        }
    return 0;
}
