// ubench_frag.hip -- what scattered bucket RUNS cost on this MI355X (round 4).  The partition kernels of the counting
// engines write, per 16 K-key tile, one short run per bucket (128 buckets: ~256 B in a u16 plane) at a position
// reserved with an atomic cursor; the neighbouring run of the same bucket is written by another workgroup, usually on
// another XCD (another L2).  Question: are the partial 128-B lines at both ends of every run what holds these kernels
// at half of the streaming rate?  Variants: run length fixed / ragged, run starts 8-B / 128-B aligned, one cursor
// per bucket / one per (bucket, XCD) so that neighbouring runs share an L2.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_frag tools/ubench_frag.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define NB 128
#define SLOTS 48          // quad slots per run (a run has <= 4 * SLOTS keys)
#define THREADS 512

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode bit 0: ragged run lengths (96..160 keys, multiples of 4) instead of 128
// mode bit 1: reservations rounded up to 64 keys (128-B aligned run starts, gaps left unwritten)
// mode bit 2: cursor per (bucket, XCD)
// mode bit 4: workgroup-scope atomic (executed in the XCD's own L2; with bit 2 every cursor has one L2)
// mode bit 3: no cursor atomic at all (positions computed from the tile number)
// cstride: distance between two cursors in 8-byte words (1: packed, 16: one 128-B line each, 512: one 4-KB page each)
template <typename REC>
__global__ void __launch_bounds__(THREADS)
k_frag(REC *__restrict__ plane, unsigned long long *__restrict__ cursor, size_t region /* keys per (bucket[, xcd]) */,
       int n_tiles, int mode, int cstride, unsigned long long *__restrict__ written) {
    __shared__ unsigned long long base[NB];
    __shared__ uint32_t len[NB];
    const uint32_t xcd = (mode & 4) ? xcc_id() : 0u;
    unsigned long long mine = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (threadIdx.x < NB) {
            uint32_t r = 128;
            if (mode & 1) r = 96 + 4 * (mix(tile * 131u + threadIdx.x) % 17u);
            const uint32_t res = (mode & 2) ? (r + 63u) & ~63u : r;
            const size_t c = (mode & 4) ? (size_t)threadIdx.x * 8 + xcd : (size_t)threadIdx.x;
            const unsigned long long at = (mode & 8) ? (unsigned long long)(tile / ((mode & 4) ? 8 : 1)) * 160ULL
                                                     : (mode & 16) ? __hip_atomic_fetch_add(&cursor[c * cstride], (unsigned long long)res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                                     : atomicAdd(&cursor[c * cstride], (unsigned long long)res);
            base[threadIdx.x] = c * region + at;
            len[threadIdx.x] = (at + res <= region) ? r : 0u;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < NB * SLOTS; q += THREADS) {
            const int b = q / SLOTS, s = q % SLOTS;
            if (4u * s < len[b]) {
                REC *p = plane + base[b] + 4u * s;
                if (sizeof(REC) == 2) *reinterpret_cast<uint2 *>(p) = make_uint2(q, tile);
                else *reinterpret_cast<uint4 *>(p) = make_uint4(q, tile, 3, 4);
                mine += 4;
            }
        }
        __syncthreads();
    }
    if (mine) atomicAdd(written, mine);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    const size_t total_keys = 1ull << 30;                 // 2 GiB as u16, 4 GiB as u32
    void *plane;
    unsigned long long *cursor, *written;
    CK(hipMalloc(&plane, total_keys * 4));
    CK(hipMalloc(&cursor, NB * 8 * 8 * 512));
    CK(hipMalloc(&written, 8));
    CK(hipMemset(plane, 0, total_keys * 4));
    const int n_tiles = 40000;                            // ~ one 670-Mb chromosome: 655 M keys
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rec = 0; rec < 2; rec++)
        for (int mode : {0, 1, 4, 5, 8, 9, 20, 21})
            for (int cstride : {1, 16, 512}) {
                if ((mode & 8) && cstride != 1) continue;
                const int bpc = 2;
                const size_t region = (mode & 4) ? total_keys / (NB * 8) : total_keys / NB;
                float best = 1e9f;
                unsigned long long w = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(cursor, 0, NB * 8 * 8 * 512));
                    CK(hipMemset(written, 0, 8));
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    if (rec == 0)
                        hipLaunchKernelGGL(k_frag<uint16_t>, dim3(cus * bpc), dim3(THREADS), 0, 0, (uint16_t *)plane, cursor, region, n_tiles, mode, cstride, written);
                    else
                        hipLaunchKernelGGL(k_frag<uint32_t>, dim3(cus * bpc), dim3(THREADS), 0, 0, (uint32_t *)plane, cursor, region, n_tiles, mode, cstride, written);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                    CK(hipMemcpy(&w, written, 8, hipMemcpyDeviceToHost));
                }
                printf("%s records, %s runs, %s, cursor stride %4d B: %7.3f ms, %6.0f GB/s written (%llu M keys)\n", rec ? "u32" : "u16",
                       (mode & 1) ? "ragged" : "fixed ", (mode & 8) ? "NO cursor atomic    " : (mode & 16) ? "cursor per (bucket, XCD), L2-scope atomic" : (mode & 4) ? "cursor per (bucket, XCD)" : "cursor per bucket   ",
                       cstride * 8, best, (double)w * (rec ? 4 : 2) / best / 1e6, w / 1000000ULL);
            }
    printf("done\n");
    return 0;
}
