// ubench_mall.hip -- does the 256-MB Infinity Cache absorb a write -> read -> overwrite cycle?  (round 5)
// The count chain writes every fine-bucket record (2 B per base) and reads it back in the next kernel.  If a region
// that is written, read and then overwritten never has to reach HBM while it fits the memory-side cache, the chain can
// cycle its level-2 buffer through a small window (count a group of level-1 buckets at a time) and shed 4 of its
// 13.8 B/base of HBM traffic.  Test: total 4 GiB written + 4 GiB read, through a window of S MiB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mall tools/ubench_mall.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(512) k_write(uint4 *p, size_t n16, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) p[i] = make_uint4(v, (uint32_t)i, v, v);
}
__global__ void __launch_bounds__(512) k_read(const uint4 *p, size_t n16, unsigned long long *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x1234567u) atomicAdd(sink, 1ULL);
}
__global__ void __launch_bounds__(512) k_copy(const uint4 *s, uint4 *d, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) d[i] = s[i];
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t total = 4ull << 30;
    uint8_t *buf, *src;
    unsigned long long *sink;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&src, total));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 0, total));
    CK(hipMemset(src, 1, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = cus * 8;
    for (size_t mb : {16, 32, 64, 128, 192, 256, 512, 1024, 4096}) {
        const size_t S = mb << 20, n16 = S / 16;
        const int reps = (int)(total / S);
        for (int mode = 0; mode < 3; mode++) {    // 0: write window, read window; 1: write only; 2: stream a 4-GiB source THROUGH the window (copy in, read back)
            float best = 1e9;
            for (int t = 0; t < 3; t++) {
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; r++) {
                    if (mode == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(512), 0, 0, (const uint4 *)(src + (size_t)r * S), (uint4 *)buf, n16);
                    else hipLaunchKernelGGL(k_write, dim3(grid), dim3(512), 0, 0, (uint4 *)buf, n16, (uint32_t)r);
                    if (mode != 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(512), 0, 0, (const uint4 *)buf, n16, sink);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double moved = (mode == 1 ? 1.0 : mode == 0 ? 2.0 : 3.0) * (double)total;
            printf("window %4zu MiB x %3d, %s: %.3f ms, %.0f GB/s of kernel traffic (%d launches)\n", mb, reps,
                   mode == 0 ? "write + read      " : mode == 1 ? "write only        " : "copy in + read    ", best, moved / best / 1e6,
                   reps * (mode == 1 ? 1 : 2));
        }
    }
    return 0;
}
