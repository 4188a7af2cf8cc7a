// ubench_wc.hip -- can a ONE-LEVEL partition of the k <= 15 count chain be fed by this memory system?  (round 5)
//
// A one-level scheme needs F = 2048..8192 buckets (2^16..2^18 slots each, counted in 16-bit LDS counters).  A tile of
// 16-32 K keys then holds 4..16 records per bucket: per-run global cursor atomics are out (27 G/s: one per <= 16 keys),
// so every workgroup appends to its OWN chunk of every bucket (position kept in LDS) and the pieces it appends are
// 12..96 bytes.  Nothing on the CU can combine them (F x 64 B of staging does not fit next to a tile in 160 KB of LDS);
// the L2 (4 MiB per XCD against F x 128 B x 64 workgroups of open lines) cannot either.  What is left is the memory
// side.  Test A measures it: n_wg workgroups, F private append streams each, one piece per stream and round, chunks of
// different workgroups interleaved inside a bucket's region exactly as a cursor would hand them out.
// Test B: G workgroups of one XCD stream the SAME buffer (a bucket read once per 2^16-slot part of it): what does the
// 2nd..Gth reader cost?  Test C: a dependent VALU chain with a known instruction count -> the clock the chip holds.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_wc tools/ubench_wc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

struct u3 { uint32_t x, y, z; };

// ---- A: private append streams
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_append(uint8_t *__restrict__ out, int F, int piece /* bytes, multiple of 12 */, int chunk /* bytes, multiple of piece */,
         int rounds, size_t region /* bytes per bucket */, unsigned long long *__restrict__ written) {
    const int qpp = piece / 12;                 // 12-byte quads per piece
    const int n_wg = gridDim.x, wg = blockIdx.x;
    unsigned long long mine = 0;
    for (int r = 0; r < rounds; r++) {
        const size_t p = (size_t)r * piece;     // stream position
        const size_t ci = p / chunk, in = p % chunk;
        const size_t off = (ci * n_wg + wg) * (size_t)chunk + in;     // chunk ci of this workgroup inside a bucket's region
        for (int q = threadIdx.x; q < F * qpp; q += THREADS) {
            const int b = q / qpp, w = q % qpp;
            u3 v; v.x = q; v.y = r; v.z = wg;
            *reinterpret_cast<u3 *>(out + (size_t)b * region + off + (size_t)w * 12) = v;
            mine += 12;
        }
        __syncthreads();    // a tile's copy-out ends in a barrier
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(written, mine);
}

// ---- B: G readers of one XCD on one buffer.  Workgroup b runs on XCD b mod 8 (tools/ubench_xcd.hip): the G readers
// of buffer j are the workgroups x + 8 (G q + s), s = 0..G-1 with j = 8 q + x.
__global__ void __launch_bounds__(1024)
k_share(const uint4 *__restrict__ in, size_t n16 /* 16-byte words per buffer */, int G, int n_buf,
        unsigned long long *__restrict__ sink) {
    const int x = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int q = t / G;
    uint32_t acc = 0;
    for (int j = 8 * q + x; j < n_buf; j += (gridDim.x / G)) {
        const uint4 *p = in + (size_t)j * n16;
        for (size_t i = threadIdx.x; i < n16; i += 1024) {
            const uint4 v = p[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345u) atomicAdd(sink, 1ULL);
}

// ---- C: clock
__global__ void __launch_bounds__(256)
k_clock(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x, b = blockIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) { a = a * 3u + b; b = b ^ (a >> 3); }
    }
    if (a == 0x1234567u) out[0] = b;
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    const size_t cap = 6ull << 30;
    uint8_t *buf;
    unsigned long long *d_w;
    CK(hipMalloc(&buf, cap));
    CK(hipMalloc(&d_w, 8));
    CK(hipMemset(buf, 0, cap));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    {   // C first: what clock does a VALU-only kernel hold
        uint32_t *d_o;
        CK(hipMalloc(&d_o, 4));
        for (int wpc = 4; wpc <= 16; wpc *= 2) {     // waves per SIMD = wpc / 4 ... 256-thread blocks = 4 waves
            const int blocks = cus * wpc / 4, iters = 20000;
            hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, d_o, 1000);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, d_o, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            // per iteration 64 x (mul-add: 1-2 VALU, xor-shift: 2 VALU): read the ISA; report wave-instr rate assuming 4 VALU per j
            const double winstr = (double)blocks * 4 * iters * 64 * 4;
            printf("clock: %2d waves/CU: %.3f ms, %.1f G wave-VALU/s (if 4 VALU per step; 1024 SIMDs x f / 4 cycles -> f = %.2f GHz)\n",
                   wpc, ms, winstr / ms / 1e6, winstr / ms / 1e6 / 1024 * 4 / 1e3);
        }
    }

    // A
    struct cfg { int F, piece, chunk, wgpc, threads; };
    std::vector<cfg> cfgs;
    for (int wgpc : {2, 1})
        for (int F : {128, 1024, 2048, 4096})
            for (int piece : {12, 24, 48, 96, 384}) {
                if (F == 128 && piece != 384) continue;
                if (F != 128 && piece == 384) continue;
                for (int chunk : {96, 384, 1536}) {
                    if (chunk % piece) continue;
                    if (F == 128 && chunk != 384) continue;
                    cfgs.push_back({F, piece, chunk, wgpc, wgpc == 2 ? 512 : 1024});
                }
            }
    for (const cfg &c : cfgs) {
        const int n_wg = cus * c.wgpc;
        const size_t per_round = (size_t)n_wg * c.F * c.piece;
        int rounds = (int)((2ull << 30) / per_round);
        if (rounds < 2) rounds = 2;
        // stream length rounded up to whole chunks; region = all workgroups' chunks
        const size_t stream = (((size_t)rounds * c.piece + c.chunk - 1) / c.chunk) * c.chunk;
        const size_t region = stream * n_wg;
        if (region * c.F > cap) { printf("skip F=%d piece=%d\n", c.F, c.piece); continue; }
        float best = 1e9;
        unsigned long long w = 0;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(d_w, 0, 8));
            CK(hipEventRecord(e0));
            if (c.threads == 512)
                hipLaunchKernelGGL(k_append<512>, dim3(n_wg), dim3(512), 0, 0, buf, c.F, c.piece, c.chunk, rounds, region, d_w);
            else
                hipLaunchKernelGGL(k_append<1024>, dim3(n_wg), dim3(1024), 0, 0, buf, c.F, c.piece, c.chunk, rounds, region, d_w);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(&w, d_w, 8, hipMemcpyDeviceToHost));
        }
        printf("append: %d WG/CU x %4d thr, F=%4d, piece %3d B, chunk %4d B, %4d rounds: %.3f ms, %6.0f GB/s (%.2f GB)\n", c.wgpc,
               c.threads, c.F, c.piece, c.chunk, rounds, best, (double)w / best / 1e6, (double)w / 1e9);
        fflush(stdout);
    }

    // B
    for (size_t mb : {1, 2}) {
        const size_t n16 = (mb << 20) / 16;
        const int n_buf = (int)((2ull << 30) / (mb << 20));
        for (int G : {1, 2, 4, 8}) {
            const int grid = cus;           // one 1024-thread workgroup per CU
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_share, dim3(grid), dim3(1024), 0, 0, (const uint4 *)buf, n16, G, n_buf, d_w);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("share: %zu-MB buffers, %d readers each (one XCD): %.3f ms for 2 GiB of distinct data, %.0f GB/s distinct, %.0f GB/s delivered\n",
                   mb, G, best, (double)(2ull << 30) / best / 1e6, (double)(2ull << 30) * G / best / 1e6);
        }
    }
    return 0;
}
