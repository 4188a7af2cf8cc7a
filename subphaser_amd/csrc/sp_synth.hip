// sp_synth.hip -- bench support: deterministic synthetic allopolyploid
// chromosome, generated directly in HBM as ASCII (there are no real genomes in
// the sandbox).  Counter-based: every base is a pure function of
// (seed, chromosome identity, position), so any sub-range can be regenerated
// and the CPU baseline sample is simply copied back from the device.
//
// Structure per SURVEY.md section 8(d): an ancestral backbone shared by the
// homoeologs of a set (8 % per-copy substitutions), 25 % subgenome-specific
// repeats and 35 % shared repeats drawn from 200-family libraries with a
// Zipf-like abundance and up to 6 % per-copy divergence, 30 % of repeat copies
// soft-masked, 1 kb of N every 50 Mb, 100 kb of (TTTAGGG)n at both ends (hot
// keys) and, on request, a planted inter-subgenome exchange.
#include "sp_device.h"

__host__ __device__ inline uint64_t synth_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t synth_h(uint64_t seed, uint64_t stream, uint64_t idx) {
    return synth_mix(synth_mix(seed ^ (stream * 0xD6E8FEB86659FD93ULL)) + idx);
}

#define SYN_SEG 1024
#define SYN_FAMILIES 200

struct sp_synth_params {
    int64_t len;       // length of the whole chromosome
    int64_t start, n;  // positions [start, start + n) are written to out[0 .. n)
    uint64_t seed;
    int set_id, sg_id, n_sg, chrom_id, exchange;
};

__host__ __device__ inline uint8_t synth_base(int64_t p, const sp_synth_params &P) {
    if (p < 100000 || p >= P.len - 100000) return (uint8_t)("TTTAGGG"[p % 7]);
    if (p >= 50000000 && (p % 50000000) < 1000) return (uint8_t)'N';
    const uint64_t q = (uint64_t)(p / SYN_SEG);
    const uint64_t hq = synth_h(P.seed, 100 + (uint64_t)P.chrom_id, q);
    const unsigned cls = (unsigned)(hq % 100);
    unsigned base;
    bool lower = false;
    if (cls < 40) {
        base = (unsigned)(synth_h(P.seed, 1000 + (uint64_t)P.set_id, (uint64_t)p) & 3);
        uint64_t r = synth_h(P.seed, 2000 + (uint64_t)P.set_id * 64 + (uint64_t)P.sg_id, (uint64_t)p);
        if ((r & 0xffff) < 5243) base = (base + 1 + (unsigned)((r >> 16) % 3)) & 3;  // 8 %
    } else {
        unsigned lib;
        if (cls < 65) {
            lib = (unsigned)P.sg_id;
            if (P.exchange && p >= P.len / 10 * 6 && p < P.len / 10 * 7) lib = (unsigned)((P.sg_id + 1) % P.n_sg);
        } else {
            lib = 255;
        }
        // Zipf-like family: f + 1 = F^u, u uniform
        const double u = (double)((hq >> 8) & 0xffffff) / 16777216.0;
        unsigned f = (unsigned)(exp(u * 5.303304908059076) /* ln 201 */) - 1;
        if (f >= SYN_FAMILIES) f = SYN_FAMILIES - 1;
        const unsigned nblk = 1 + f % 8;  // family length 1..8 kb
        const unsigned blk = (unsigned)((hq >> 40) % nblk);
        const uint64_t cpos = (uint64_t)blk * SYN_SEG + (uint64_t)(p % SYN_SEG);
        base = (unsigned)(synth_h(P.seed, 5000 + (uint64_t)lib * 1024 + f, cpos) & 3);
        const unsigned div16 = (unsigned)(((hq >> 32) & 0xff) * 3932 / 255);  // up to 6 % of 65536
        uint64_t r = synth_h(P.seed, 3000 + (uint64_t)P.chrom_id, (uint64_t)p);
        if ((r & 0xffff) < div16) base = (base + 1 + (unsigned)((r >> 16) % 3)) & 3;
        lower = ((hq >> 52) % 10) < 3;
    }
    uint8_t ch = (uint8_t)("ACGT"[base]);
    return lower ? (uint8_t)(ch | 0x20) : ch;
}

__global__ void __launch_bounds__(256)
synth_fill(uint8_t *__restrict__ out, sp_synth_params P) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n16 = (P.n + 15) / 16;
    const bool aligned = (((uintptr_t)out) & 15) == 0;
    for (; t < n16; t += stride) {
        int64_t p0 = t * 16;
        if (aligned && p0 + 16 <= P.n) {
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t v = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) v |= (uint32_t)synth_base(P.start + p0 + q * 4 + j, P) << (8 * j);
                w[q] = v;
            }
            *reinterpret_cast<uint4 *>(out + p0) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int j = 0; j < 16 && p0 + j < P.n; j++) out[p0 + j] = synth_base(P.start + p0 + j, P);
        }
    }
}

extern "C" int sp_synth_chrom_range(sp_ctx *ctx, uint8_t *d_out, int64_t len, int64_t start, int64_t n, uint64_t seed,
                                    int set_id, int sg_id, int n_sg, int chrom_id, int exchange) {
    if (!ctx || !d_out || len <= 0 || n_sg < 1 || start < 0 || n < 0 || start + n > len)
        return sp_fail(ctx, SP_EINVAL, "sp_synth_chrom_range: bad arguments");
    if (n == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    sp_synth_params P;
    P.len = len;
    P.start = start;
    P.n = n;
    P.seed = seed;
    P.set_id = set_id;
    P.sg_id = sg_id;
    P.n_sg = n_sg;
    P.chrom_id = chrom_id;
    P.exchange = exchange;
    int64_t blocks = ((n + 15) / 16 + 255) / 256;
    if (blocks > (int64_t)ctx->n_cu * 32) blocks = (int64_t)ctx->n_cu * 32;
    SP_LAUNCH(ctx, "synth_fill", synth_fill, dim3((unsigned)blocks), dim3(256), 0, d_out, P);
    return SP_OK;
}

extern "C" int sp_synth_chrom(sp_ctx *ctx, uint8_t *d_out, int64_t len, uint64_t seed, int set_id,
                              int sg_id, int n_sg, int chrom_id, int exchange) {
    if (len <= 0) return sp_fail(ctx, SP_EINVAL, "sp_synth_chrom: bad arguments");
    return sp_synth_chrom_range(ctx, d_out, len, 0, len, seed, set_id, sg_id, n_sg, chrom_id, exchange);
}
