// sp_ctx.hip -- context, error plumbing, profiling, genome ingest (K0 pack).
#include <stdarg.h>

#include "sp_common.h"

thread_local std::string g_sp_err;

int sp_fail(sp_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_sp_err = buf;
    return code;
}

int sp_scratch(sp_ctx *ctx, int64_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->d_scratch) {
            SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            SP_HIP(ctx, hipFree(ctx->d_scratch));
            ctx->d_scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        SP_HIP(ctx, hipMalloc(&ctx->d_scratch, (size_t)bytes));
        ctx->scratch_bytes = bytes;
    }
    *out = ctx->d_scratch;
    return SP_OK;
}

int sp_buf_ensure(sp_ctx *ctx, sp_buf &b, int64_t bytes) {
    if (bytes <= b.cap) return SP_OK;
    if (b.p) {
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SP_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    int64_t want = bytes + bytes / 8 + 4096;
    SP_HIP(ctx, hipMalloc(&b.p, (size_t)want));
    b.cap = want;
    return SP_OK;
}
void sp_buf_free(sp_buf &b) {
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

void sp_prof_begin(sp_ctx *ctx, const char *name) {
    if (!ctx->prof) return;
    sp_prof_entry e;
    e.name = name;
    hipEventCreate(&e.e0);
    hipEventCreate(&e.e1);
    hipEventRecord(e.e0, ctx->stream);
    ctx->prof_pending.push_back(e);
}
void sp_prof_end(sp_ctx *ctx) {
    if (!ctx->prof) return;
    hipEventRecord(ctx->prof_pending.back().e1, ctx->stream);
}
void sp_prof_flush(sp_ctx *ctx) {
    if (ctx->prof_pending.empty()) return;
    hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->prof_pending) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, e.e0, e.e1);
        auto &acc = ctx->prof_acc[e.name];
        acc.first += 1;
        acc.second += ms;
        hipEventDestroy(e.e0);
        hipEventDestroy(e.e1);
    }
    ctx->prof_pending.clear();
}

void sp_sparse_release(sp_ctx *ctx);   // sp_sparse.hip

extern "C" {

int sp_version(void) { return 100; }

const char *sp_last_error(const sp_ctx *ctx) { return ctx ? ctx->err.c_str() : g_sp_err.c_str(); }

int sp_ctx_create(int device, void *stream, sp_ctx **out) {
    if (!out) return sp_fail(nullptr, SP_EINVAL, "sp_ctx_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return sp_fail(nullptr, SP_ENODEV, "no HIP device visible (libsubphaser_hip needs an MI355X)");
    if (device < 0 || device >= n)
        return sp_fail(nullptr, SP_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    sp_ctx *ctx = new sp_ctx();
    ctx->device = device;
    SP_HIP(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    SP_HIP(ctx, hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        SP_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    *out = ctx;
    return SP_OK;
}

static void free_chrom(sp_chrom &c) {
    if (c.d_pk) hipFree(c.d_pk);
    if (c.d_nm) hipFree(c.d_nm);
    if (c.d_tab && !c.tab_external) hipFree(c.d_tab);
    if (c.d_ovf) hipFree(c.d_ovf);
    if (c.d_ovf_idx) hipFree(c.d_ovf_idx);
    c.d_ovf_idx = nullptr;
    c.ovf_idx_n = c.ovf_idx_cap = 0;
    if (c.ev_packed) hipEventDestroy(c.ev_packed);
    c = sp_chrom();
}

static void free_filter(sp_ctx *ctx) {
    if (ctx->d_flag_row) hipFree(ctx->d_flag_row);
    if (ctx->d_flag_hist) hipFree(ctx->d_flag_hist);
    if (ctx->d_blk_row) hipFree(ctx->d_blk_row);
    if (ctx->d_blk_hist) hipFree(ctx->d_blk_hist);
    ctx->d_flag_row = ctx->d_flag_hist = nullptr;
    ctx->d_blk_row = ctx->d_blk_hist = nullptr;
    ctx->filtered = false;
}

int sp_ctx_destroy(sp_ctx *ctx) {
    if (!ctx) return SP_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    sp_prof_flush(ctx);
    for (auto &c : ctx->chroms) free_chrom(c);
    free_filter(ctx);
    sp_sparse_release(ctx);
    if (ctx->d_label) hipFree(ctx->d_label);
    if (ctx->d_bloom) hipFree(ctx->d_bloom);
    if (ctx->d_scratch) hipFree(ctx->d_scratch);
    if (ctx->d_ws2) hipFree(ctx->d_ws2);
    for (auto &ln : ctx->lanes) {
        if (ln.d_ws2) hipFree(ln.d_ws2);
        sp_buf_free(ln.b_ovfw);
        if (ln.h_desc) hipHostFree(ln.h_desc);
        sp_buf_free(ln.b_sp_a);
        sp_buf_free(ln.b_sp_b);
        sp_buf_free(ln.b_sp_c);
        sp_buf_free(ln.b_sp_tmp);
        sp_buf_free(ln.b_s3_small);
        if (ln.ev_a) hipEventDestroy(ln.ev_a);
        if (ln.ev_b) hipEventDestroy(ln.ev_b);
        if (ln.done) hipEventDestroy(ln.done);
        if (ln.stream) hipStreamDestroy(ln.stream);
    }
    if (ctx->lane_go) hipEventDestroy(ctx->lane_go);
    if (ctx->h_s3) hipHostFree(ctx->h_s3);
    if (ctx->h_desc) hipHostFree(ctx->h_desc);
    for (auto &e : ctx->s3_ev)
        if (e) hipEventDestroy(e);
    sp_buf_free(ctx->b_map);
    sp_buf_free(ctx->b_mapdesc);
    sp_buf_free(ctx->b_ival);
    sp_buf_free(ctx->b_ptab);
    sp_buf_free(ctx->b_lflags);
    if (ctx->h_lflags) hipHostFree(ctx->h_lflags);
    sp_buf_free(ctx->b_cntlen);
    sp_buf_free(ctx->b_ctab);
    sp_buf_free(ctx->b_covf);
    sp_buf_free(ctx->b_tab32);
    sp_buf_free(ctx->b_ovfw);
    sp_buf_free(ctx->b_labkeys);
    sp_buf_free(ctx->b_emit);
    sp_buf_free(ctx->b_slots);
    sp_buf_free(ctx->b_fpar);
    sp_buf_free(ctx->b_fflat);
    sp_buf_free(ctx->b_fq);
    sp_buf_free(ctx->b_wtab);
    sp_buf_free(ctx->b_enr);
    sp_buf_free(ctx->b_tt);
    sp_buf_free(ctx->b_win);
    if (ctx->copy_stream) {
        hipStreamSynchronize(ctx->copy_stream);
        hipStreamDestroy(ctx->copy_stream);
        hipEventDestroy(ctx->copy_event);
    }
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return SP_OK;
}

int sp_sync(sp_ctx *ctx) {
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

void *sp_stream(sp_ctx *ctx) { return (void *)ctx->stream; }

int sp_genome_reset(sp_ctx *ctx, int n_chrom) {
    if (!ctx || n_chrom < 0) return sp_fail(ctx, SP_EINVAL, "sp_genome_reset: bad arguments");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((size_t)n_chrom == ctx->chroms.size()) {
        // same shape as before: keep every device buffer (packed genome, tables) for reuse
        for (auto &c : ctx->chroms) {
            c.len = 0;
            c.nw = 0;
            c.length_sum = 0;
            c.n_dump = 0;
            c.n_ovf = 0;
            c.ovf_idx_n = 0;     // (the buffer stays; the next count's overflow pass fills it again)
        }
    } else {
        for (auto &c : ctx->chroms) free_chrom(c);
        ctx->chroms.assign((size_t)n_chrom, sp_chrom());
    }
    ctx->counted = false;
    ctx->filtered = false;
    return SP_OK;
}
}  // extern "C"

// ------------------------------------------------------------------ K0 ----
// ASCII -> 2-bit codes + invalid mask.  One thread packs 32 bases (two code
// words, one mask word): 32 B in (two 16-B loads when aligned), 12 B out.
// HBM-bound: 1 B/base read + 0.375 B/base written.
__device__ __forceinline__ void pack_byte(uint32_t b, uint32_t &code, uint32_t &inv) {
    uint32_t u = b & 0xDFu;  // fold case
    code = ((b >> 1) ^ (b >> 2)) & 3u;  // A/a=0 C/c=1 G/g=2 T/t=3
    inv = !(u == 'A' || u == 'C' || u == 'G' || u == 'T');
}

// Four bases at once (round 4: the byte-by-byte form is 425 VALU instructions per 32 bases -- k0_pack was bound by
// exactly that, 5.4 of its 5.5 ms per wheat-like pass): codes of the four bytes of `w`, the letters those codes stand
// for through one byte permute, a zero-byte test of letters ^ folded input for validity, then the codes squeezed into
// eight bits and the four invalid flags into four.
__device__ __forceinline__ void pack_word4(uint32_t w, uint32_t &c8, uint32_t &inv4) {
    uint32_t t = ((w >> 1) ^ (w >> 2)) & 0x03030303u;                       // A/a = 0, C/c = 1, G/g = 2, T/t = 3
    const uint32_t letters = __builtin_amdgcn_perm(0u, 0x54474341u, t);     // 'A' 'C' 'G' 'T' by code
    const uint32_t x = letters ^ (w & 0xDFDFDFDFu);                         // zero byte <=> a valid base
    const uint32_t f = ((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u) >> 7;   // 1 in every invalid byte
    t &= ~(f * 3u);                                                         // invalid bases pack as code 0
    t |= t >> 6;
    t |= t >> 12;
    c8 = t & 0xFFu;
    inv4 = ((f * 0x00204081u) >> 21) & 0xFu;                                // bits 0, 8, 16, 24 -> bits 0..3
}
// MSB-first twin of an LSB-first word of 16 codes: reverse the bits, then swap the two bits of every code back
__device__ __forceinline__ uint32_t pack_msb_first(uint32_t w) {
    const uint32_t r = __brev(w);
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}

__global__ void __launch_bounds__(256)
k0_pack(const uint8_t *__restrict__ ascii, int64_t len, uint32_t *__restrict__ pk, uint32_t *__restrict__ pm,
        uint32_t *__restrict__ nm, int64_t n_mask_words) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool aligned = (((uintptr_t)ascii) & 15) == 0;
    for (; t < n_mask_words; t += stride) {
        int64_t base = t * 32;
        uint32_t w0 = 0, w1 = 0, m = 0, r0 = 0, r1 = 0;   // r*: MSB-first twins
        if (base + 32 <= len && aligned) {
            const uint4 *p = reinterpret_cast<const uint4 *>(ascii + base);
            uint4 a = p[0], b = p[1];
            uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                uint32_t c8, inv4;
                pack_word4(v[q], c8, inv4);
                if (q < 4) w0 |= c8 << (8 * q);
                else w1 |= c8 << (8 * (q - 4));
                m |= inv4 << (4 * q);
            }
            r0 = pack_msb_first(w0);
            r1 = pack_msb_first(w1);
        } else {
            for (int i = 0; i < 32; i++) {
                uint32_t c = 0, iv = 1;
                if (base + i < len) pack_byte(ascii[base + i], c, iv);
                if (iv) c = 0;
                if (i < 16) { w0 |= c << (2 * i); r0 |= c << (30 - 2 * i); }
                else { w1 |= c << (2 * (i - 16)); r1 |= c << (30 - 2 * (i - 16)); }
                m |= iv << i;
            }
        }
        *reinterpret_cast<uint2 *>(pk + 2 * t) = make_uint2(w0, w1);
#if !SP_DERIVE_PM      // (round 5: the scans derive the MSB-first words from the LSB-first ones, sp_device.h)
        *reinterpret_cast<uint2 *>(pm + 2 * t) = make_uint2(r0, r1);
#else
        (void)r0; (void)r1; (void)pm;
#endif
        nm[t] = m;
    }
}

__global__ void __launch_bounds__(256)
k0_unpack(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, int64_t len,
          uint8_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < len; i += stride) {
        uint32_t c = (pk[i >> 4] >> (2 * (i & 15))) & 3u;
        uint32_t iv = (nm[i >> 5] >> (i & 31)) & 1u;
        out[i] = iv ? 'N' : "ACGT"[c];
    }
}

static int genome_add_impl(sp_ctx *ctx, int chrom, const uint8_t *d_ascii, int64_t len) {
    sp_chrom &c = ctx->chroms[(size_t)chrom];
    c.len = len;
    c.length_sum = 0;
    c.n_dump = 0;
    int64_t nmw = (len + 31) / 32 + SP_PAD_WORDS;  // mask words incl. padding
    c.nw = (len + 15) / 16;
    if (nmw > c.cap_mw) {
        if (c.d_pk) hipFree(c.d_pk);
        if (c.d_nm) hipFree(c.d_nm);
        c.d_pk = c.d_pm = c.d_nm = nullptr;
        c.cap_mw = 0;
#if SP_DERIVE_PM      // one stream: 0.375 B/base resident instead of 0.625 (3.5 GB less for the wheat-like genome)
        SP_HIP(ctx, hipMalloc(&c.d_pk, (size_t)(2 * nmw) * sizeof(uint32_t)));
        c.d_pm = c.d_pk;
#else
        SP_HIP(ctx, hipMalloc(&c.d_pk, (size_t)(4 * nmw) * sizeof(uint32_t)));   // LSB-first | MSB-first
        c.d_pm = c.d_pk + 2 * nmw;
#endif
        SP_HIP(ctx, hipMalloc(&c.d_nm, (size_t)nmw * sizeof(uint32_t)));
        c.cap_mw = nmw;
    }
    int64_t blocks = (nmw + 255) / 256;
    if (blocks > 8LL * ctx->n_cu * 8) blocks = 8LL * ctx->n_cu * 8;
    if (blocks < 1) blocks = 1;
    SP_LAUNCH(ctx, "k0_pack", k0_pack, dim3((unsigned)blocks), dim3(256), 0, d_ascii, len, c.d_pk,
              c.d_pm, c.d_nm, nmw);
    if (!c.ev_packed) SP_HIP(ctx, hipEventCreateWithFlags(&c.ev_packed, hipEventDisableTiming));
    SP_HIP(ctx, hipEventRecord(c.ev_packed, ctx->stream));
    ctx->counted = false;
    return SP_OK;
}

extern "C" {

int sp_genome_add_device(sp_ctx *ctx, int chrom, const uint8_t *d_ascii, int64_t len) {
    if (!ctx || chrom < 0 || chrom >= (int)ctx->chroms.size() || len < 0 || (!d_ascii && len > 0))
        return sp_fail(ctx, SP_EINVAL, "sp_genome_add_device: bad arguments (chrom=%d len=%lld)", chrom,
                       (long long)len);
    SP_HIP(ctx, hipSetDevice(ctx->device));
    return genome_add_impl(ctx, chrom, d_ascii, len);
}

int sp_genome_add(sp_ctx *ctx, int chrom, const uint8_t *ascii, int64_t len) {
    if (!ctx || chrom < 0 || chrom >= (int)ctx->chroms.size() || len < 0 || (!ascii && len > 0))
        return sp_fail(ctx, SP_EINVAL, "sp_genome_add: bad arguments (chrom=%d len=%lld)", chrom,
                       (long long)len);
    SP_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    SP_HIP(ctx, hipMalloc(&d, (size_t)(len > 0 ? len : 1)));
    hipError_t e = hipMemcpyAsync(d, ascii, (size_t)len, hipMemcpyHostToDevice, ctx->stream);
    int rc = SP_OK;
    if (e != hipSuccess) rc = sp_fail(ctx, SP_EHIP, "H2D copy failed: %s", hipGetErrorString(e));
    if (rc == SP_OK) rc = genome_add_impl(ctx, chrom, d, len);
    hipStreamSynchronize(ctx->stream);
    hipFree(d);
    return rc;
}

int sp_genome_len(sp_ctx *ctx, int chrom, int64_t *len) {
    if (!ctx || !len || chrom < 0 || chrom >= (int)ctx->chroms.size())
        return sp_fail(ctx, SP_EINVAL, "sp_genome_len: bad arguments");
    *len = ctx->chroms[(size_t)chrom].len;
    return SP_OK;
}

int sp_genome_unpack(sp_ctx *ctx, int chrom, uint8_t *ascii_out, int64_t len) {
    if (!ctx || chrom < 0 || chrom >= (int)ctx->chroms.size() || !ascii_out)
        return sp_fail(ctx, SP_EINVAL, "sp_genome_unpack: bad arguments");
    sp_chrom &c = ctx->chroms[(size_t)chrom];
    if (len != c.len) return sp_fail(ctx, SP_EINVAL, "sp_genome_unpack: len mismatch");
    if (len == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    SP_HIP(ctx, hipMalloc(&d, (size_t)len));
    int64_t blocks = (len + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    SP_LAUNCH(ctx, "k0_unpack", k0_unpack, dim3((unsigned)blocks), dim3(256), 0, c.d_pk, c.d_nm, len, d);
    SP_HIP(ctx, hipMemcpyAsync(ascii_out, d, (size_t)len, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(d);
    return SP_OK;
}

int sp_prof_enable(sp_ctx *ctx, int on) {
    if (!ctx) return SP_EINVAL;
    if (!on) sp_prof_flush(ctx);
    ctx->prof = on != 0;
    return SP_OK;
}
int sp_prof_reset(sp_ctx *ctx) {
    if (!ctx) return SP_EINVAL;
    sp_prof_flush(ctx);
    ctx->prof_acc.clear();
    return SP_OK;
}
int sp_prof_report(sp_ctx *ctx, char *buf, int64_t cap) {
    if (!ctx || !buf || cap < 3) return SP_EINVAL;
    sp_prof_flush(ctx);
    std::string s = "{";
    bool first = true;
    for (auto &kv : ctx->prof_acc) {
        char tmp[256];
        snprintf(tmp, sizeof tmp, "%s\"%s\": {\"calls\": %lld, \"ms\": %.6f}", first ? "" : ", ",
                 kv.first.c_str(), (long long)kv.second.first, kv.second.second);
        s += tmp;
        first = false;
    }
    s += "}";
    if ((int64_t)s.size() + 1 > cap) return sp_fail(ctx, SP_EINVAL, "sp_prof_report: buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return SP_OK;
}

int sp_host_alloc(sp_ctx *ctx, int64_t bytes, void **h_ptr) {
    if (!ctx || !h_ptr || bytes < 0) return sp_fail(ctx, SP_EINVAL, "sp_host_alloc: bad arguments");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipHostMalloc(h_ptr, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault));
    return SP_OK;
}
int sp_host_free(sp_ctx *ctx, void *h_ptr) {
    if (!ctx) return SP_EINVAL;
    SP_HIP(ctx, hipHostFree(h_ptr));
    return SP_OK;
}
int sp_host_register(sp_ctx *ctx, void *h_ptr, int64_t bytes) {
    if (!ctx || !h_ptr || bytes <= 0) return sp_fail(ctx, SP_EINVAL, "sp_host_register: bad arguments");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipHostRegister(h_ptr, (size_t)bytes, hipHostRegisterPortable));
    return SP_OK;
}
int sp_host_unregister(sp_ctx *ctx, void *h_ptr) {
    if (!ctx || !h_ptr) return SP_EINVAL;
    SP_HIP(ctx, hipHostUnregister(h_ptr));
    return SP_OK;
}
int sp_dev_alloc(sp_ctx *ctx, int64_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr || bytes < 0) return sp_fail(ctx, SP_EINVAL, "sp_dev_alloc: bad arguments");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipMalloc(d_ptr, (size_t)(bytes > 0 ? bytes : 1)));
    return SP_OK;
}
int sp_dev_free(sp_ctx *ctx, void *d_ptr) {
    if (!ctx) return SP_EINVAL;
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SP_HIP(ctx, hipFree(d_ptr));
    return SP_OK;
}
int sp_dev_copy_to_host(sp_ctx *ctx, void *dst, const void *d_src, int64_t bytes) {
    if (!ctx || !dst || !d_src || bytes < 0) return sp_fail(ctx, SP_EINVAL, "sp_dev_copy_to_host: bad arguments");
    SP_HIP(ctx, hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}
int sp_dev_copy_from_host(sp_ctx *ctx, void *d_dst, const void *src, int64_t bytes) {
    if (!ctx || !d_dst || !src || bytes < 0) return sp_fail(ctx, SP_EINVAL, "sp_dev_copy_from_host: bad arguments");
    SP_HIP(ctx, hipMemcpyAsync(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}
}  // extern "C"
