// sp_common.h -- internals shared by the HIP translation units of libsubphaser_hip.so
// Target: gfx950 (MI355X, CDNA4) only.  64-wide wavefronts, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/subphaser_hip.h"

#define SP_WAVE 64

struct sp_chrom {
    bool tab_external = false;  // d_tab is caller-owned (sp_tables_bind): never freed here
    int64_t len = 0;     // bases
    int64_t nw = 0;      // 16-base words actually covering len
    int64_t cap_mw = 0;  // capacity of d_pk/d_nm in mask words (buffers are reused across sp_genome_reset)
    uint32_t *d_pk = nullptr;  // 2-bit codes, base i at bits 2*(i%16) of word i/16; padded with SP_PAD_WORDS
    uint32_t *d_pm = nullptr;  // the same codes MSB-first (base i at bits 30 - 2*(i%16)); lives behind d_pk in one allocation
    uint32_t *d_nm = nullptr;  // invalid mask, base i at bit (i%32) of word i/32; padding marked invalid
    uint8_t *d_tab = nullptr;  // dense RAW count table [nslots], one byte per slot: 0..254 = the count,
                               // 255 = the count is >= 255 and lives in the overflow list (valid after sp_count)
    uint2 *d_ovf = nullptr;    // overflow list: (slot, raw count >= 255), ascending slot
    int64_t n_ovf = 0, cap_ovf = 0;
    uint32_t *d_ovf_idx = nullptr;   // per-bucket starts of d_ovf (what ovf_scan computed), kept for the filter's look-ups
    int64_t ovf_idx_n = 0, ovf_idx_cap = 0;   // entries it holds (n_buckets + 1; 0: none) and its capacity
    int64_t length_sum = 0;    // sum of counts >= lower_count
    int64_t n_dump = 0;        // number of k-mers with count >= lower_count
    hipEvent_t ev_packed = nullptr;   // recorded behind this chromosome's pack kernel: a counting lane waits for it, not for
                                      // the packing of the chromosomes after it
};
#define SP_PAD_WORDS 8
#ifndef SP_DERIVE_PM
#define SP_DERIVE_PM 1      // the MSB-first packed stream is derived in registers instead of stored (sp_device.h)
#endif

// growth-only device buffer: hipMalloc/hipFree are slow (a hipMalloc that follows the release of
// tens of GiB was measured at 1.4-2.3 s on MI355X), so hot-path buffers are kept and reused.
struct sp_buf {
    void *p = nullptr;
    int64_t cap = 0;
};

// temporary device allocation, released at scope exit (error returns included)
template <typename T>
struct sp_tmp {
    T *p = nullptr;
    sp_tmp() = default;
    sp_tmp(const sp_tmp &) = delete;
    sp_tmp &operator=(const sp_tmp &) = delete;
    ~sp_tmp() {
        if (p) hipFree(p);
    }
    hipError_t alloc(size_t n_elems) { return hipMalloc((void **)&p, (n_elems ? n_elems : 1) * sizeof(T)); }
    operator T *() const { return p; }
};

// k > 15: per-chromosome sorted (canonical key, count >= lower_count) arrays
struct sp_sparse_chrom {
    uint64_t *d_keys = nullptr;
    uint32_t *d_cnts = nullptr;
    int64_t n = 0, cap = 0, length_sum = 0;
};

// what the filter / emit / dump kernels read: one chromosome's byte table (or a slot-range slice of it,
// already offset so that index = slot - slot_base) and its overflow list (absolute slots)
struct sp_tabref {
    const uint8_t *tab;
    const uint2 *ovf;
    int64_t n_ovf;
    // where the pairs of every bucket of 2^SP_OVF_SHIFT slots start in `ovf` (n_buckets + 1 entries), or NULL:
    // a look-up searches its bucket's handful of pairs instead of the whole list (merged / caller-owned lists have none)
    const uint32_t *ovf_idx = nullptr;
};

struct sp_prof_entry {
    std::string name;
    hipEvent_t e0, e1;
};

struct sp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // device->host copies that overlap the next stage (sp_filter_fetch_async)
    hipEvent_t copy_event = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    std::string err;
    // genome
    std::vector<sp_chrom> chroms;
    // counting
    int k = 0;
    int lower = 0;
    int64_t nslots = 0;   // dense table size
    bool counted = false;
    int64_t c2_recounts = 0;   // chromosomes engine 2 had to count twice (a bucket outgrew its sampled region)
    // filter view: which tables / slot range sp_filter works on (default: the local chromosomes,
    // all slots).  Multi-GPU runs point it at slot-range slices gathered from every rank.
    bool fv_on = false;
    std::vector<sp_tabref> fv_tabs;
    std::vector<int64_t> fv_lengths;
    int64_t fv_slot_base = 0, fv_nslots = 0;
    // k > 15 twin: caller-owned sorted (key, count) lists of one KEY RANGE of every chromosome
    bool sv_on = false;
    std::vector<const uint64_t *> sv_keys;
    std::vector<const uint32_t *> sv_cnts;
    std::vector<int64_t> sv_n;
    // filter results (device)
    bool filtered = false;
    int64_t n_union = 0, n_rows = 0, n_hist = 0;
    uint32_t *d_flag_row = nullptr;   // bitmap over slots: differential rows
    uint32_t *d_flag_hist = nullptr;  // bitmap over slots: fold-passing
    uint64_t *d_blk_row = nullptr;    // per-block exclusive offsets
    uint64_t *d_blk_hist = nullptr;
    int64_t n_fblocks = 0;
    // labels
    uint8_t *d_label = nullptr;  // [nslots] 0 = none, 1+sg; bit 7 = seen
    sp_buf b_ptab, b_labkeys;    // pair table (sp_map.hip: 4^(k-1) x u32) and the labelled keys it was built from
    int ptab_k = 0;              // k the pair table currently holds a label set for (0: not built / unknown state)
    int64_t ptab_n = 0;          // number of keys of that set (still in b_labkeys): sp_labels_set un-builds them
    unsigned long long *h_lflags = nullptr;   // page-locked twin of b_lflags, written by k4_flags_out
    sp_buf b_lflags;             // device flags of sp_labels_set (sp_map.hip)
    int64_t bloom_last_n = -1;   // label count and size of the pair filter sp_map_filter_build chose last (same count: same size, no fill check)
    int bloom_last_bits = 0, bloom_last_k = 0;
    sp_buf b_ctab, b_covf;       // compact pair table (S <= 3: buckets of two tagged entries) and its overflow table (sp_map.h)
    int ct_bb = 0;               // bucket bits of the compact table the current label set lives in (0: the direct table)
    uint32_t ct_ovf_mask = 0;
    int sq_bb = 0;               // k > 15: bucket bits of the quad-bucket table (sp_sparse.hip; lives in b_ctab / b_covf), 0: hash table
    uint64_t sq_ovf_mask = 0;
    int map_engine = 0;          // 0 = pair table (S <= 7), 1 = label table
    bool labels_ready = false;
    uint32_t *d_bloom = nullptr; // L2-resident pair filter over hashed (k-1)-mers (sp_map.hip), 2^bloom_bits bits
    int bloom_bits = 0;
    int n_sg = 0;
    int64_t n_labels = 0;
    // scratch
    void *d_scratch = nullptr;
    int64_t scratch_bytes = 0;
    void *d_ws2 = nullptr;       // engine-2 workspace (histograms, offsets, key buffers)
    int64_t ws2_bytes = 0;
    sp_buf b_cntlen;             // sp_count's per-chromosome tallies (sum, n, overflow pairs, overrun flag)
    sp_buf b_tab32, b_ovfw;      // engine 1: u32 scratch table; overflow staging (unordered pairs + per-bucket index)
    // Small genomes: the partition chain of a 20-Mb chromosome is ten launches of a few tens of microseconds that do
    // not fill the chip; chromosomes are counted on SP_LANES streams side by side, each with its own workspace.
    struct lane_t {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        void *d_ws2 = nullptr;
        int64_t ws2_bytes = 0;
        sp_buf b_ovfw;
        void *h_desc = nullptr;      // page-locked: the descriptor table of this lane's batched list count (sp_count2.hip)
        int64_t h_desc_cap = 0;
        // k > 15 (sp_sparse2.hip, round 4): this lane's partition buffers, small arrays and the two events its split-phase
        // chain is waited at
        sp_buf b_sp_a, b_sp_b, b_sp_c, b_sp_tmp, b_s3_small;
        hipEvent_t ev_a = nullptr, ev_b = nullptr;
    };
#define SP_MAX_LANES 7
    lane_t lanes[SP_MAX_LANES];  // lanes 1..7 (lane 0 = the context's own stream and buffers above)
    lane_t *lane = nullptr;      // the auxiliary lane the engine-2 chain is being issued on, or NULL
    void *h_desc = nullptr;      // page-locked descriptor table of a batched list count issued on the context's own stream
    int64_t h_desc_cap = 0;
    hipEvent_t lane_go = nullptr;
    // sparse engine (k = 16..32)
    bool sparse_mode = false;
    bool list_mode = false;     // k <= 15, engine 3: per-chromosome sorted (SLOT, count) lists in `sparse`, no byte tables;
                                // the map stage stays the dense pair-table one
    std::vector<sp_sparse_chrom> sparse;
    sp_buf b_sp_a, b_sp_b, b_sp_c, b_sp_tmp, b_sf_keys, b_sf_counts, b_sf_tot, b_sf_hist, b_s3_small, b_slots;
    unsigned long long *h_s3 = nullptr;   // page-locked: 8 words per k > 15 counting lane (flags and totals of a chromosome's chain)
    hipEvent_t s3_ev[2] = {nullptr, nullptr};   // the split-phase events of a chain issued on the context's own stream
    int64_t sf_n = 0;
    uint64_t *d_hkeys = nullptr;   // open-addressing hash table of the labelled k-mers: 16-B entries {key, label}
    int64_t hcap = 0;
    sp_buf b_tt;            // k-mer t-test workspace (sp_enrich.hip)
    sp_buf b_wtab, b_enr;   // window table (device) and the enrichment outputs
    sp_buf b_fq;      // global slow queue of the filter
    sp_buf b_fflat;   // flat set tables of the filter (sp_filter.hip)
    sp_buf b_map, b_mapdesc, b_ival, b_emit, b_fpar, b_win;  // reusable device buffers of sp_map_bins / k3_emit / sp_filter / stack
    bool map_all_valid = false;
    // profiling
    bool prof = false;
    std::vector<sp_prof_entry> prof_pending;
    std::map<std::string, std::pair<int64_t, double>> prof_acc;
};

extern thread_local std::string g_sp_err;

int sp_fail(sp_ctx *ctx, int code, const char *fmt, ...);
int sp_scratch(sp_ctx *ctx, int64_t bytes, void **out);
struct sp_ctx;
int sp_buf_ensure(sp_ctx *ctx, sp_buf &b, int64_t bytes);
void sp_buf_free(sp_buf &b);
void sp_prof_begin(sp_ctx *ctx, const char *name);
void sp_prof_end(sp_ctx *ctx);
void sp_prof_flush(sp_ctx *ctx);

#define SP_HIP(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess)                                                             \
            return sp_fail(ctx, e__ == hipErrorOutOfMemory ? SP_ENOMEM : SP_EHIP,          \
                           "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                           __LINE__);                                                      \
    } while (0)

// Launch wrapper: optional per-kernel event timing + launch error check.
#define SP_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                       \
    do {                                                                            \
        sp_prof_begin(ctx, name);                                                   \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__); \
        sp_prof_end(ctx);                                                           \
        SP_HIP(ctx, hipGetLastError());                                             \
    } while (0)

// ---------------------------------------------------------------- k-mer math
// Dense slot index.
//  odd k : every strand pair {x, rc(x)} has exactly one member whose middle
//          base is A or C (the middle base complements itself); that member
//          with the high bit of its middle base removed is a bijection onto
//          [0, 2^(2k-1)).
//  even k: slot = canonical value min(x, rc(x)) in [0, 4^k) (half the slots unused).
struct sp_kparams {
    int k;
    int odd;
    uint64_t kmask;  // 2k low bits
    int rcshift;     // 2(k-1)
};

__host__ __device__ inline sp_kparams sp_make_kparams(int k) {
    sp_kparams p;
    p.k = k;
    p.odd = k & 1;
    p.kmask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    p.rcshift = 2 * (k - 1);
    return p;
}

__host__ __device__ inline int64_t sp_dense_slots(int k) {
    return (k & 1) ? (1LL << (2 * k - 1)) : (1LL << (2 * k));
}

__host__ __device__ inline uint64_t sp_slot_of(uint64_t fwd, uint64_t rc, const sp_kparams &p) {
    if (p.odd) {
        uint64_t rep = ((fwd >> p.k) & 1ULL) ? rc : fwd;
        uint64_t lowmask = (1ULL << p.k) - 1ULL;
        return (rep & lowmask) | ((rep >> (p.k + 1)) << p.k);
    }
    return fwd < rc ? fwd : rc;
}

__host__ __device__ inline uint64_t sp_revcomp(uint64_t x, int k) {
    // complement then reverse 2-bit groups
    x = ~x;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = (x >> 32) | (x << 32);
    return x >> (64 - 2 * k);
}

// canonical key of a dense slot
__host__ __device__ inline uint64_t sp_key_of_slot(uint64_t slot, const sp_kparams &p) {
    if (p.odd) {
        uint64_t lowmask = (1ULL << p.k) - 1ULL;
        uint64_t rep = (slot & lowmask) | ((slot >> p.k) << (p.k + 1));
        uint64_t r = sp_revcomp(rep, p.k);
        return rep < r ? rep : r;
    }
    return slot;
}

// slot of a canonical (or any-orientation) key
__host__ __device__ inline uint64_t sp_slot_of_key(uint64_t key, const sp_kparams &p) {
    return sp_slot_of(key, sp_revcomp(key, p.k), p);
}
