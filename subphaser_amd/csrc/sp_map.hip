// sp_map.hip -- K4 (label table) + K5 (fused scan + label lookup + bin reduce).
//
// Replaces Seqs.map_kmer3 / chunk_chromfiles / map_kmer_each4
// (reference: subphaser/Seqs.py:74-153, 209-244): every k-mer START position
// whose canonical k-mer is subgenome-specific adds 1 to (bin of start, SG).
#include "sp_device.h"
#include "sp_map.h"

// ----------------------------------------------------------------- K4
// Two-level label lookup.  A random 1-byte gather from the 512-MiB label table
// runs at ~54 G lookups/s on MI355X while a gather from an L2-resident structure
// runs at ~250 G/s (profiles/r01_ubench_mi355x.txt).  Most genome positions do
// not carry a subgenome-specific k-mer, so positions are first screened by a
// blocked Bloom filter that fits each XCD's 4-MiB L2 and only positions that
// pass it touch the exact table.
//
// The filter is keyed on the (k-1)-mer that two neighbouring k-mers (starts
// 2i and 2i+1) SHARE, so ONE L2 probe screens two start positions (7 G probes
// instead of 14 G on the wheat-like genome; K5 106 -> 84 ms at 2^25 bits, 70 ms
// at 2^24 bits).  Every labelled k-mer inserts its prefix and its suffix
// (k-1)-mer in canonical form: if the k-mer at s is labelled, its suffix is in
// the filter; if the k-mer at s+1 is labelled, its prefix -- the same (k-1)-mer
// -- is.  Size: the smallest power of two whose measured fill stays below
// MAP_FILL_MAX, at most 2^25 bits (a 2-MiB filter leaves half of the L2 to the
// genome stream and the table lines; 2^26 bits does not fit and is slower than
// no growth).
// pair filter: prefix and suffix (k-1)-mers of every labelled canonical k-mer
__global__ void __launch_bounds__(256)
k4_pair_filter(const unsigned long long *__restrict__ keys, int64_t n, int k, uint32_t *__restrict__ bloom, int nbits) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    uint64_t pre = 0, suf = 0;          // k = 1: the shared 0-mer is the constant 0
    if (k > 1) {
        const uint64_t m1mask = (k - 1 >= 32) ? ~0ULL : ((1ULL << (2 * (k - 1))) - 1ULL);
        pre = key >> 2;
        suf = key & m1mask;
        const uint64_t rpre = sp_revcomp(pre, k - 1), rsuf = sp_revcomp(suf, k - 1);
        pre = pre < rpre ? pre : rpre;
        suf = suf < rsuf ? suf : rsuf;
    }
    const map_bloom_probe p = map_bloom(pre, k, nbits);
    atomicOr(&bloom[p.word], p.bits);
    const map_bloom_probe q = map_bloom(suf, k, nbits);
    atomicOr(&bloom[q.word], q.bits);
}

__global__ void __launch_bounds__(256)
k4_filter_fill(const uint32_t *__restrict__ bloom, int64_t n_words, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[16];
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x)
        c += __popc(bloom[i]);
    unsigned long long t = sp_block_sum_u64(c, red);
    if (threadIdx.x == 0 && t) atomicAdd(out, t);
}

__global__ void __launch_bounds__(256)
k4_labels(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ sg, int64_t n,
          sp_kparams kp, uint8_t *__restrict__ label) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    label[sp_slot_of_key(keys[i], kp)] = (uint8_t)(1u + sg[i]);
}

// ----------------------------------------------------------------- K4b: exact PAIR table
// The exact lookup that follows the filter used to be one random byte gather per candidate START from
// the 2^(2k-1)-byte label table -- an L2 miss each (54 G/s on this chip), 2.7 G of them on the wheat-like
// genome.  The pair table is keyed like the filter, on the canonical (k-1)-mer x two neighbouring starts
// share, and its 32-bit entry answers BOTH of them: eight 4-bit fields
//     field b     (b = 0..3)  label of the k-mer  b + x   ("L": x extended to the left by base b)
//     field 4 + b             label of the k-mer  x + b   ("R")
// in the orientation in which x is canonical; a field is (1 + SG) in its low three bits (0 = not a
// subgenome-specific k-mer) and a "seen" flag in bit 3.  One gather per candidate PAIR instead of one per
// candidate start: labelled k-mers come in runs (repeat copies), so almost every candidate pair carries
// two hits.  4^(k-1) entries (1 GiB at k = 15); S <= 7 (the label-table path below stays for S > 7).
__global__ void __launch_bounds__(256)
k4_pair_table(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ sg, int64_t n, int k,
              uint32_t *__restrict__ ptab) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    const uint32_t l = 1u + sg[i];
    // a strand pair has ONE prefix location and ONE suffix location (both orientations lead to the same
    // two fields); palindromic x is handled by the <= in map_pair_loc_* and the same <= in the lookup
    const map_pair_loc a = map_pair_loc_prefix(key, k), b = map_pair_loc_suffix(key, k);
    atomicOr(&ptab[a.idx], l << (4 * a.field));
    atomicOr(&ptab[b.idx], l << (4 * b.field));
    const uint64_t r = sp_revcomp(key, k);
    const map_pair_loc c = map_pair_loc_prefix(r, k), d = map_pair_loc_suffix(r, k);
    atomicOr(&ptab[c.idx], l << (4 * c.field));
    atomicOr(&ptab[d.idx], l << (4 * d.field));
}

// un-build: zero the (at most four) entries every key of the PREVIOUS label set touched.  The table is 4^(k-1)
// words (1 GiB at k = 15) of which a label set writes a few million: clearing it with a memset cost 3.4 ms per
// sp_labels_set call, this costs what k4_pair_table costs.
__global__ void __launch_bounds__(256)
k4_pair_clear(const unsigned long long *__restrict__ keys, int64_t n, int k, uint32_t *__restrict__ ptab) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i], r = sp_revcomp(key, k);
    ptab[map_pair_loc_prefix(key, k).idx] = 0u;
    ptab[map_pair_loc_suffix(key, k).idx] = 0u;
    ptab[map_pair_loc_prefix(r, k).idx] = 0u;
    ptab[map_pair_loc_suffix(r, k).idx] = 0u;
}

__global__ void __launch_bounds__(256)
k4_pair_seen(const unsigned long long *__restrict__ keys, int64_t n, int k, const uint32_t *__restrict__ ptab,
             unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[16];
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i], r = sp_revcomp(key, k);
        const map_pair_loc a = map_pair_loc_prefix(key, k), b = map_pair_loc_suffix(key, k);
        const map_pair_loc e = map_pair_loc_prefix(r, k), f = map_pair_loc_suffix(r, k);
        const uint32_t seen = ((ptab[a.idx] >> (4 * a.field)) | (ptab[b.idx] >> (4 * b.field)) |
                               (ptab[e.idx] >> (4 * e.field)) | (ptab[f.idx] >> (4 * f.field))) & 8u;
        c += seen ? 1 : 0;
    }
    unsigned long long t = sp_block_sum_u64(c, red);
    if (threadIdx.x == 0 && t) atomicAdd(out, t);
}

// ----------------------------------------------------------------- K4c: compact pair table (sp_map.h), S <= 3
// insert-or-find a key, then OR the field in.  The four entries of a bucket are tried in order and never freed, so two
// threads that insert the same key meet in the same entry; a bucket with four foreign entries raises its overflow flag
// and the key goes to the overflow table (linear probing; *fail is set when that table is full -- the host then falls
// back to the direct table).
__device__ __forceinline__ void map_ct_insert(const map_ptab &T, const map_ct_key &q, uint32_t bits, unsigned long long *fail) {
    uint32_t *e = reinterpret_cast<uint32_t *>(T.buckets + q.bucket);
    const uint32_t fresh = (q.tag << 25) | bits;
    // (the keys that share a (k-3)-mer share the bucket by design -- x1 and x2 of a run -- so every key starts at the
    // entry its own tag names instead of all of them queueing at entry 0: fewer failed compare-and-swaps beyond L2)
    for (int j = 0; j < 4; j++) {
        const int i = (int)((q.tag + (uint32_t)j) & 3u);
        const uint32_t old = atomicCAS(&e[i], 0u, fresh);
        if (old == 0u) return;
        if ((old >> 25) == q.tag && (old & MAP_CT_PAYLOAD)) {
            atomicOr(&e[i], bits);
            return;
        }
    }
    atomicOr(&e[0], MAP_CT_OVF);
    const unsigned long long mine = ((unsigned long long)(q.kid + 1u) << 32) | bits;
    uint32_t i = map_ct_ovf_home(q.kid) & T.ovf_mask;
    for (uint32_t probes = 0; probes <= T.ovf_mask; probes++) {
        const unsigned long long old = atomicCAS(&T.ovf[i], 0ULL, mine);
        if (old == 0ULL) {
            atomicAdd(fail + 1, 1ULL);      // (statistics: keys in the overflow table)
            return;
        }
        if ((uint32_t)(old >> 32) == q.kid + 1u) {
            atomicOr(&T.ovf[i], (unsigned long long)bits);
            return;
        }
        i = (i + 1u) & T.ovf_mask;
    }
    atomicAdd(fail, 1ULL);
}
__global__ void __launch_bounds__(256)
k4_ctab_build(const unsigned long long *__restrict__ keys, const uint8_t *__restrict__ sg, int64_t n, int k, map_ptab T,
              unsigned long long *__restrict__ fail) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t o2[2] = {keys[i], sp_revcomp(keys[i], k)};
    const uint32_t l = 1u + sg[i];
    // each (k-1)-mer / (k-3)-mer combination is canonical-as-read in exactly one of the two orientations (in both for a
    // palindromic (k-3)-mer; a palindromic k-mer, even k, enters the same fields twice: idempotent)
    for (int r = 0; r < 2; r++) {
        map_ct_site st[4];
        const int ns = map_ct_sites(o2[r], k, st);
        for (int j = 0; j < ns; j++)
            map_ct_insert(T, map_ct_key_of(T, st[j].t, st[j].side, st[j].e), l << (MAP_CT_FIELD * st[j].field), fail);
    }
}
__global__ void __launch_bounds__(256)
k4_ctab_seen(const unsigned long long *__restrict__ keys, int64_t n, int k, map_ptab T, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[16];
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t o2[2] = {keys[i], sp_revcomp(keys[i], k)};
        uint32_t seen = 0;
        for (int r = 0; r < 2; r++) {
            map_ct_site st[4];
            const int ns = map_ct_sites(o2[r], k, st);
            for (int j = 0; j < ns; j++) {
                const map_ct_key q = map_ct_key_of(T, st[j].t, st[j].side, st[j].e);
                const uint4 B = T.buckets[q.bucket];
                seen |= (map_ct_find(T, B, q).fields >> (MAP_CT_FIELD * st[j].field)) & 4u;
            }
        }
        c += seen ? 1 : 0;
    }
    unsigned long long t = sp_block_sum_u64(c, red);
    if (threadIdx.x == 0 && t) atomicAdd(out, t);
}

// K5, pair-table engine (S <= 7): one block-iteration covers MAP_BLOCK units of 64 starts.  ONE launch maps every
// chromosome of the call: ranges are numbered through the whole genome (desc[c].range0 = first range of chromosome
// c) and dealt to the blocks round-robin, so the tail of the launch is one range per block instead of one partly
// filled round of blocks per chromosome (21 launches of 3.3 rounds each on the wheat-like genome).
struct map_chrom_desc {
    const uint32_t *pk, *pm, *nm;
    int64_t n_units;       // units of 64 starts
    int64_t nslots;        // output slots (bins + chunk duplicates) of this chromosome
    int64_t range0;        // number of ranges of the chromosomes before it
    int *counts;           // [nslots x S]
    unsigned long long *n_mapped;
};

#ifndef MAP_MIN_WAVES
#define MAP_MIN_WAVES 8     // waves per SIMD the register allocation must leave room for (four 512-thread workgroups per CU)
#endif
// ----------------------------------------------------------------- K5: the rolled walk (round 5)
// Rounds 2-4 unrolled the walk over a unit completely: 258 KB of machine code (the k > 15 twin: 532 KB) -- sixteen
// copies of the quad walk, each with four inlined copies of the hit path (64-bit divisions for the output slot, an LDS
// flush loop, the overflow probe loop) -- against 64 KB of instruction cache: every wave streamed the whole kernel from
// the L2 once per 64 starts, ~14 G instruction-line requests per wheat-like pass next to the 7 G filter probes the
// kernel was believed to be bound by, and the reason why no change to its memory accesses ever moved it (quad buckets:
// 1.55 G -> 0.96 G look-ups, same 45.9 ms; more loads in flight per lane: slower).  Those kernels were removed in round
// 6.  k5_map2 keeps the loop over the quads of a
// unit ROLLED (windows by run-time shifts out of a rotating pair of registers per stream), records a hit as a bit in
// three 64-bit label planes instead of calling into the slot arithmetic, and settles a unit's hits once: popcounts
// into the range's LDS histogram when the whole range lies in one output slot run (the common case; the boundaries
// are found once per range), the general walk otherwise.  TABLE: 1 = compact quad buckets (S <= 3), 0 = direct pair
// table (S <= 7).
// Round 6, second half: the walk over a unit has TWO phases, and the second is shared out over the wave.  PMC counters put
// k5_map2 at 70 % VALU utilisation (15.9 G wave instructions per wheat-like pass = 289 per quad and lane), half of them in the
// candidate path -- bucket key, tag compares, label fields, planes -- which every wave executed for every quad, because SOME lane
// of 64 is a candidate (27 % each; inside a repeat copy all sixteen quads of a lane are, so compacting per lane gains nothing:
// measured, same instruction count).  Phase 1 is the filter alone: it leaves the candidate quads of the WAVE in an LDS queue
// (lane, quad, which pairs).  Phase 2 deals the queue out to all lanes: ~280 candidates per 1024 quads are 4-5 iterations of the
// candidate path per unit instead of 16.  A lane rebuilds the windows of the quad it was dealt from the owner's ten packed words,
// parked in LDS (one column per thread), and ORs the labels it finds into the owner's planes there.
#ifndef MAP2_QI
#define MAP2_QI 1      // quads per iteration of the filter phase
#endif
struct map_quad_win {
    uint32_t V1, V2, xf1, xr1, xf2, xr2;       // 16-base windows at j / j + 2 (MSB-first), x1 / x2 forward and reverse complement
};
__device__ __forceinline__ map_quad_win map_quad_windows(uint32_t l0, uint32_t l1, uint32_t m0, uint32_t m1, int r /* 0, 4, 8, 12 */,
                                                         int sh, uint32_t m1mask) {
    map_quad_win q;
    const uint32_t W1 = __builtin_amdgcn_alignbit(l1, l0, 2 * r), W2 = __builtin_amdgcn_alignbit(l1, l0, 2 * r + 4);
    const unsigned long long mm = ((unsigned long long)m0 << 32) | m1;
    q.V1 = (uint32_t)((mm << (2 * r)) >> 32);
    q.V2 = (uint32_t)((mm << (2 * r + 4)) >> 32);
    q.xf1 = (q.V1 >> sh) & m1mask;       // x1 forward / reverse complement, key order
    q.xr1 = (~W1 >> 2) & m1mask;
    q.xf2 = (q.V2 >> sh) & m1mask;
    q.xr2 = (~W2 >> 2) & m1mask;
    return q;
}
template <int TABLE>
__device__ __forceinline__ void map_unit_scan64(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm,
                                                const uint32_t *__restrict__ nm, int64_t s0, const sp_kparams32 &kp,
                                                const uint32_t *__restrict__ bloom, int nbits, const map_ptab &T,
                                                unsigned long long lab[3], const map_unit_lds &U,
                                                unsigned long long cm = ~0ULL /* starts that count */) {
    constexpr int FW = TABLE ? MAP_CT_FIELD : 4;
    constexpr uint32_t LBL = TABLE ? 3u : 7u, SEEN = TABLE ? 4u : 8u, FMASK = TABLE ? 7u : 15u;
    constexpr uint32_t ANY = TABLE ? MAP_CT_ANY : 0x77777777u;
    constexpr int NP = TABLE ? 2 : 3, ROUND_W = TABLE ? MAP2_ROUND_W : 2;
    // validity of the 64 starts: k-mer at s0+j, shared (k-1)-mer at s0+j+1
    unsigned long long ok_k, ok_x;
    {
        const uint64_t badA = sp_bad_starts64(nm, s0, kp.k - 1), badB = sp_bad_starts64(nm, s0 + 32, kp.k - 1);
        const uint64_t invA = (uint64_t)nm[s0 >> 5] | ((uint64_t)nm[(s0 >> 5) + 1] << 32);
        const uint64_t invB = (uint64_t)nm[(s0 >> 5) + 1] | ((uint64_t)nm[(s0 >> 5) + 2] << 32);
        const uint32_t kA = ~(uint32_t)(badA | (invA >> (kp.k - 1))), kB = ~(uint32_t)(badB | (invB >> (kp.k - 1)));
        const uint32_t xA = ~(uint32_t)(badA >> 1), xB = ~(uint32_t)(badB >> 1);
        ok_k = (unsigned long long)kA | ((unsigned long long)kB << 32);
        ok_x = (unsigned long long)xA | ((unsigned long long)xB << 32);
    }
    if (__all((ok_x & 0x5555555555555555ULL) == 0)) return;
    // the lanes of the wave that are here (a range's last wave, a grid-stride loop's last round): they share the candidates
    const unsigned long long here = __ballot(1);
    const int n_here = __popcll(here);
    const int me = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
    const int tid = (int)threadIdx.x, lane = tid & 63, wave0 = tid & ~63;
    uint32_t *sw = U.words + tid;
    uint16_t *queue = U.queue + (tid >> 6) * MAP_QCAP(TABLE);
    const int64_t w0 = s0 >> 4;   // a multiple of 4: 16-byte aligned
    const uint4 la = *reinterpret_cast<const uint4 *>(pk + w0);
    uint32_t l0 = la.x, l1 = la.y, l2 = la.z, l3 = la.w, l4 = pk[w0 + 4];
#if SP_DERIVE_PM
    (void)pm;
    uint32_t m0 = sp_msb_of_lsb(l0), m1 = sp_msb_of_lsb(l1), m2 = sp_msb_of_lsb(l2), m3 = sp_msb_of_lsb(l3), m4 = sp_msb_of_lsb(l4);
#else
    const uint4 ma = *reinterpret_cast<const uint4 *>(pm + w0);
    uint32_t m0 = ma.x, m1 = ma.y, m2 = ma.z, m3 = ma.w, m4 = pm[w0 + 4];
#endif
    {
        const unsigned long long okc = ok_k & cm;      // (a start outside `cm` -- interval mode: not covered by a feature -- is neither counted nor marked seen)
        sw[0 * MAP_BLOCK] = l0; sw[1 * MAP_BLOCK] = l1; sw[2 * MAP_BLOCK] = l2; sw[3 * MAP_BLOCK] = l3; sw[4 * MAP_BLOCK] = l4;
        sw[5 * MAP_BLOCK] = (uint32_t)okc; sw[6 * MAP_BLOCK] = (uint32_t)(okc >> 32);
#pragma unroll
        for (int bit = 0; bit < NP; bit++) U.planes[bit * MAP_BLOCK + tid] = 0ULL;
    }
    const int sh = 32 - 2 * kp.k, sh1 = 30 - 2 * kp.k;
    const uint32_t m1mask = kp.kmask >> 2;
    const int sb = T.sb;
    const uint32_t smask = TABLE ? ((1u << sb) - 1u) : 0u;
    // the filter's addressing (sp_map.h): by the smaller-hashed core of the (k-1)-mer, or by the (k-1)-mer itself
    const bool core = (nbits & MAP_BLOOM_CORE) && kp.k >= 5;      // (uniform)
    const uint32_t cmask = kp.k >= 5 ? ((1u << (2 * (kp.k - 3))) - 1u) : 0u;
    const int wsh = 32 - (MAP_BLOOM_NBITS(nbits) - 5);
    uint32_t last_wi = 0xFFFFFFFFu, last_w = 0u;                  // the word this lane fetched last (index, content)
    uint32_t h_carry = 0u;                                         // hash of the core the previous quad ended with
#pragma unroll 1
    for (int wr = 0; wr < 4; wr += ROUND_W) {
        // ---- phase 1: the filter over ROUND_W words of sixteen starts; the wave's candidate quads go to its queue
        uint32_t qn = 0;                                           // (uniform)
#pragma unroll 1
        for (int w = wr; w < wr + ROUND_W; w++) {
#pragma unroll 1
            for (int r0 = 0; r0 < 16; r0 += 4 * MAP2_QI) {
                // MAP2_QI quads per iteration: their filter words are requested together (the word a pair re-uses is known from the
                // indices alone), then looked at
                uint32_t bt1[MAP2_QI], bt2[MAP2_QI], wi1[MAP2_QI], wi2[MAP2_QI], f1[MAP2_QI], f2[MAP2_QI];
                bool v1[MAP2_QI], v2[MAP2_QI], need1[MAP2_QI], need2[MAP2_QI];
                uint32_t lw = last_wi;
#pragma unroll
                for (int q = 0; q < MAP2_QI; q++) {
                    const int r = r0 + 4 * q, j = 16 * w + r;     // the quad's first start; its pairs share x1 (at j + 1) and x2 (at j + 3)
                    const map_quad_win Q = map_quad_windows(l0, l1, m0, m1, r, sh, m1mask);
                    const uint32_t c1 = Q.xf1 < Q.xr1 ? Q.xf1 : Q.xr1, c2 = Q.xf2 < Q.xr2 ? Q.xf2 : Q.xr2;
                    uint32_t h1, h2;
                    bt1[q] = map_bloom_bits3((uint64_t)c1, h1);
                    bt2[q] = map_bloom_bits3((uint64_t)c2, h2);
                    const uint32_t okx = (uint32_t)(ok_x >> j);
                    v1[q] = okx & 1u;
                    v2[q] = okx & 4u;
                    if (core) {
                        // the chain of cores: a = first k-3 bases of x1, b = last of x1 = first of x2, c = last of x2 (= the next
                        // quad's a); canonical = the smaller of the forward reading and its reverse complement, which is the
                        // OTHER end of the reverse-complemented (k-1)-mer
                        const uint32_t tb_f = Q.xf1 & cmask, tb_r = Q.xr1 >> 4, tc_f = Q.xf2 & cmask, tc_r = Q.xr2 >> 4;
                        uint32_t ha = h_carry;              // (this quad's a IS the previous quad's c: the same bases)
                        if (j == 0) {                       // (uniform: the unit's first quad)
                            const uint32_t ta_f = Q.xf1 >> 4, ta_r = Q.xr1 & cmask;
                            ha = map_core_hash((uint64_t)(ta_f < ta_r ? ta_f : ta_r));
                        }
                        const uint32_t hb = map_core_hash((uint64_t)(tb_f < tb_r ? tb_f : tb_r));
                        const uint32_t hc = map_core_hash((uint64_t)(tc_f < tc_r ? tc_f : tc_r));
                        h_carry = hc;
                        wi1[q] = map_core_word(ha < hb ? ha : hb, nbits);
                        wi2[q] = map_core_word(hb < hc ? hb : hc, nbits);
                    } else {
                        wi1[q] = h1 >> wsh;
                        wi2[q] = h2 >> wsh;
                    }
                    // a word this lane holds already (the previous pair's) is not fetched again: with the core addressing that
                    // is the case for a third of the pairs -- and a gather is priced per active lane
                    need1[q] = v1[q] && wi1[q] != lw;
                    need2[q] = v2[q] && wi2[q] != (v1[q] ? wi1[q] : lw);
                    f1[q] = f2[q] = 0;
                    if (need1[q]) f1[q] = bloom[wi1[q]];
                    if (need2[q]) f2[q] = bloom[wi2[q]];
                    if (v2[q]) lw = wi2[q];
                    else if (v1[q]) lw = wi1[q];
                }
                last_wi = lw;
#pragma unroll
                for (int q = 0; q < MAP2_QI; q++) {
                    const int j = 16 * w + r0 + 4 * q;
                    const uint32_t wd1 = v1[q] ? (need1[q] ? f1[q] : last_w) : 0u;
                    const uint32_t wd2 = v2[q] ? (need2[q] ? f2[q] : (v1[q] ? wd1 : last_w)) : 0u;
                    if (v2[q]) last_w = wd2;
                    else if (v1[q]) last_w = wd1;
                    const bool cand1 = (wd1 & bt1[q]) == bt1[q], cand2 = (wd2 & bt2[q]) == bt2[q];       // (an invalid pair's word is 0: never a candidate)
                    const unsigned long long bal = __ballot(cand1 || cand2);
                    if (cand1 || cand2)
                        queue[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] =
                            (uint16_t)((uint32_t)lane | ((uint32_t)(j >> 2) << 6) | (cand1 ? 0x400u : 0u) | (cand2 ? 0x800u : 0u));
                    qn += (uint32_t)__popcll(bal);
                }
            }
            l0 = l1; l1 = l2; l2 = l3; l3 = l4;
            m0 = m1; m1 = m2; m2 = m3; m3 = m4;
        }
        // ---- phase 2: the queue, dealt out to the lanes that are here (LDS operations of a wave complete in program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t e0 = (uint32_t)me; e0 < qn; e0 += (uint32_t)(MAP2_P2 * n_here)) {
            // MAP2_P2 entries per lane and iteration: their bucket loads (Infinity Cache: the longest wait of the kernel) travel together
            bool live[MAP2_P2], cand1[MAP2_P2], cand2[MAP2_P2], sfw[MAP2_P2];
            int owner[MAP2_P2], jj[MAP2_P2];
            map_quad_win Q[MAP2_P2];
            uint32_t okk[MAP2_P2], e1[MAP2_P2], e2[MAP2_P2];
            map_ct_key k1[MAP2_P2], k2[MAP2_P2];
            uint4 B[MAP2_P2];
#pragma unroll
            for (int p = 0; p < MAP2_P2; p++) {
                const uint32_t e = e0 + (uint32_t)(p * n_here);
                live[p] = e < qn;
                const uint32_t ent = live[p] ? (uint32_t)queue[e] : 0u;
                const int qi = (int)((ent >> 6) & 15u), w = qi >> 2;
                owner[p] = wave0 + (int)(ent & 63u);
                jj[p] = 4 * qi;
                cand1[p] = ent & 0x400u;
                cand2[p] = ent & 0x800u;
                const uint32_t *ow = U.words + owner[p];
                const uint32_t ola = ow[w * MAP_BLOCK], olb = ow[(w + 1) * MAP_BLOCK];
                Q[p] = map_quad_windows(ola, olb, sp_msb_of_lsb(ola), sp_msb_of_lsb(olb), jj[p] & 15, sh, m1mask);
                okk[p] = ow[(5 + (w >> 1)) * MAP_BLOCK] >> (jj[p] & 31);      // countable starts j .. j + 3 of the owner's unit
                e1[p] = e2[p] = 0;
                sfw[p] = false;
                B[p] = make_uint4(0u, 0u, 0u, 0u);
                if (TABLE) {
                    // s = the last k-3 bases of x1 = the first k-3 bases of x2 (a candidate's (k-1)-mer is valid, so s is)
                    const uint32_t s_f = cand1[p] ? (Q[p].xf1 & smask) : (Q[p].xf2 >> 4), s_r = cand1[p] ? (Q[p].xr1 >> 4) : (Q[p].xr2 & smask);
                    sfw[p] = s_f <= s_r;
                    const uint32_t t = sfw[p] ? s_f : s_r;
                    // x1 = e + s: side L read forward, side R (e reverse-complemented) read backward; x2 = s + e: the mirror image
                    k1[p] = map_ct_key_of(T, t, sfw[p] ? 0u : 1u, sfw[p] ? (Q[p].xf1 >> sb) : (Q[p].xr1 & 15u));
                    k2[p] = map_ct_key_of(T, t, sfw[p] ? 1u : 0u, sfw[p] ? (Q[p].xf2 & 15u) : (Q[p].xr2 >> sb));
                    if (live[p]) B[p] = T.buckets[k1[p].bucket];
                } else {
                    if (cand1[p]) e1[p] = T.direct[Q[p].xf1 < Q[p].xr1 ? Q[p].xf1 : Q[p].xr1];
                    if (cand2[p]) e2[p] = T.direct[Q[p].xf2 < Q[p].xr2 ? Q[p].xf2 : Q[p].xr2];
                }
            }
#pragma unroll
            for (int p = 0; p < MAP2_P2; p++) {
                if (!live[p]) continue;
                const int j = jj[p];
                uint32_t loc1 = Q[p].xf1 < Q[p].xr1 ? Q[p].xf1 : Q[p].xr1, loc2 = Q[p].xf2 < Q[p].xr2 ? Q[p].xf2 : Q[p].xr2;
                bool fw1 = Q[p].xf1 <= Q[p].xr1, fw2 = Q[p].xf2 <= Q[p].xr2;
                if (TABLE) {
                    if (cand1[p]) {
                        const map_ct_hit h = map_ct_find(T, B[p], k1[p]);
                        e1[p] = h.fields;
                        loc1 = h.loc;
                    }
                    if (cand2[p]) {
                        const map_ct_hit h = map_ct_find(T, B[p], k2[p]);
                        e2[p] = h.fields;
                        loc2 = h.loc;
                    }
                    fw1 = fw2 = sfw[p];       // the fields are laid out in the orientation in which t is canonical
                }
                // the (up to four) labelled starts of the quad: start j + 2h = b0 + x, start j + 2h + 1 = x + b1
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t ee = h ? e2[p] : e1[p];
                    if (!(ee & ANY)) continue;
                    const uint32_t V = h ? Q[p].V2 : Q[p].V1;
                    const bool fw = h ? fw2 : fw1;
                    const uint32_t b0 = V >> 30, b1 = (V >> sh1) & 3u;
                    const int f0 = fw ? (int)b0 : 7 - (int)b0, f1 = fw ? 4 + (int)b1 : 3 - (int)b1;
                    const uint32_t cnt2 = okk[p] >> (2 * h);
                    const uint32_t v0 = (cnt2 & 1u) ? (ee >> (FW * f0)) & FMASK : 0u;
                    const uint32_t v1 = (cnt2 & 2u) ? (ee >> (FW * f1)) & FMASK : 0u;
                    const uint32_t two = (v0 & LBL) | ((v1 & LBL) << 8);      // labels of the two starts
                    if (!two) continue;
#pragma unroll
                    for (int bit = 0; bit < NP; bit++) {
                        const unsigned long long b2 = (unsigned long long)(((two >> bit) & 1u) | (((two >> (8 + bit)) & 1u) << 1));
                        if (b2) atomicOr(&U.planes[bit * MAP_BLOCK + owner[p]], b2 << (j + 2 * h));
                    }
                    uint32_t mark = 0;       // "seen": first touch only
                    if ((v0 & LBL) && !(v0 & SEEN)) mark |= SEEN << (FW * f0);
                    if ((v1 & LBL) && !(v1 & SEEN)) mark |= SEEN << (FW * f1);
                    if (mark) {
                        if (TABLE) map_ct_mark(T, h ? loc2 : loc1, mark);
                        else atomicOr(&T.direct[h ? loc2 : loc1], mark);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int bit = 0; bit < NP; bit++) lab[bit] = U.planes[bit * MAP_BLOCK + tid];
}

template <int TABLE>
__global__ void __launch_bounds__(MAP_BLOCK, MAP_MIN_WAVES)
k5_map2(const map_chrom_desc *__restrict__ desc, int n_chrom, int64_t n_ranges, sp_kparams32 kp, sp_map_params P,
        map_ptab ptab, const uint32_t *__restrict__ bloom, int bloom_bits) {
    __shared__ int hist[MAP_LDS_ENTRIES];
    __shared__ unsigned long long red[16];
    MAP_UNIT_LDS_DECL(TABLE, 7);
    unsigned long long mapped = 0;
    int cur = -1;          // chromosome the block is accumulating `mapped` for
    auto flush_mapped = [&]() {     // block-uniform control flow
        if (cur < 0) return;
        const unsigned long long t = sp_block_sum_u64(mapped, red);
        if (threadIdx.x == 0 && t) atomicAdd(desc[cur].n_mapped, t);
        mapped = 0;
    };
    for (int64_t rg = blockIdx.x; rg < n_ranges; rg += gridDim.x) {
        int lo = 0, hi = n_chrom;              // last chromosome with range0 <= rg (uniform: scalar loads)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (desc[mid].range0 <= rg) lo = mid;
            else hi = mid;
        }
        if (lo != cur) {
            flush_mapped();
            cur = lo;
        }
        const map_chrom_desc D = desc[lo];
        const int64_t r = rg - D.range0;
        const int64_t u = r * MAP_BLOCK + threadIdx.x;
        // the range's first output slot and where it ends (uniform; once per range, not per hit)
        const int64_t slot_lo = map_slot(r * MAP_RANGE, P, kp.k), end_lo = map_slot_end(r * MAP_RANGE, P, kp.k);
        const bool one_slot = end_lo >= (r + 1) * MAP_RANGE;
        if (P.use_lds) {
            for (int i = threadIdx.x; i < MAP_LDS_ENTRIES; i += MAP_BLOCK) hist[i] = 0;
            __syncthreads();
        }
        if (u < D.n_units) {
            unsigned long long lab[3] = {0ULL, 0ULL, 0ULL};
            map_unit_scan64<TABLE>(D.pk, D.pm, D.nm, u * SP_UNIT, kp, bloom, bloom_bits, ptab, lab, ulds);
            if (lab[0] | lab[1] | lab[2]) {
                auto add = [&](int64_t os, unsigned long long within) {       // the unit's hits among the starts `within` -> slot os
                    for (int sg = 0; sg < P.S; sg++) {
                        const int l = sg + 1;
                        const unsigned long long m = ((l & 1) ? lab[0] : ~lab[0]) & ((l & 2) ? lab[1] : ~lab[1]) &
                                                     ((l & 4) ? lab[2] : ~lab[2]) & within;
                        const int v = __popcll(m);
                        if (!v) continue;
                        if (P.use_lds) atomicAdd(&hist[(int)(os - slot_lo) * P.S + sg], v);
                        else if (os < D.nslots) atomicAdd(&D.counts[os * P.S + sg], v);
                        mapped += v;
                    }
                };
                const int64_t s0 = u * SP_UNIT;
                if (one_slot) {
                    add(slot_lo, ~0ULL);
                } else {
                    int64_t p = s0;
                    while (p < s0 + SP_UNIT) {
                        int64_t e = map_slot_end(p, P, kp.k);
                        if (e > s0 + SP_UNIT) e = s0 + SP_UNIT;
                        const int a = (int)(p - s0), b = (int)(e - s0);      // starts [a, b) of the unit
                        const unsigned long long within = (b >= 64 ? ~0ULL : ((1ULL << b) - 1ULL)) & ~((1ULL << a) - 1ULL);
                        if ((lab[0] | lab[1] | lab[2]) & within) add(map_slot(p, P, kp.k), within);
                        p = e;
                    }
                }
            }
        }
        if (P.use_lds) {
            __syncthreads();
            for (int i = threadIdx.x; i < MAP_LDS_ENTRIES; i += MAP_BLOCK) {
                int v = hist[i];
                if (v) {
                    int64_t os = slot_lo + i / P.S;
                    if (os < D.nslots) atomicAdd(&D.counts[os * P.S + (i % P.S)], v);
                }
            }
            __syncthreads();
        }
    }
    flush_mapped();
}

// interval (BED) mode on the rolled walk: the label planes of a unit, restricted to the covered starts, ARE its masks
template <int TABLE>
__global__ void __launch_bounds__(MAP_BLOCK, MAP_MIN_WAVES)
k5_map_mask2(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm, sp_kparams32 kp,
             int64_t n_units, int S, map_ptab ptab, const uint32_t *__restrict__ bloom, int bloom_bits,
             const unsigned long long *__restrict__ cov, unsigned long long *__restrict__ masks /* n_units x S */) {
    MAP_UNIT_LDS_DECL(TABLE, 7);
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; u < n_units; u += stride) {
        const unsigned long long cv = cov[u];
        if (__all(cv == 0ULL)) continue;         // nothing of this wave's 4096 starts lies in a feature
        unsigned long long lab[3] = {0ULL, 0ULL, 0ULL};
        map_unit_scan64<TABLE>(pk, pm, nm, u * SP_UNIT, kp, bloom, bloom_bits, ptab, lab, ulds, cv);
        for (int sg = 0; sg < S; sg++) {
            const int l = sg + 1;
            masks[u * S + sg] = ((l & 1) ? lab[0] : ~lab[0]) & ((l & 2) ? lab[1] : ~lab[1]) & ((l & 4) ? lab[2] : ~lab[2]) & cv;
        }
    }
}

// the exact pair table the current label set lives in (compact when ctx->ct_bb != 0)
static map_ptab map_ptab_of(const sp_ctx *ctx) {
    map_ptab T;
    T.direct = ctx->ct_bb ? nullptr : (uint32_t *)ctx->b_ptab.p;
    T.buckets = ctx->ct_bb ? (uint4 *)ctx->b_ctab.p : nullptr;
    T.ovf = (unsigned long long *)ctx->b_covf.p;
    T.ovf_mask = ctx->ct_ovf_mask;
    T.sb = 2 * (ctx->k - 3);
    T.tb = ctx->ct_bb ? T.sb - ctx->ct_bb : 0;
    return T;
}

// descriptors of the chromosomes [first, first + n) -> device, then the one launch
static int map_launch_dense(sp_ctx *ctx, const std::vector<map_chrom_desc> &hd, int64_t n_ranges, const sp_map_params &P) {
    if (n_ranges <= 0) return SP_OK;
    int rcb = sp_buf_ensure(ctx, ctx->b_mapdesc, (int64_t)(hd.size() * sizeof(map_chrom_desc)));
    if (rcb) return rcb;
    // (pageable source: hipMemcpyAsync returns once the runtime has staged it, `hd` may die afterwards)
    SP_HIP(ctx, hipMemcpyAsync(ctx->b_mapdesc.p, hd.data(), hd.size() * sizeof(map_chrom_desc), hipMemcpyHostToDevice,
                              ctx->stream));
    const sp_kparams32 kp = sp_make_kparams32(ctx->k);
    int64_t grid = n_ranges;
    if (grid > (int64_t)ctx->n_cu * MAP_GRID_MULT) grid = (int64_t)ctx->n_cu * MAP_GRID_MULT;
    const map_ptab T = map_ptab_of(ctx);
    if (T.buckets)
        SP_LAUNCH(ctx, "k5_map", k5_map2<1>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, (const map_chrom_desc *)ctx->b_mapdesc.p,
                  (int)hd.size(), n_ranges, kp, P, T, (const uint32_t *)ctx->d_bloom, ctx->bloom_bits);
    else
        SP_LAUNCH(ctx, "k5_map", k5_map2<0>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, (const map_chrom_desc *)ctx->b_mapdesc.p,
                  (int)hd.size(), n_ranges, kp, P, T, (const uint32_t *)ctx->d_bloom, ctx->bloom_bits);
    return SP_OK;
}

// K5, label-table engine (any S <= 126; the rolling scan + one byte gather per candidate start)
__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_lab(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, sp_kparams32 kp,
       sp_map_params P, uint8_t *__restrict__ label, const uint32_t *__restrict__ bloom, int bloom_bits,
       int *__restrict__ slot_counts, unsigned long long *__restrict__ n_mapped) {
    __shared__ int hist[MAP_LDS_ENTRIES];
    __shared__ unsigned long long red[16];
    unsigned long long mapped = 0;
    const int64_t n_ranges = (P.n_units + MAP_BLOCK - 1) / MAP_BLOCK;
    for (int64_t r = blockIdx.x; r < n_ranges; r += gridDim.x) {
        const int64_t u = r * MAP_BLOCK + threadIdx.x;
        const int64_t slot_lo = map_slot(r * MAP_RANGE, P, kp.k);
        if (P.use_lds) {
            for (int i = threadIdx.x; i < MAP_LDS_ENTRIES; i += MAP_BLOCK) hist[i] = 0;
            __syncthreads();
        }
        if (u < P.n_units) {
            map_pair_scan<uint32_t>(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, [&](int64_t start, uint32_t fwd, uint32_t rc) {
                const uint32_t slot = sp_slot_of32(fwd, rc, kp);
                const uint32_t l = label[slot];
                if (l) {
                    if (!(l & 0x80u)) label[slot] = (uint8_t)(l | 0x80u);  // idempotent "seen" mark
                    const int sg = (int)(l & 0x7fu) - 1;
                    const int64_t os = map_slot(start, P, kp.k);
                    if (P.use_lds)
                        atomicAdd(&hist[(os - slot_lo) * P.S + sg], 1);
                    else if (os < P.nslots)
                        atomicAdd(&slot_counts[os * P.S + sg], 1);
                    mapped++;
                }
            });
        }
        if (P.use_lds) {
            __syncthreads();
            for (int i = threadIdx.x; i < MAP_LDS_ENTRIES; i += MAP_BLOCK) {
                int v = hist[i];
                if (v) {
                    int64_t os = slot_lo + i / P.S;
                    if (os < P.nslots) atomicAdd(&slot_counts[os * P.S + (i % P.S)], v);
                }
            }
            __syncthreads();
        }
    }
    unsigned long long t = sp_block_sum_u64(mapped, red);
    if (threadIdx.x == 0 && t) atomicAdd(n_mapped, t);
}

// Window stack on the device (Circos.stack_matrix semantics, Circos.py:734-742): the window of a
// slot is (bin start) / window_size, bin = slot - chunk(slot); the two slots of a chunk-boundary bin
// land in the same window.
__global__ void __launch_bounds__(256)
k5_stack(const int *__restrict__ slot_counts, int64_t total_slots, int S, int C,
         const long long *__restrict__ slot_off, const long long *__restrict__ win_off,
         const long long *__restrict__ seg_start /* position of local base 0 inside its chromosome, or NULL */,
         int64_t bin_size, int64_t chunk_size, int64_t window_size, int k, unsigned long long *__restrict__ win_counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_slots * S) return;
    const int v = slot_counts[i];
    if (!v) return;
    const int64_t gslot = i / S;
    const int sg = (int)(i - gslot * S);
    int lo = 0, hi = C;  // chromosome of the slot
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (slot_off[mid] <= gslot) lo = mid;
        else hi = mid;
    }
    const int64_t slot = gslot - slot_off[lo];
    int64_t chunk = 0;
    if (chunk_size > 0) {
        // first slot of chunk j >= 1: (j*W - (k-1)) / b + j ; chunk(slot) = #{j : first(j) <= slot}
        chunk = slot * bin_size / (chunk_size + bin_size);
        while ((((chunk + 1) * chunk_size - (k - 1)) / bin_size + (chunk + 1)) <= slot) chunk++;
        while (chunk > 0 && ((chunk * chunk_size - (k - 1)) / bin_size + chunk) > slot) chunk--;
    }
    const int64_t win = ((seg_start ? seg_start[lo] : 0) + (slot - chunk) * bin_size) / window_size;
    atomicAdd(&win_counts[(win_off[lo] + win) * S + sg], (unsigned long long)v);
}

// Feature mode on the rolled walk (round 6; it ran on the unrolled walk of rounds 2-4 until then).  The features lie back to
// back in one packed sequence; a k-mer belongs to feature f iff it lies entirely inside [foff[f], foff[f + 1]).  Per unit of
// 64 starts: (1) the starts whose k-mer crosses no feature boundary (`fit`) -- only those are scanned, counted and marked
// "seen", exactly the k-mers a FASTA of the features holds; (2) the label planes of the unit; (3) one walk over the features
// that overlap the unit: popcounts of the planes inside the feature -> counts[f][sg].
template <int TABLE>
__global__ void __launch_bounds__(MAP_BLOCK, MAP_MIN_WAVES)
k5_map_feat2(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ pm, const uint32_t *__restrict__ nm,
             sp_kparams32 kp, int64_t n_units, const int64_t *__restrict__ foff, int64_t n_feat, int S,
             map_ptab ptab, const uint32_t *__restrict__ bloom, int bloom_bits,
             unsigned long long *__restrict__ counts) {
    MAP_UNIT_LDS_DECL(TABLE, 7);
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int k = kp.k;
    auto span = [](int64_t a, int64_t b) -> unsigned long long {      // bits [a, b) of a unit, 0 <= a, b <= 64
        if (b <= a) return 0ULL;
        return (b >= 64 ? ~0ULL : ((1ULL << b) - 1ULL)) & ~((1ULL << a) - 1ULL);
    };
    for (; u < n_units; u += stride) {
        const int64_t s0 = u * SP_UNIT;
        int64_t lo = 0, hi = n_feat;           // the feature the unit starts in: last f with foff[f] <= s0
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (foff[mid] <= s0) lo = mid;
            else hi = mid;
        }
        // (1) a boundary at `e` = foff[f + 1] disqualifies the starts (e - k, e): their k-mers run into the next feature
        unsigned long long fit = ~0ULL;
        for (int64_t f = lo; f < n_feat; f++) {
            const int64_t e = foff[f + 1];
            if (e - k + 1 >= s0 + SP_UNIT) break;
            fit &= ~span((e - k + 1 > s0 ? e - k + 1 : s0) - s0, (e < s0 + SP_UNIT ? e : s0 + SP_UNIT) - s0);
            if (e >= s0 + SP_UNIT) break;
        }
        unsigned long long lab[3] = {0ULL, 0ULL, 0ULL};
        map_unit_scan64<TABLE>(pk, pm, nm, s0, kp, bloom, bloom_bits, ptab, lab, ulds, fit);
        if (!(lab[0] | lab[1] | lab[2])) continue;
        // (3) the features that overlap the unit (a start beyond the last feature's end is padding: never valid)
        for (int64_t f = lo; f < n_feat; f++) {
            const int64_t a = foff[f], e = foff[f + 1];
            if (a >= s0 + SP_UNIT) break;
            const unsigned long long within = span((a > s0 ? a : s0) - s0, (e < s0 + SP_UNIT ? e : s0 + SP_UNIT) - s0) & fit;
            if (!((lab[0] | lab[1] | lab[2]) & within)) continue;
            for (int sg = 0; sg < S; sg++) {
                const int l = sg + 1;
                const unsigned long long m = ((l & 1) ? lab[0] : ~lab[0]) & ((l & 2) ? lab[1] : ~lab[1]) &
                                             ((l & 4) ? lab[2] : ~lab[2]) & within;
                if (m) atomicAdd(&counts[f * S + sg], (unsigned long long)__popcll(m));
            }
        }
    }
}

__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_feat_lab(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, sp_kparams32 kp,
            int64_t n_units, const int64_t *__restrict__ foff, int64_t n_feat, int S,
            uint8_t *__restrict__ label, const uint32_t *__restrict__ bloom, int bloom_bits,
            unsigned long long *__restrict__ counts) {
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; u < n_units; u += stride) {
        map_feat_cursor cur;
        cur.f = -1;
        cur.next = 0;
        map_pair_scan<uint32_t>(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, [&](int64_t start, uint32_t fwd, uint32_t rc) {
            const uint32_t slot = sp_slot_of32(fwd, rc, kp);
            const uint32_t l = label[slot];
            if (l && map_feat_locate(cur, start, kp.k, foff, n_feat)) {
                if (!(l & 0x80u)) label[slot] = (uint8_t)(l | 0x80u);
                const int sg = (int)(l & 0x7fu) - 1;
                atomicAdd(&counts[cur.f * S + sg], 1ULL);
            }
        });
    }
}

// ----------------------------------------------------------------- interval mode (BED features over the resident genome)
// The reference maps `-custom_features` as a FASTA of feature sequences (__main__.py:509-517, Seqs.py:228-244): at
// wheat scale that is 10 Gb of sequence that is already resident in HBM as the genome.  Intervals need no upload:
//   kv_cover    marks every k-mer start that lies, with its whole k-mer, inside at least one interval (1 bit per start);
//   k5_map_mask the usual scan, restricted to covered starts: one 64-bit mask per (unit of 64 starts, subgenome) --
//               bit j = start 64 u + j carries a k-mer of that subgenome; labelled k-mers are marked "seen" only at
//               covered starts, exactly the k-mers a feature FASTA would contain;
//   kv_count    one wave per interval: popcounts of the masks over its units (edge units masked) -> counts[i][sg].
// Overlapping and nested intervals cost nothing extra, every interval reads only its own units.
__global__ void __launch_bounds__(256)
kv_cover(const int32_t *__restrict__ chrom, const int64_t *__restrict__ start, const int64_t *__restrict__ end, int64_t n,
         int k, const int64_t *__restrict__ ubase /* first unit of every chromosome */, unsigned long long *__restrict__ cov) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wv; i < n; i += nw) {
        const int64_t a = start[i], b = end[i] - k + 1;     // starts [a, b)
        if (b <= a) continue;
        unsigned long long *cw = cov + ubase[chrom[i]];
        const int64_t u0 = a >> 6, u1 = (b - 1) >> 6;
        for (int64_t u = u0 + lane; u <= u1; u += 64) {
            unsigned long long m = ~0ULL;
            if (u == u0) m &= ~0ULL << (a & 63);
            if (u == u1 && (b & 63)) m &= (1ULL << (b & 63)) - 1ULL;
            if (m == ~0ULL) {
                if (cw[u] != ~0ULL) cw[u] = ~0ULL;     // (plain store of all ones: idempotent, races are benign)
            } else {
                atomicOr(&cw[u], m);
            }
        }
    }
}

// (label-table engine, more than 7 subgenomes; the pair-table engines take k5_map_mask2 above)
__global__ void __launch_bounds__(MAP_BLOCK)
k5_map_mask_lab(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm, sp_kparams32 kp, int64_t n_units, int S,
                uint8_t *__restrict__ label, const uint32_t *__restrict__ bloom, int bloom_bits,
                const unsigned long long *__restrict__ cov, unsigned long long *__restrict__ masks /* n_units x S */) {
    int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; u < n_units; u += stride) {
        const unsigned long long cv = cov[u];
        if (__all(cv == 0ULL)) continue;         // nothing of this wave's 4096 starts lies in a feature
        map_pair_scan<uint32_t>(pk, nm, u * SP_UNIT, kp, bloom, bloom_bits, [&](int64_t start, uint32_t fwd, uint32_t rc) {
            const unsigned long long bit = 1ULL << (start & 63);
            if (!(cv & bit)) return;
            const uint32_t slot = sp_slot_of32(fwd, rc, kp);
            const uint32_t l = label[slot];
            if (l) {
                if (!(l & 0x80u)) label[slot] = (uint8_t)(l | 0x80u);
                atomicOr(&masks[u * S + ((int)(l & 0x7fu) - 1)], bit);
            }
        });
    }
}

__global__ void __launch_bounds__(256)
kv_count(const int32_t *__restrict__ chrom, const int64_t *__restrict__ start, const int64_t *__restrict__ end, int64_t n,
         int k, int S, const int64_t *__restrict__ ubase, const unsigned long long *__restrict__ masks,
         unsigned long long *__restrict__ counts /* n x S */) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wv; i < n; i += nw) {
        const int64_t a = start[i], b = end[i] - k + 1;
        const int64_t u0 = a >> 6, u1 = b > a ? (b - 1) >> 6 : u0 - 1;
        const unsigned long long *mk = masks + ubase[chrom[i]] * S;
        for (int sg = 0; sg < S; sg++) {
            unsigned long long c = 0;
            for (int64_t u = u0 + lane; u <= u1; u += 64) {
                unsigned long long m = mk[u * S + sg];
                if (u == u0) m &= ~0ULL << (a & 63);
                if (u == u1 && (b & 63)) m &= (1ULL << (b & 63)) - 1ULL;
                c += __popcll(m);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            if (lane == 0) counts[i * S + sg] = c;
        }
    }
}

__global__ void __launch_bounds__(256)
k4_count_seen(const uint8_t *__restrict__ label, int64_t nslots, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[16];
    unsigned long long n = 0;
    const int64_t n16 = nslots >> 4;
    const uint4 *l4 = reinterpret_cast<const uint4 *>(label);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (int64_t)gridDim.x * blockDim.x) {
        uint4 v = l4[i];
        n += __popc(v.x & 0x80808080u) + __popc(v.y & 0x80808080u) + __popc(v.z & 0x80808080u) +
             __popc(v.w & 0x80808080u);
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n16 << 4) + threadIdx.x; i < nslots; i += blockDim.x) n += (label[i] >> 7) & 1u;
    unsigned long long t = sp_block_sum_u64(n, red);
    if (threadIdx.x == 0 && t) atomicAdd(out, t);
}

// pack kernel from sp_ctx.hip
__global__ void k0_pack(const uint8_t *ascii, int64_t len, uint32_t *pk, uint32_t *pm, uint32_t *nm, int64_t n_mask_words);

static int64_t map_nslots_host(int64_t len, int64_t bin_size, int64_t chunk_size, int k) {
    int64_t L = len > 0 ? len : 1;
    int64_t nb = (L + bin_size - 1) / bin_size;
    int64_t nch = chunk_size > 0 ? (L + (k - 1)) / chunk_size + 1 : 1;
    return nb + nch;
}

int sp_sparse_labels_set(sp_ctx *ctx, const uint64_t *keys, const uint8_t *sg, int64_t n, bool on_device);   // sp_sparse.hip

// the flags of one sp_labels_set call -> page-locked host memory the kernel writes directly: a hipMemcpy of 32 bytes
// would queue on the copy engine behind the matrix rows that are still travelling to the host (2 ms per pass)
__global__ void k4_flags_out(const unsigned long long *__restrict__ d_flags, unsigned long long *__restrict__ h_flags) {
    if (threadIdx.x < 4) h_flags[threadIdx.x] = d_flags[threadIdx.x];
}

// labels handed over in device memory: the largest one, for the `label < n_sg` check the host does on host arrays
__global__ void __launch_bounds__(256)
k4_label_max(const uint8_t *__restrict__ sg, int64_t n, unsigned int *__restrict__ out) {
    unsigned int m = 0;
    // (16 labels per load where the array allows it: byte by byte this took 0.25 ms for 2.2 M labels)
    const int64_t head = ((16 - (int64_t)((uintptr_t)sg & 15)) & 15) < n ? ((16 - (int64_t)((uintptr_t)sg & 15)) & 15) : n;
    const int64_t n16 = (n - head) / 16;
    const uint4 *v4 = reinterpret_cast<const uint4 *>(sg + head);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = v4[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned int x = (w[q] >> (8 * b)) & 255u;
                m = x > m ? x : m;
            }
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (i < head || i >= head + 16 * n16) m = sg[i] > m ? sg[i] : m;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int x = __shfl_down(m, o, 64);
        m = x > m ? x : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
int sp_sparse_map_launch(sp_ctx *ctx, sp_chrom &c, const sp_map_params &P, int *d_counts, unsigned long long *d_n);
int sp_sparse_feat_launch(sp_ctx *ctx, const uint32_t *d_pk, const uint32_t *d_pm, const uint32_t *d_nm, int64_t n_units,
                          const int64_t *d_foff, int64_t n_feat, int S, unsigned long long *d_counts);
int sp_sparse_hit(sp_ctx *ctx, unsigned long long *d_n);
int sp_sparse_mask_launch(sp_ctx *ctx, sp_chrom &c, int64_t n_units, int S, const unsigned long long *d_cov,
                          unsigned long long *d_masks);

// Build the pair filter over the labelled keys (device array, canonical 2-bit keys) at the smallest
// size whose fill stays below MAP_FILL_MAX.  Shared by the dense and the sparse label paths.
int sp_map_filter_build(sp_ctx *ctx, const unsigned long long *d_keys, int64_t n) {
    if (!ctx->d_bloom) SP_HIP(ctx, hipMalloc(&ctx->d_bloom, ((size_t)1 << MAP_BLOOM_MAX_BITS) / 8 + 64));   // + fill counter
    int bits = MAP_BLOOM_MIN_BITS;
    while (bits < MAP_BLOOM_MAX_BITS - 1 && ((int64_t)1 << bits) < 6 * n) bits++;
    // The size is a speed heuristic (fill <= MAP_FILL_MAX), never a matter of correctness: a label set of the size of
    // the previous one (the same set, pass after pass, in a pipeline that re-maps) takes the size chosen then without
    // measuring the fill again -- the measurement is a device -> host round trip per candidate size.
    const bool cached = n > 0 && n == ctx->bloom_last_n && ctx->k == ctx->bloom_last_k && ctx->bloom_last_bits >= bits;
    if (cached) bits = ctx->bloom_last_bits;
    unsigned long long *d_small = (unsigned long long *)((char *)ctx->d_bloom + ((size_t)1 << MAP_BLOOM_MAX_BITS) / 8);
    // round 6: words addressed by the smaller-hashed core of the (k-1)-mer (sp_map.h); SP_MAP_FILTER=x keeps the round-1
    // addressing by the (k-1)-mer itself (cross-check).  ctx->bloom_bits carries the flag to every kernel that probes.
    const char *env_mf = getenv("SP_MAP_FILTER");
    const int mode = (env_mf && env_mf[0] == 'x') ? 0 : MAP_BLOOM_CORE;
    for (;;) {
        const size_t bytes = ((size_t)1 << bits) / 8;
        SP_HIP(ctx, hipMemsetAsync(ctx->d_bloom, 0, bytes, ctx->stream));
        ctx->bloom_bits = bits | mode;
        if (n == 0) break;
        SP_LAUNCH(ctx, "k4_pair_filter", k4_pair_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_keys, n,
                  ctx->k, ctx->d_bloom, bits | mode);
        if (bits >= MAP_BLOOM_MAX_BITS || cached) break;
        SP_HIP(ctx, hipMemsetAsync(d_small, 0, 8, ctx->stream));
        SP_LAUNCH(ctx, "k4_filter_fill", k4_filter_fill, dim3(256), dim3(256), 0, (const uint32_t *)ctx->d_bloom,
                  (int64_t)(bytes / 4), d_small);
        unsigned long long set = 0;
        SP_HIP(ctx, hipMemcpyAsync(&set, d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const double fill = (double)set / (double)((int64_t)1 << bits);
        if (getenv("SP_DEBUG_FILTER")) fprintf(stderr, "[sp] pair filter: 2^%d bits, fill %.3f (n=%lld)\n", bits, fill, (long long)n);
        if (fill <= MAP_FILL_MAX) break;
        bits++;
    }
    ctx->bloom_last_n = n;
    ctx->bloom_last_bits = bits;
    ctx->bloom_last_k = ctx->k;
    return SP_OK;
}

extern "C" {

static int labels_set_impl(sp_ctx *ctx, const uint64_t *keys, const uint8_t *sg, int64_t n, int n_sg, bool on_device) {
    if (!ctx || n < 0 || (n > 0 && (!keys || !sg)) || n_sg < 1 || n_sg > 126)
        return sp_fail(ctx, SP_EINVAL, "sp_labels_set: bad arguments");
    if (ctx->k <= 0 || (ctx->nslots <= 0 && !ctx->sparse_mode))
        return sp_fail(ctx, SP_EINVAL, "sp_labels_set: call sp_count first (it fixes k)");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    // device flags of this call: [0] largest label (device hand-over), [2] compact table: overflow table full,
    // [3] keys in the overflow table.  Read back ONCE, at the end.
    int rcfl = sp_buf_ensure(ctx, ctx->b_lflags, 64);
    if (rcfl) return rcfl;
    unsigned long long *d_flags = (unsigned long long *)ctx->b_lflags.p;
    SP_HIP(ctx, hipMemsetAsync(d_flags, 0, 64, ctx->stream));
    if (on_device && n > 0) {
        SP_LAUNCH(ctx, "k4_label_max", k4_label_max, dim3((unsigned)(n < 65536 ? 1 : ctx->n_cu)), dim3(256), 0, sg, n, (unsigned int *)d_flags);
    } else {
        uint8_t mx = 0;     // (a reduction the compiler vectorises; an early-exit loop over 2 M labels took 1 ms)
        for (int64_t i = 0; i < n; i++) mx = sg[i] > mx ? sg[i] : mx;
        if (n > 0 && mx >= n_sg) return sp_fail(ctx, SP_EINVAL, "sp_labels_set: label %d >= n_sg %d", (int)mx, n_sg);
    }
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (!ctx->h_lflags) SP_HIP(ctx, hipHostMalloc((void **)&ctx->h_lflags, 64, hipHostMallocDefault));
    if (on_device && n > 0) {
        // the labels are checked BEFORE any table is touched (one more round trip of a 64-byte flag block, ~20 us): a
        // bad hand-over must leave the previous label set intact -- and must not leave labels >= n_sg in the hashed
        // table of the k > 15 engine (advisor r04)
        SP_LAUNCH(ctx, "k4_flags_out", k4_flags_out, dim3(1), dim3(64), 0, (const unsigned long long *)d_flags, ctx->h_lflags);
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if ((int)(unsigned int)ctx->h_lflags[0] >= n_sg)
            return sp_fail(ctx, SP_EINVAL, "sp_labels_set: label %d >= n_sg %d", (int)(unsigned int)ctx->h_lflags[0], n_sg);
    }
    auto label_check = [&]() -> int {      // after k4_flags_out and a synchronisation of the stream
        if (!(on_device && n > 0)) return SP_OK;
        const unsigned long long hf = ctx->h_lflags[0];
        if ((int)(unsigned int)hf >= n_sg) {
            ctx->labels_ready = false;
            return sp_fail(ctx, SP_EINVAL, "sp_labels_set: label %d >= n_sg %d", (int)(unsigned int)hf, n_sg);
        }
        return SP_OK;
    };
    if (ctx->sparse_mode) {
        ctx->n_sg = n_sg;
        ctx->n_labels = n;
        int rcs = sp_sparse_labels_set(ctx, keys, sg, n, on_device);
        if (rcs) return rcs;
        SP_LAUNCH(ctx, "k4_flags_out", k4_flags_out, dim3(1), dim3(64), 0, (const unsigned long long *)d_flags, ctx->h_lflags);
        SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return label_check();
    }
    const char *eng = getenv("SP_MAP_ENGINE");
    ctx->map_engine = (n_sg > MAP_PAIR_MAX_SG || (eng && eng[0] == '1')) ? 1 : 0;
    ctx->n_sg = n_sg;
    ctx->labels_ready = false;
    const int64_t entries = 1LL << (2 * (ctx->k - 1));
    // Compact table (sp_map.h) when the labels fit 2-bit fields, the direct table is beyond every cache and the
    // compact one is at least four times smaller: 2^bb buckets of four entries, bb >= log2(SP_CTAB_FACTOR (default 2) x n)
    // -- a labelled k-mer brings ~3.6 keys (two (k-1)-mers, each under two (k-3)-mers, shared with its neighbours in
    // a run) -- and at most 2 bits of the mixed (k-3)-mer in the tag.
    ctx->ct_bb = 0;
    bool compact = false;
    int bb = 8;
    {
        const char *env_ct = getenv("SP_CTAB");           // "0": never (cross-check), "1": whenever the tag fits
        const char *env_f = getenv("SP_CTAB_FACTOR");
        const int64_t factor = env_f && atoll(env_f) > 0 ? atoll(env_f) : 2;
        const int sb = 2 * (ctx->k - 3);
        while (bb < 30 && ((int64_t)1 << bb) < factor * (n > 0 ? n : 1)) bb++;
        if (bb < sb - MAP_CT_MAX_TAG_BITS) bb = sb - MAP_CT_MAX_TAG_BITS;
        const bool fits = ctx->map_engine == 0 && n_sg <= 3 && ctx->k >= 5 && bb >= 1 && bb <= sb && bb <= 27;
        const bool forced = env_ct && env_ct[0] == '1';
        compact = fits && !(env_ct && env_ct[0] == '0') &&
                  (forced || (entries * 4 > (64LL << 20) && ((int64_t)16 << bb) * 4 <= entries * 4));
    }
    // direct table: the previous label set (same k, same buffer) is un-built key by key instead of memset -- while its
    // keys are still in b_labkeys
    bool table_clean = false;
    if (!compact && ctx->map_engine == 0 && ctx->ptab_k == ctx->k && ctx->b_ptab.p && ctx->b_ptab.cap >= entries * 4) {
        if (ctx->ptab_n > 0)
            SP_LAUNCH(ctx, "k4_pair_clear", k4_pair_clear, dim3((unsigned)((ctx->ptab_n + 255) / 256)), dim3(256), 0,
                      (const unsigned long long *)ctx->b_labkeys.p, ctx->ptab_n, ctx->k, (uint32_t *)ctx->b_ptab.p);
        table_clean = true;
    }
    ctx->ptab_k = 0;      // (from here on the direct table is in an unknown state unless it is rebuilt below)
    ctx->ptab_n = 0;
    ctx->n_labels = n;
    int rcb = sp_buf_ensure(ctx, ctx->b_labkeys, (n > 0 ? n : 1) * 9);
    if (rcb) return rcb;
    unsigned long long *d_keys = (unsigned long long *)ctx->b_labkeys.p;
    uint8_t *d_sg = (uint8_t *)(d_keys + (n > 0 ? n : 1));
    if (n > 0 && (const void *)keys != (const void *)d_keys) {
        SP_HIP(ctx, hipMemcpyAsync(d_keys, keys, (size_t)n * 8, kind, ctx->stream));
        SP_HIP(ctx, hipMemcpyAsync(d_sg, sg, (size_t)n, kind, ctx->stream));
    }
    if (compact) {
        const int64_t nb = (int64_t)1 << bb;
        int64_t ovf_n = 4096;
        while (ovf_n < n / 2) ovf_n <<= 1;
        rcb = sp_buf_ensure(ctx, ctx->b_ctab, nb * 16);
        if (rcb) return rcb;
        rcb = sp_buf_ensure(ctx, ctx->b_covf, ovf_n * 8);
        if (rcb) return rcb;
        SP_HIP(ctx, hipMemsetAsync(ctx->b_ctab.p, 0, (size_t)nb * 16, ctx->stream));
        SP_HIP(ctx, hipMemsetAsync(ctx->b_covf.p, 0, (size_t)ovf_n * 8, ctx->stream));
        ctx->ct_bb = bb;
        ctx->ct_ovf_mask = (uint32_t)(ovf_n - 1);
        if (n > 0)
            SP_LAUNCH(ctx, "k4_ctab_build", k4_ctab_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                      (const unsigned long long *)d_keys, (const uint8_t *)d_sg, n, ctx->k, map_ptab_of(ctx), d_flags + 2);
    }
    auto build_direct = [&]() -> int {
        int rc2 = sp_buf_ensure(ctx, ctx->b_ptab, entries * 4);
        if (rc2) return rc2;
        if (!table_clean) SP_HIP(ctx, hipMemsetAsync(ctx->b_ptab.p, 0, (size_t)entries * 4, ctx->stream));
        if (n > 0)
            SP_LAUNCH(ctx, "k4_pair_table", k4_pair_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                      (const unsigned long long *)d_keys, (const uint8_t *)d_sg, n, ctx->k, (uint32_t *)ctx->b_ptab.p);
        ctx->ptab_k = ctx->k;
        ctx->ptab_n = n;
        return SP_OK;
    };
    if (ctx->map_engine == 0 && !compact) {
        rcb = build_direct();
        if (rcb) return rcb;
    } else if (ctx->map_engine != 0) {
        if (!ctx->d_label) SP_HIP(ctx, hipMalloc(&ctx->d_label, (size_t)ctx->nslots));
        SP_HIP(ctx, hipMemsetAsync(ctx->d_label, 0, (size_t)ctx->nslots, ctx->stream));
        if (n > 0) {
            const sp_kparams kp = sp_make_kparams(ctx->k);
            SP_LAUNCH(ctx, "k4_labels", k4_labels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                      (const unsigned long long *)d_keys, (const uint8_t *)d_sg, n, kp, ctx->d_label);
        }
    }
    const int rcf = sp_map_filter_build(ctx, n > 0 ? d_keys : nullptr, n);
    if (rcf) return rcf;
    SP_LAUNCH(ctx, "k4_flags_out", k4_flags_out, dim3(1), dim3(64), 0, (const unsigned long long *)d_flags, ctx->h_lflags);
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int rcl = label_check();
    if (rcl) return rcl;
    if (compact && n > 0) {
        const unsigned long long hf[2] = {ctx->h_lflags[2], ctx->h_lflags[3]};
        if (getenv("SP_DEBUG_FILTER"))
            fprintf(stderr, "[sp] compact pair table: 2^%d buckets, %llu entries in the overflow table (%lld labelled k-mers)\n",
                    bb, hf[1], (long long)n);
        if (hf[0]) {          // overflow table full (adversarial key sets only): the direct table takes over
            ctx->ct_bb = 0;
            table_clean = false;
            rcb = build_direct();
            if (rcb) return rcb;
            SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    ctx->labels_ready = true;
    return SP_OK;
}

int sp_labels_set(sp_ctx *ctx, const uint64_t *keys, const uint8_t *sg, int64_t n, int n_sg) {
    return labels_set_impl(ctx, keys, sg, n, n_sg, false);
}

int sp_labels_set_device(sp_ctx *ctx, const uint64_t *d_keys, const uint8_t *d_sg, int64_t n, int n_sg) {
    return labels_set_impl(ctx, d_keys, d_sg, n, n_sg, true);
}

int sp_map_nslots(sp_ctx *ctx, int chrom, int64_t bin_size, int64_t chunk_size, int64_t *nslots) {
    if (!ctx || !nslots || chrom < 0 || chrom >= (int)ctx->chroms.size() || bin_size < 1 || chunk_size < 0)
        return sp_fail(ctx, SP_EINVAL, "sp_map_nslots: bad arguments");
    *nslots = map_nslots_host(ctx->chroms[(size_t)chrom].len, bin_size, chunk_size, ctx->k);
    return SP_OK;
}

int sp_map_bins(sp_ctx *ctx, int chrom, int64_t bin_size, int64_t chunk_size, int32_t *slot_counts,
                int64_t nslots, int64_t *n_mapped) {
    if (!ctx || !slot_counts || chrom < 0 || chrom >= (int)ctx->chroms.size() || bin_size < 1 ||
        chunk_size < 0)
        return sp_fail(ctx, SP_EINVAL, "sp_map_bins: bad arguments");
    if (!(ctx->sparse_mode ? ctx->d_hkeys != nullptr : ctx->labels_ready))
        return sp_fail(ctx, SP_EINVAL, "sp_map_bins: call sp_labels_set first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    sp_chrom &c = ctx->chroms[(size_t)chrom];
    const int S = ctx->n_sg;
    int64_t need = map_nslots_host(c.len, bin_size, chunk_size, ctx->k);
    if (nslots < need) return sp_fail(ctx, SP_EINVAL, "sp_map_bins: nslots %lld < %lld", (long long)nslots, (long long)need);
    const size_t bytes = (size_t)nslots * S * sizeof(int);
    const size_t bytes8 = (bytes + 7) & ~(size_t)7;
    ctx->map_all_valid = false;
    int rcb = sp_buf_ensure(ctx, ctx->b_map, (int64_t)bytes8 + 8);
    if (rcb) return rcb;
    int *d_counts = (int *)ctx->b_map.p;
    unsigned long long *d_n = (unsigned long long *)((char *)d_counts + bytes8);
    SP_HIP(ctx, hipMemsetAsync(d_counts, 0, bytes8 + 8, ctx->stream));
    sp_map_params P;
    P.n_units = (c.len + SP_UNIT - 1) / SP_UNIT;
    P.bin_size = bin_size;
    P.chunk_size = chunk_size;
    P.nslots = nslots;
    P.S = S;
    int64_t local = MAP_RANGE / bin_size + 3 + (chunk_size > 0 ? MAP_RANGE / chunk_size + 2 : 0);
    P.use_lds = (local * S <= MAP_LDS_ENTRIES) ? 1 : 0;
    if (ctx->sparse_mode) {
        int rcs = sp_sparse_map_launch(ctx, c, P, d_counts, d_n);
        if (rcs) return rcs;
    } else if (P.n_units > 0) {
        const sp_kparams32 kp = sp_make_kparams32(ctx->k);
        int64_t n_ranges = (P.n_units + MAP_BLOCK - 1) / MAP_BLOCK;
        int64_t grid = n_ranges;
        if (grid > (int64_t)ctx->n_cu * MAP_GRID_MULT) grid = (int64_t)ctx->n_cu * MAP_GRID_MULT;
        if (ctx->map_engine == 0) {
            std::vector<map_chrom_desc> hd(1);
            hd[0] = map_chrom_desc{c.d_pk, c.d_pm, c.d_nm, P.n_units, nslots, 0, d_counts, d_n};
            int rcl = map_launch_dense(ctx, hd, n_ranges, P);
            if (rcl) return rcl;
        } else
            SP_LAUNCH(ctx, "k5_map_lab", k5_map_lab, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, c.d_pk, c.d_nm, kp, P,
                      ctx->d_label, ctx->d_bloom, ctx->bloom_bits, d_counts, d_n);
    }
    unsigned long long hn = 0;
    SP_HIP(ctx, hipMemcpyAsync(slot_counts, d_counts, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(&hn, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_mapped) *n_mapped = (int64_t)hn;
    return SP_OK;
}

int sp_map_bins_all(sp_ctx *ctx, int64_t bin_size, int64_t chunk_size, const int64_t *slot_off,
                    int32_t *slot_counts, int64_t *n_mapped) {
    if (!ctx || !slot_off || !slot_counts || bin_size < 1 || chunk_size < 0)
        return sp_fail(ctx, SP_EINVAL, "sp_map_bins_all: bad arguments");
    if (!(ctx->sparse_mode ? ctx->d_hkeys != nullptr : ctx->labels_ready))
        return sp_fail(ctx, SP_EINVAL, "sp_map_bins_all: call sp_labels_set first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int C = (int)ctx->chroms.size();
    const int S = ctx->n_sg;
    for (int i = 0; i < C; i++) {
        int64_t need = map_nslots_host(ctx->chroms[(size_t)i].len, bin_size, chunk_size, ctx->k);
        if (slot_off[i + 1] - slot_off[i] < need)
            return sp_fail(ctx, SP_EINVAL, "sp_map_bins_all: chromosome %d needs %lld slots", i, (long long)need);
    }
    const int64_t total = slot_off[C];
    const size_t bytes = (size_t)total * S * sizeof(int);
    const size_t bytes8 = (bytes + 7) & ~(size_t)7;
    int rcb = sp_buf_ensure(ctx, ctx->b_map, (int64_t)(bytes8 + 8 * (size_t)C));
    if (rcb) return rcb;
    int *d_counts = (int *)ctx->b_map.p;
    unsigned long long *d_n = (unsigned long long *)((char *)d_counts + bytes8);
    SP_HIP(ctx, hipMemsetAsync(d_counts, 0, bytes8 + 8 * (size_t)C, ctx->stream));
    const sp_kparams32 kp = sp_make_kparams32(ctx->k);
    int64_t local = MAP_RANGE / bin_size + 3 + (chunk_size > 0 ? MAP_RANGE / chunk_size + 2 : 0);
    std::vector<map_chrom_desc> hd;     // pair-table engine: every chromosome in ONE launch
    int64_t all_ranges = 0;
    sp_map_params Pall;
    Pall.n_units = 0;
    Pall.bin_size = bin_size;
    Pall.chunk_size = chunk_size;
    Pall.nslots = 0;
    Pall.S = S;
    Pall.use_lds = (local * S <= MAP_LDS_ENTRIES) ? 1 : 0;
    for (int i = 0; i < C; i++) {
        sp_chrom &c = ctx->chroms[(size_t)i];
        sp_map_params P;
        P.n_units = (c.len + SP_UNIT - 1) / SP_UNIT;
        P.bin_size = bin_size;
        P.chunk_size = chunk_size;
        P.nslots = slot_off[i + 1] - slot_off[i];
        P.S = S;
        P.use_lds = (local * S <= MAP_LDS_ENTRIES) ? 1 : 0;
        if (P.n_units == 0) continue;
        if (ctx->sparse_mode) {
            int rcs = sp_sparse_map_launch(ctx, c, P, d_counts + slot_off[i] * S, d_n + i);
            if (rcs) return rcs;
            continue;
        }
        int64_t n_ranges = (P.n_units + MAP_BLOCK - 1) / MAP_BLOCK;
        if (ctx->map_engine == 0) {
            hd.push_back(map_chrom_desc{c.d_pk, c.d_pm, c.d_nm, P.n_units, P.nslots, all_ranges, d_counts + slot_off[i] * S,
                                        d_n + i});
            all_ranges += n_ranges;
            continue;
        }
        int64_t grid = n_ranges;
        if (grid > (int64_t)ctx->n_cu * MAP_GRID_MULT) grid = (int64_t)ctx->n_cu * MAP_GRID_MULT;
        SP_LAUNCH(ctx, "k5_map_lab", k5_map_lab, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, c.d_pk, c.d_nm, kp, P,
                  ctx->d_label, ctx->d_bloom, ctx->bloom_bits, d_counts + slot_off[i] * S, d_n + i);
    }
    if (!hd.empty()) {
        int rcl = map_launch_dense(ctx, hd, all_ranges, Pall);
        if (rcl) return rcl;
    }
    SP_HIP(ctx, hipMemcpyAsync(slot_counts, d_counts, bytes, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<unsigned long long> hn((size_t)C, 0);
    SP_HIP(ctx, hipMemcpyAsync(hn.data(), d_n, 8 * (size_t)C, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_mapped)
        for (int i = 0; i < C; i++) n_mapped[i] = (int64_t)hn[(size_t)i];
    ctx->map_all_valid = true;   // b_map holds the slot counts of every chromosome (for sp_stack_windows)
    return SP_OK;
}

// window stack into a DEVICE table (accumulated: the caller clears it).  Entry i of slot_off / win_off /
// seg_start describes local chromosome (or chromosome segment) i: its slots in the table sp_map_bins_all left
// behind, the first window row of the chromosome it belongs to, and where its base 0 lies in that chromosome.
int sp_stack_windows_dev(sp_ctx *ctx, int64_t bin_size, int64_t chunk_size, int64_t window_size, const int64_t *slot_off,
                         const int64_t *win_off, const int64_t *seg_start, void *d_win) {
    if (!ctx || !slot_off || !win_off || !d_win || bin_size < 1 || chunk_size < 0 || window_size < 1)
        return sp_fail(ctx, SP_EINVAL, "sp_stack_windows_dev: bad arguments");
    if (!ctx->map_all_valid) return sp_fail(ctx, SP_EINVAL, "sp_stack_windows: call sp_map_bins_all first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int C = (int)ctx->chroms.size();
    const int S = ctx->n_sg;
    const int64_t total_slots = slot_off[C];
    int rcb = sp_buf_ensure(ctx, ctx->b_win, (int64_t)(24 * (size_t)(C + 1)));
    if (rcb) return rcb;
    long long *d_soff = (long long *)ctx->b_win.p;
    long long *d_woff = d_soff + (C + 1), *d_seg = d_woff + (C + 1);
    SP_HIP(ctx, hipMemcpyAsync(d_soff, slot_off, 8 * (size_t)(C + 1), hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_woff, win_off, 8 * (size_t)C, hipMemcpyHostToDevice, ctx->stream));
    if (seg_start) SP_HIP(ctx, hipMemcpyAsync(d_seg, seg_start, 8 * (size_t)C, hipMemcpyHostToDevice, ctx->stream));
    if (total_slots > 0)
        SP_LAUNCH(ctx, "k5_stack", k5_stack, dim3((unsigned)((total_slots * S + 255) / 256)), dim3(256), 0,
                  (const int *)ctx->b_map.p, total_slots, S, C, d_soff, d_woff, seg_start ? d_seg : (long long *)nullptr,
                  bin_size, chunk_size, window_size, ctx->k, (unsigned long long *)d_win);
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // slot_off / win_off are the caller's
    return SP_OK;
}

static int stack_check(sp_ctx *ctx, int64_t window_size, const int64_t *win_off) {
    const int C = (int)ctx->chroms.size();
    for (int i = 0; i < C; i++) {
        int64_t need = (ctx->chroms[(size_t)i].len + window_size - 1) / window_size + 1;
        if (win_off[i + 1] - win_off[i] < need)
            return sp_fail(ctx, SP_EINVAL, "sp_stack_windows: chromosome %d needs %lld windows", i, (long long)need);
    }
    return SP_OK;
}

int sp_stack_windows(sp_ctx *ctx, int64_t bin_size, int64_t chunk_size, int64_t window_size,
                     const int64_t *slot_off, const int64_t *win_off, int64_t *win_counts) {
    if (!ctx || !slot_off || !win_off || !win_counts || bin_size < 1 || chunk_size < 0 || window_size < 1)
        return sp_fail(ctx, SP_EINVAL, "sp_stack_windows: bad arguments");
    if (!ctx->map_all_valid) return sp_fail(ctx, SP_EINVAL, "sp_stack_windows: call sp_map_bins_all first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = stack_check(ctx, window_size, win_off);
    if (rc) return rc;
    const int C = (int)ctx->chroms.size();
    const size_t wbytes = (size_t)win_off[C] * ctx->n_sg * 8;
    rc = sp_buf_ensure(ctx, ctx->b_wtab, (int64_t)wbytes + 64);
    if (rc) return rc;
    SP_HIP(ctx, hipMemsetAsync(ctx->b_wtab.p, 0, wbytes, ctx->stream));
    rc = sp_stack_windows_dev(ctx, bin_size, chunk_size, window_size, slot_off, win_off, nullptr, ctx->b_wtab.p);
    if (rc) return rc;
    SP_HIP(ctx, hipMemcpyAsync(win_counts, ctx->b_wtab.p, wbytes, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_enrich_dev(sp_ctx *ctx, const void *d_counts, int64_t W, int S, double max_pval, double min_ratio, double *pvals,
                  int32_t *argmin, uint8_t *sig, double *ratios);   // sp_enrich.hip

// map -> stack -> enrich without leaving the device: the window table is built in HBM, the column totals are
// reduced there, every window row (empty ones included: they change no total) is tested, and the table and the
// decisions come back in one go.  The caller keeps the rows that have any count (Circos.py:734-742).
int sp_stack_enrich(sp_ctx *ctx, int64_t bin_size, int64_t chunk_size, int64_t window_size, const int64_t *slot_off,
                    const int64_t *win_off, double max_pval, double min_ratio, int64_t *win_counts, double *pvals,
                    int32_t *argmin, uint8_t *sig, double *ratios) {
    if (!ctx || !slot_off || !win_off || !win_counts || !pvals || !argmin || !sig || !ratios || bin_size < 1 ||
        chunk_size < 0 || window_size < 1)
        return sp_fail(ctx, SP_EINVAL, "sp_stack_enrich: bad arguments");
    if (!ctx->map_all_valid) return sp_fail(ctx, SP_EINVAL, "sp_stack_enrich: call sp_map_bins_all first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = stack_check(ctx, window_size, win_off);
    if (rc) return rc;
    const int C = (int)ctx->chroms.size();
    const int64_t W = win_off[C];
    const size_t wbytes = (size_t)W * ctx->n_sg * 8;
    rc = sp_buf_ensure(ctx, ctx->b_wtab, (int64_t)wbytes + 64);
    if (rc) return rc;
    SP_HIP(ctx, hipMemsetAsync(ctx->b_wtab.p, 0, wbytes, ctx->stream));
    rc = sp_stack_windows_dev(ctx, bin_size, chunk_size, window_size, slot_off, win_off, nullptr, ctx->b_wtab.p);
    if (rc) return rc;
    SP_HIP(ctx, hipMemcpyAsync(win_counts, ctx->b_wtab.p, wbytes, hipMemcpyDeviceToHost, ctx->stream));
    return sp_enrich_dev(ctx, ctx->b_wtab.p, W, ctx->n_sg, max_pval, min_ratio, pvals, argmin, sig, ratios);
}

int sp_map_features(sp_ctx *ctx, const uint8_t *ascii, const int64_t *off, int64_t n_feat,
                    int64_t *counts) {
    if (!ctx || !off || !counts || n_feat < 0 || (n_feat > 0 && off[n_feat] > 0 && !ascii))
        return sp_fail(ctx, SP_EINVAL, "sp_map_features: bad arguments");
    if (!(ctx->sparse_mode ? ctx->d_hkeys != nullptr : ctx->labels_ready))
        return sp_fail(ctx, SP_EINVAL, "sp_map_features: call sp_labels_set first");
    const int S = ctx->n_sg;
    memset(counts, 0, (size_t)n_feat * S * sizeof(int64_t));
    if (n_feat == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    // the features are uploaded as they lie in the caller's buffer (back to back); the kernel rejects
    // k-mers that run across a feature boundary, so no separator copy is needed on the host
    const int64_t base = off[0], total = off[n_feat] - off[0];
    std::vector<int64_t> foff((size_t)n_feat + 1);
    for (int64_t f = 0; f <= n_feat; f++) {
        if (f > 0 && off[f] < off[f - 1]) return sp_fail(ctx, SP_EINVAL, "sp_map_features: offsets must be non-decreasing");
        foff[(size_t)f] = off[f] - base;
    }
    if (total == 0) return SP_OK;
    sp_tmp<uint8_t> d_ascii;
    sp_tmp<uint32_t> d_pk, d_nm;
    sp_tmp<int64_t> d_foff;
    sp_tmp<unsigned long long> d_counts;
    int64_t nmw = (total + 31) / 32 + SP_PAD_WORDS;
    SP_HIP(ctx, d_ascii.alloc((size_t)total));
    SP_HIP(ctx, d_pk.alloc((size_t)(4 * nmw)));   // LSB-first | MSB-first
    SP_HIP(ctx, d_nm.alloc((size_t)nmw));
    SP_HIP(ctx, d_foff.alloc((size_t)(n_feat + 1)));
    SP_HIP(ctx, d_counts.alloc((size_t)n_feat * S));
    SP_HIP(ctx, hipMemcpyAsync(d_ascii, ascii + base, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_foff, foff.data(), (size_t)(n_feat + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)n_feat * S * 8, ctx->stream));
    int64_t blocks = (nmw + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    SP_LAUNCH(ctx, "k0_pack", k0_pack, dim3((unsigned)blocks), dim3(256), 0, (const uint8_t *)d_ascii.p, total, d_pk.p,
              d_pk.p + 2 * nmw, d_nm.p, nmw);
    int64_t n_units = (total + SP_UNIT - 1) / SP_UNIT;
    if (ctx->sparse_mode) {
        int rcs = sp_sparse_feat_launch(ctx, d_pk.p, d_pk.p + 2 * nmw, d_nm.p, n_units, d_foff.p, n_feat, S, d_counts.p);
        if (rcs) return rcs;
    } else {
        const sp_kparams32 kp = sp_make_kparams32(ctx->k);
        int64_t grid = (n_units + MAP_BLOCK - 1) / MAP_BLOCK;
        if (grid > (int64_t)ctx->n_cu * MAP_GRID_MULT) grid = (int64_t)ctx->n_cu * MAP_GRID_MULT;
        if (ctx->map_engine == 0 && ctx->ct_bb)
            SP_LAUNCH(ctx, "k5_map_feat", k5_map_feat2<1>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, (const uint32_t *)d_pk.p,
                      (const uint32_t *)(d_pk.p + 2 * nmw), (const uint32_t *)d_nm.p, kp, n_units,
                      (const int64_t *)d_foff.p, n_feat, S, map_ptab_of(ctx), (const uint32_t *)ctx->d_bloom,
                      ctx->bloom_bits, d_counts.p);
        else if (ctx->map_engine == 0)
            SP_LAUNCH(ctx, "k5_map_feat", k5_map_feat2<0>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, (const uint32_t *)d_pk.p,
                      (const uint32_t *)(d_pk.p + 2 * nmw), (const uint32_t *)d_nm.p, kp, n_units,
                      (const int64_t *)d_foff.p, n_feat, S, map_ptab_of(ctx), (const uint32_t *)ctx->d_bloom,
                      ctx->bloom_bits, d_counts.p);
        else
            SP_LAUNCH(ctx, "k5_map_feat_lab", k5_map_feat_lab, dim3((unsigned)grid), dim3(MAP_BLOCK), 0,
                      (const uint32_t *)d_pk.p, (const uint32_t *)d_nm.p, kp, n_units, (const int64_t *)d_foff.p, n_feat,
                      S, ctx->d_label, (const uint32_t *)ctx->d_bloom, ctx->bloom_bits, d_counts.p);
    }
    SP_HIP(ctx, hipMemcpyAsync(counts, d_counts, (size_t)n_feat * S * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_map_intervals(sp_ctx *ctx, const int32_t *chrom, const int64_t *start, const int64_t *end, int64_t n,
                     int64_t *counts) {
    if (!ctx || n < 0 || (n > 0 && (!chrom || !start || !end || !counts)))
        return sp_fail(ctx, SP_EINVAL, "sp_map_intervals: bad arguments");
    if (!(ctx->sparse_mode ? ctx->d_hkeys != nullptr : ctx->labels_ready))
        return sp_fail(ctx, SP_EINVAL, "sp_map_intervals: call sp_labels_set first");
    const int C = (int)ctx->chroms.size(), S = ctx->n_sg, k = ctx->k;
    for (int64_t i = 0; i < n; i++) {
        if (chrom[i] < 0 || chrom[i] >= C) return sp_fail(ctx, SP_EINVAL, "sp_map_intervals: interval %lld: chromosome %d out of range", (long long)i, chrom[i]);
        if (start[i] < 0 || end[i] < start[i] || end[i] > ctx->chroms[(size_t)chrom[i]].len)
            return sp_fail(ctx, SP_EINVAL, "sp_map_intervals: interval %lld = [%lld, %lld) outside chromosome %d (%lld bases)",
                           (long long)i, (long long)start[i], (long long)end[i], chrom[i], (long long)ctx->chroms[(size_t)chrom[i]].len);
    }
    if (n == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int64_t> ubase((size_t)C + 1, 0);
    for (int c = 0; c < C; c++) ubase[(size_t)c + 1] = ubase[(size_t)c] + (ctx->chroms[(size_t)c].len + SP_UNIT - 1) / SP_UNIT + 1;
    const int64_t total_units = ubase[(size_t)C];
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_cov = 0, o_masks = al((size_t)total_units * 8), o_ub = o_masks + al((size_t)total_units * S * 8),
                 o_ch = o_ub + al((size_t)(C + 1) * 8), o_st = o_ch + al((size_t)n * 4), o_en = o_st + al((size_t)n * 8),
                 o_cnt = o_en + al((size_t)n * 8), bytes = o_cnt + al((size_t)n * S * 8);
    int rc = sp_buf_ensure(ctx, ctx->b_ival, (int64_t)bytes);
    if (rc) return rc;
    char *B = (char *)ctx->b_ival.p;
    unsigned long long *d_cov = (unsigned long long *)(B + o_cov), *d_masks = (unsigned long long *)(B + o_masks),
                       *d_counts = (unsigned long long *)(B + o_cnt);
    int64_t *d_ub = (int64_t *)(B + o_ub), *d_st = (int64_t *)(B + o_st), *d_en = (int64_t *)(B + o_en);
    int32_t *d_ch = (int32_t *)(B + o_ch);
    SP_HIP(ctx, hipMemsetAsync(B, 0, o_ub, ctx->stream));      // coverage + masks
    SP_HIP(ctx, hipMemcpyAsync(d_ub, ubase.data(), (size_t)(C + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_ch, chrom, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_st, start, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipMemcpyAsync(d_en, end, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    int64_t gw = (n + 3) / 4;       // four intervals (waves) per block
    if (gw > (int64_t)ctx->n_cu * 32) gw = (int64_t)ctx->n_cu * 32;
    SP_LAUNCH(ctx, "kv_cover", kv_cover, dim3((unsigned)gw), dim3(256), 0, (const int32_t *)d_ch, (const int64_t *)d_st,
              (const int64_t *)d_en, n, k, (const int64_t *)d_ub, d_cov);
    for (int c = 0; c < C; c++) {
        sp_chrom &ch = ctx->chroms[(size_t)c];
        const int64_t n_units = (ch.len + SP_UNIT - 1) / SP_UNIT;
        if (n_units == 0) continue;
        if (ctx->sparse_mode) {
            int rcs = sp_sparse_mask_launch(ctx, ch, n_units, S, d_cov + ubase[(size_t)c], d_masks + ubase[(size_t)c] * S);
            if (rcs) return rcs;
            continue;
        }
        const sp_kparams32 kp = sp_make_kparams32(k);
        int64_t grid = (n_units + MAP_BLOCK - 1) / MAP_BLOCK;
        if (grid > (int64_t)ctx->n_cu * MAP_GRID_MULT) grid = (int64_t)ctx->n_cu * MAP_GRID_MULT;
        if (ctx->map_engine == 0 && ctx->ct_bb)
            SP_LAUNCH(ctx, "k5_map_mask", k5_map_mask2<1>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, ch.d_pk, ch.d_pm, ch.d_nm, kp,
                      n_units, S, map_ptab_of(ctx), (const uint32_t *)ctx->d_bloom, ctx->bloom_bits,
                      (const unsigned long long *)(d_cov + ubase[(size_t)c]), d_masks + ubase[(size_t)c] * S);
        else if (ctx->map_engine == 0)
            SP_LAUNCH(ctx, "k5_map_mask", k5_map_mask2<0>, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, ch.d_pk, ch.d_pm, ch.d_nm, kp,
                      n_units, S, map_ptab_of(ctx), (const uint32_t *)ctx->d_bloom, ctx->bloom_bits,
                      (const unsigned long long *)(d_cov + ubase[(size_t)c]), d_masks + ubase[(size_t)c] * S);
        else
            SP_LAUNCH(ctx, "k5_map_mask_lab", k5_map_mask_lab, dim3((unsigned)grid), dim3(MAP_BLOCK), 0, ch.d_pk, ch.d_nm,
                      kp, n_units, S, ctx->d_label, (const uint32_t *)ctx->d_bloom, ctx->bloom_bits,
                      (const unsigned long long *)(d_cov + ubase[(size_t)c]), d_masks + ubase[(size_t)c] * S);
    }
    SP_LAUNCH(ctx, "kv_count", kv_count, dim3((unsigned)gw), dim3(256), 0, (const int32_t *)d_ch, (const int64_t *)d_st,
              (const int64_t *)d_en, n, k, S, (const int64_t *)d_ub, (const unsigned long long *)d_masks, d_counts);
    SP_HIP(ctx, hipMemcpyAsync(counts, d_counts, (size_t)n * S * 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

int sp_labels_hit(sp_ctx *ctx, int64_t *n_hit) {
    if (!ctx || !n_hit) return sp_fail(ctx, SP_EINVAL, "sp_labels_hit: bad arguments");
    if (!(ctx->sparse_mode ? ctx->d_hkeys != nullptr : ctx->labels_ready))
        return sp_fail(ctx, SP_EINVAL, "sp_labels_hit: call sp_labels_set first");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    sp_tmp<unsigned long long> d_n;
    unsigned long long h = 0;
    SP_HIP(ctx, d_n.alloc(1));
    SP_HIP(ctx, hipMemsetAsync(d_n.p, 0, 8, ctx->stream));
    if (ctx->sparse_mode) {
        int rcs = sp_sparse_hit(ctx, d_n.p);
        if (rcs) return rcs;
    } else {
        if (ctx->map_engine == 0) {
            if (ctx->n_labels > 0 && ctx->ct_bb)
                SP_LAUNCH(ctx, "k4_ctab_seen", k4_ctab_seen, dim3((unsigned)(ctx->n_cu * 4)), dim3(256), 0,
                          (const unsigned long long *)ctx->b_labkeys.p, ctx->n_labels, ctx->k, map_ptab_of(ctx), d_n.p);
            else if (ctx->n_labels > 0)
                SP_LAUNCH(ctx, "k4_pair_seen", k4_pair_seen, dim3((unsigned)(ctx->n_cu * 4)), dim3(256), 0,
                          (const unsigned long long *)ctx->b_labkeys.p, ctx->n_labels, ctx->k,
                          (const uint32_t *)ctx->b_ptab.p, d_n.p);
        } else {
            SP_LAUNCH(ctx, "k4_count_seen", k4_count_seen, dim3((unsigned)(ctx->n_cu * 8)), dim3(256), 0,
                      (const uint8_t *)ctx->d_label, ctx->nslots, d_n.p);
        }
    }
    SP_HIP(ctx, hipMemcpyAsync(&h, d_n.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_hit = (int64_t)h;
    return SP_OK;
}
}  // extern "C"
