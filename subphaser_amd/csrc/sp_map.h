// sp_map.h -- what the dense (sp_map.hip) and the sparse (sp_sparse.hip) map stages share: the pair
// filter, the pair-table field layout, the output-slot arithmetic of Seqs.map_kmer3's chunks and bins.
#pragma once
#include "sp_device.h"

#ifndef MAP_BLOOM_MAX_BITS
#define MAP_BLOOM_MAX_BITS 25
#endif
#ifndef MAP_BLOOM_MIN_BITS
#define MAP_BLOOM_MIN_BITS 12
#endif
#ifndef MAP_FILL_MAX
#define MAP_FILL_MAX 0.40
#endif
// Blocked Bloom filter: one 32-bit word per key (ONE memory access per probe), three bits in it.
//
// Round 6 -- WHICH word.  The map stage is bound by the number of these probes: one scattered 4-byte gather per pair of starts,
// priced per ACTIVE lane (tools/ubench_gather.hip: two gathers per quad issued by 2/3 of the lanes each cost 2/3).  Until now the
// word was the hashed (k-1)-mer x itself, so neighbouring pairs never met in a word.  A (k-1)-mer has two (k-3)-mer CORES, its
// first and its last k-3 bases, and the pairs along a scan form a chain: the last core of the pair at s is the first core of the
// pair at s + 2.  With MAP_BLOOM_CORE the word of x is addressed by the core of x with the SMALLER hash (canonical cores, so
// both strands agree): a core that beats both of its neighbours in the chain answers BOTH of its pairs with one gather, and a
// lane needs a fresh word for 2/3 of its pairs -- a third of the probes gone, with ONE insertion per (k-1)-mer: the filter keeps
// its size (2 MiB for the wheat-like label set: it must stay inside an XCD's L2 next to the table lines streaming through) and
// its false-positive rate.  (The alternative of the round-5 review -- an 8-byte block addressed by the core two pairs of a quad
// share, every (k-1)-mer under both cores -- halves the gathers but doubles the entries: 34 % false-positive quads at 2 MiB, and
// at 4 MiB the filter falls out of the L2 and the gain is gone: profiles/r06_notes.md, section 3.)  The minimum of two uniform
// hashes is not uniform (density 2 (1 - u)); 1 - (1 - u)^2 is, and that is what indexes the words, so the fill stays even.
// The three bits inside the word still come from the hashed (k-1)-mer.  k < 5 (cores of fewer than two bases): the old addressing.
#define MAP_BLOOM_CORE 0x100          // flag in the `nbits` argument of everything below (ctx->bloom_bits carries it)
#define MAP_BLOOM_NBITS(nb) ((nb) & 0xFF)
struct map_bloom_probe {
    uint32_t word, bits;
};
__host__ __device__ __forceinline__ uint32_t map_core_hash(uint64_t t) {       // of a canonical (k-3)-mer; well mixed in its HIGH bits
    uint32_t x = (uint32_t)t ^ (uint32_t)(t >> 32);
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    return x * 0x85EBCA6Bu;      // (multiply - xorshift - multiply: the scan computes one of these per pair)
}
__host__ __device__ __forceinline__ uint32_t map_core_word(uint32_t hmin /* the smaller of a (k-1)-mer's two core hashes */, int nbits) {
    const uint32_t n = ~hmin;
    const uint32_t sq = (uint32_t)(((unsigned long long)n * (unsigned long long)n) >> 32);      // (1 - u)^2
    return (~sq) >> (32 - (MAP_BLOOM_NBITS(nbits) - 5));
}
__host__ __device__ __forceinline__ uint32_t map_bloom_bits3(uint64_t x64, uint32_t &h) {      // the three bits of a canonical (k-1)-mer
    const uint32_t x = (uint32_t)x64 ^ (uint32_t)(x64 >> 32);
    h = x * 0x9E3779B1u;
    const uint32_t h2 = x * 0x85EBCA6Bu;
    return (1u << ((h >> 7) & 31u)) | (1u << ((h >> 2) & 31u)) | (1u << (h2 >> 27));
}
// word + bits of the CANONICAL (k-1)-mer x64 of a k-mer pair (generic form: the filter build and the rolling scans; the rolled
// scans of sp_map.hip / sp_sparse.hip compute the same from the windows they already hold)
__host__ __device__ __forceinline__ map_bloom_probe map_bloom(uint64_t x64, int k, int nbits) {
    map_bloom_probe p;
    uint32_t h;
    p.bits = map_bloom_bits3(x64, h);
    if (!(nbits & MAP_BLOOM_CORE) || k < 5) {
        p.word = h >> (32 - (MAP_BLOOM_NBITS(nbits) - 5));
        return p;
    }
    const int cb = 2 * (k - 3);
    const uint64_t smask = (1ULL << cb) - 1ULL;
    const uint64_t a = x64 >> 4, b = x64 & smask;               // first / last k-3 bases
    const uint64_t ar = sp_revcomp(a, k - 3), br = sp_revcomp(b, k - 3);
    const uint32_t ha = map_core_hash(a < ar ? a : ar), hb = map_core_hash(b < br ? b : br);
    p.word = map_core_word(ha < hb ? ha : hb, nbits);
    return p;
}
__device__ __forceinline__ bool map_bloom_test(const uint32_t *__restrict__ bloom, int nbits, int k, uint64_t x) {
    const map_bloom_probe p = map_bloom(x, k, nbits);
    return (bloom[p.word] & p.bits) == p.bits;
}

// Walk one unit and call hit(start, fwd, rc) for every VALID start whose pair passes the filter.
template <typename KeyT, typename KP, typename F>
__device__ __forceinline__ void map_pair_scan(const uint32_t *__restrict__ pk, const uint32_t *__restrict__ nm,
                                              int64_t s0, const KP &kp, const uint32_t *__restrict__ bloom,
                                              int nbits, F &&hit) {
    bool pairhit = false;
    const KeyT m1mask = (KeyT)(kp.kmask >> 2);
    sp_scan_unit_all<SP_UNIT, KeyT>(pk, nm, s0, kp, [&](int64_t start, KeyT fwd, KeyT rc, bool valid_k, bool valid_k1) {
        if (!(start & 1)) {   // first of the pair: screen both members through their shared (k-1)-mer
            pairhit = false;
            if (valid_k1) {
                const KeyT mf = fwd & m1mask, mr = rc >> 2;
                pairhit = map_bloom_test(bloom, nbits, kp.k, (uint64_t)(mf < mr ? mf : mr));
            }
        }
        if (pairhit && valid_k) hit(start, fwd, rc);
    });
}

#define MAP_PAIR_MAX_SG 7

struct map_pair_loc {
    uint32_t idx;   // canonical (k-1)-mer
    int field;
};
// where the k-mer `o` (any orientation; k >= 1) lives when it is looked up through its prefix / suffix (k-1)-mer
__host__ __device__ __forceinline__ map_pair_loc map_pair_loc_prefix(uint64_t o, int k) {
    map_pair_loc r;
    const uint32_t b = (uint32_t)(o & 3ULL);
    if (k == 1) { r.idx = 0; r.field = 4 + (int)b; return r; }
    const uint64_t p = o >> 2, pc = sp_revcomp(p, k - 1);
    if (p <= pc) { r.idx = (uint32_t)p; r.field = 4 + (int)b; }       // x + b
    else { r.idx = (uint32_t)pc; r.field = 3 - (int)b; }              // rc: comp(b) + rc(x)
    return r;
}
__host__ __device__ __forceinline__ map_pair_loc map_pair_loc_suffix(uint64_t o, int k) {
    map_pair_loc r;
    const uint32_t b = (uint32_t)(o >> (2 * (k - 1))) & 3u;
    if (k == 1) { r.idx = 0; r.field = (int)b; return r; }
    const uint64_t m1mask = (1ULL << (2 * (k - 1))) - 1ULL;
    const uint64_t x = o & m1mask, xc = sp_revcomp(x, k - 1);
    if (x <= xc) { r.idx = (uint32_t)x; r.field = (int)b; }           // b + x
    else { r.idx = (uint32_t)xc; r.field = 7 - (int)b; }              // rc: rc(x) + comp(b)
    return r;
}

// ----------------------------------------------------------------- compact exact pair table (S <= 3; round 5: QUAD buckets)
// The direct pair table is 4^(k-1) words (1 GiB at k = 15) for a few million used entries, and every gather from it
// leaves the chip's caches (54 G/s).  Round 4 put the entries into 2^23 buckets of two tagged words behind a bijective
// mix of the (k-1)-mer (64 MB, served by the Infinity Cache: one 8-byte load per candidate PAIR, 51 -> 46 ms per
// wheat-like pass).  What is left of the map stage after the filter probes is exactly those look-ups (1.4 G per pass at
// 12.8 ns per G), and labelled k-mers come in runs: the two pairs of a QUAD of starts 4g .. 4g+3 are almost always
// candidates together.  Their (k-1)-mers x1 (at 4g+1) and x2 (at 4g+3) overlap in the (k-3)-mer s = suffix of x1 =
// prefix of x2, so the table is now addressed by s: bucket = top bits of a bijective mix of the CANONICAL t = min(s,
// rc(s)), FOUR tagged 32-bit entries per 16-byte bucket, and ONE 16-byte load answers all four starts of the quad.
// An entry belongs to a (k-1)-mer y read in the orientation in which its (k-3)-mer at one end is canonical:
//     side 0 ("L"):  y = e + t      side 1 ("R"):  y = t + e        e = the two bases beyond t (4 bits)
// and holds the labels of the eight k-mers b + y and y + b in THAT orientation (field b / 4 + b, as in the direct
// table).  Every (k-1)-mer of a labelled k-mer is entered twice -- under its prefix and under its suffix (k-3)-mer --
// because which of the two a genome position shares with its neighbour pair depends on the position modulo 4.
// Entry: tag << 25 | overflow flag << 24 | eight 3-bit fields (label 1..3, "seen" in the third bit); tag = the low tb
// <= 2 bits of mix(t) << 5 | side << 4 | e; bucket + tag determine (t, side, e): a tag match is an exact match.
// 0 = empty.  A key that finds its bucket full goes to a small open-addressing overflow table of {key id + 1, fields}
// words and raises the flag in the bucket's first entry.
#define MAP_CT_FIELD 3
#define MAP_CT_ANY 0x6DB6DBu          // the label bits of all eight fields
#define MAP_CT_PAYLOAD 0xFFFFFFu
#define MAP_CT_OVF (1u << 24)
#define MAP_CT_MAX_TAG_BITS 2         // bits of mix(t) in the tag (next to side and e)
struct map_ptab {
    uint32_t *direct;                 // 4^(k-1) words, 4-bit fields (S <= 7), or NULL
    uint4 *buckets;                   // compact table: four entries per bucket
    unsigned long long *ovf;          // overflow table: (key id + 1) << 32 | fields, 0 = empty
    uint32_t ovf_mask;                // its size - 1
    int sb, tb;                       // bits of the shared (k-3)-mer 2(k-3), bits of mix(t) kept in the tag
};
__host__ __device__ __forceinline__ uint32_t map_ct_mix(uint32_t x, int kb) {
    const uint32_t m = kb >= 32 ? 0xFFFFFFFFu : ((1u << kb) - 1u);
    uint32_t h = (x * 0x9E3779B1u) & m;        // odd multiplier: a bijection of the kb-bit values
    h ^= h >> ((kb + 1) / 2);                  // xorshift by at least half the width: an involution-like bijection
    h = (h * 0x85EBCA6Bu) & m;
    h ^= h >> ((kb + 1) / 2);
    return h;
}
__host__ __device__ __forceinline__ uint32_t map_ct_ovf_home(uint32_t x) { return (x * 0xC2B2AE35u) >> 7; }
struct map_ct_key {
    uint32_t bucket, tag, kid;        // kid: (t, side, e) packed, the overflow table's key
};
__host__ __device__ __forceinline__ map_ct_key map_ct_key_of(const map_ptab &T, uint32_t t, uint32_t side, uint32_t e) {
    const uint32_t h = map_ct_mix(t, T.sb);
    map_ct_key r;
    r.bucket = h >> T.tb;
    r.tag = ((h & ((1u << T.tb) - 1u)) << 5) | (side << 4) | e;
    r.kid = (t << 5) | (side << 4) | e;
    return r;
}
// the eight fields of a key (0: absent) and where they live (for the "seen" mark): bucket * 4 + entry, or
// 0x80000000 | overflow slot.  `B` = the key's bucket, already loaded.
struct map_ct_hit {
    uint32_t fields, loc;
};
__device__ __forceinline__ map_ct_hit map_ct_find(const map_ptab &T, const uint4 &B, const map_ct_key &q) {
    map_ct_hit r;
    const uint32_t w[4] = {B.x, B.y, B.z, B.w};
    r.fields = 0;
    r.loc = 4u * q.bucket;
#pragma unroll
    for (int i = 3; i >= 0; i--)
        if ((w[i] >> 25) == q.tag && (w[i] & MAP_CT_PAYLOAD)) {
            r.fields = w[i] & MAP_CT_PAYLOAD;
            r.loc = 4u * q.bucket + (uint32_t)i;
        }
    if (!r.fields && (B.x & MAP_CT_OVF)) {         // rare: the bucket overflowed and the key is in none of its entries
        uint32_t i = map_ct_ovf_home(q.kid) & T.ovf_mask;
        // bounded like map_ct_insert's loop: a table that ended exactly full has no empty slot to stop an absent key
        // (advisor r04; sp_labels_set also keeps the table at most half full)
        for (uint32_t probes = 0; probes <= T.ovf_mask; probes++) {
            const unsigned long long o = T.ovf[i];
            if (o == 0ULL) break;
            if ((uint32_t)(o >> 32) == q.kid + 1u) {
                r.fields = (uint32_t)o & MAP_CT_PAYLOAD;
                r.loc = 0x80000000u | i;
                break;
            }
            i = (i + 1u) & T.ovf_mask;
        }
    }
    return r;
}
__device__ __forceinline__ void map_ct_mark(const map_ptab &T, uint32_t loc, uint32_t bits) {
    if (loc & 0x80000000u) atomicOr(&T.ovf[loc & 0x7fffffffu], (unsigned long long)bits);
    else atomicOr(reinterpret_cast<uint32_t *>(T.buckets) + loc, bits);
}
// The (up to four) table keys under which the k-mer `o` (ONE orientation, as given) is entered / found: its prefix and
// its suffix (k-1)-mer, each under its prefix and its suffix (k-3)-mer -- but only where that (k-3)-mer is canonical AS
// READ in this orientation (u <= rc(u)); the other orientation of the k-mer supplies the rest, and a palindromic
// (k-3)-mer is entered from both (the scan meets it with either flank on either side).  k >= 4.
struct map_ct_site {
    uint32_t t, side, e;
    int field;
};
__host__ __device__ __forceinline__ int map_ct_sites(uint64_t o, int k, map_ct_site out[4]) {
    const int sb = 2 * (k - 3);
    const uint64_t m1mask = (1ULL << (2 * (k - 1))) - 1ULL, smask = (1ULL << sb) - 1ULL;
    const uint64_t x[2] = {o >> 2, o & m1mask};                                  // prefix / suffix (k-1)-mer
    const int field[2] = {4 + (int)(o & 3ULL), (int)((o >> (2 * (k - 1))) & 3ULL)};      // o = x + b  /  o = b + x
    int n = 0;
    for (int i = 0; i < 2; i++) {
        const uint64_t u1 = x[i] >> 4, u2 = x[i] & smask;
        if (u1 <= sp_revcomp(u1, k - 3)) {       // x = u1 + e: side R
            out[n].t = (uint32_t)u1; out[n].side = 1u; out[n].e = (uint32_t)(x[i] & 15ULL); out[n].field = field[i];
            n++;
        }
        if (u2 <= sp_revcomp(u2, k - 3)) {       // x = e + u2: side L
            out[n].t = (uint32_t)u2; out[n].side = 0u; out[n].e = (uint32_t)(x[i] >> sb); out[n].field = field[i];
            n++;
        }
    }
    return n;
}

// ----------------------------------------------------------------- K5
// One block-iteration covers MAP_UNITS_PER_BLOCK units of 64 starts.  Hits are
// accumulated in an LDS histogram over the (few) output slots the range
// touches and flushed with one global atomic per non-zero entry.
#ifndef MAP_GRID_MULT
#define MAP_GRID_MULT 16
#endif
#ifndef MAP_BLOCK
#define MAP_BLOCK 512   // rounds 2-5 (one-phase walks): 256 -> 70.5 ms, 512 -> 68.3, 768 -> 66.2, 1024 -> 73.8.  Round 6 (two-phase walk,
                        // bound by waiting, not by instruction issue): 768 x 2 workgroups per CU (6 waves per SIMD) 34.5 ms,
                        // 512 x 4 (8 per SIMD, 64 VGPRs) 32.6, 256 x 8 33.0, 384 x 5 35.4, 1024 x 2 33.6
#endif
#define MAP_RANGE (MAP_BLOCK * SP_UNIT)  // starts per block iteration
// ---- LDS of the two-phase walks over a unit (map_unit_scan64 in sp_map.hip, map_unit_scan64_h in sp_sparse.hip)
#ifndef MAP2_P2
#define MAP2_P2 1      // queue entries per lane and iteration of the candidate phase (2: 34.5 against 34.3 ms -- the wait is not per wave)
#endif
struct map_unit_lds {
    uint32_t *words;                 // [5 + 2 or 6 + 2][MAP_BLOCK]: the unit's packed words (LSB-first; the MSB-first twin is derived), then its countable starts
    unsigned long long *planes;      // [2 or 3][MAP_BLOCK]: the label planes of every thread's unit
    uint16_t *queue;                 // [MAP_BLOCK / 64][MAP_QCAP]: the wave's candidate quads
};
#ifndef MAP2_ROUND_W
#define MAP2_ROUND_W 2      // 16-start words of every lane per round of the two phases (compact table; 4 = the whole unit: one
                            // partly filled iteration fewer per unit, but the queue's LDS then allows three workgroups per CU, not four)
#endif
#define MAP_QCAP(TABLE) ((TABLE) ? 256 * MAP2_ROUND_W : 512)      // the whole unit / two of the four 16-start words of every lane per round
#define MAP_UNIT_LDS_DECL(TABLE, WORDS)                                                   \
    __shared__ uint32_t s_uw[(WORDS) * MAP_BLOCK];                                        \
    __shared__ unsigned long long s_up[((TABLE) ? 2 : 3) * MAP_BLOCK];                    \
    __shared__ uint16_t s_uq[(MAP_BLOCK / 64) * MAP_QCAP(TABLE)];                         \
    const map_unit_lds ulds = {s_uw, s_up, s_uq}

#ifndef MAP_LDS_ENTRIES
#define MAP_LDS_ENTRIES 1024      // (4096 until round 6: the LDS went to the walk's candidate queue; a range of 49 152 starts needs
                                   // ~ 10 x S entries at 10-kb bins, 1024 cover bins down to ~150 bases at S = 3 -- below that: global atomics)
#endif

struct sp_map_params {
    int64_t n_units;
    int64_t bin_size;
    int64_t chunk_size;
    int64_t nslots;
    int S;
    int use_lds;
};

__device__ __forceinline__ int64_t map_slot(int64_t s, const sp_map_params &P, int k) {
    int64_t chunk = 0;
    if (P.chunk_size > 0 && s >= P.chunk_size - (k - 1)) chunk = (s + (k - 1)) / P.chunk_size;
    return s / P.bin_size + chunk;
}

// first start after `s` whose output slot differs from map_slot(s) (next bin or next chunk boundary)
__device__ __forceinline__ int64_t map_slot_end(int64_t s, const sp_map_params &P, int k) {
    int64_t e = (s / P.bin_size + 1) * P.bin_size;
    if (P.chunk_size > 0) {
        const int64_t chunk = (s >= P.chunk_size - (k - 1)) ? (s + (k - 1)) / P.chunk_size : 0;
        const int64_t c = (chunk + 1) * P.chunk_size - (k - 1);
        e = c < e ? c : e;
    }
    return e;
}

// feature mode: per-feature totals.  The features lie back to back in one packed sequence (no
// separators): a k-mer belongs to feature f iff it lies entirely inside [foff[f], foff[f+1]).  The
// feature of a unit's first hit is found by binary search, later hits of the unit walk forward.
struct map_feat_cursor {
    int64_t f, next;   // current feature and foff[f + 1]; f < 0: not located yet
};
__device__ __forceinline__ bool map_feat_locate(map_feat_cursor &c, int64_t start, int k,
                                                const int64_t *__restrict__ foff, int64_t n_feat) {
    if (c.f < 0) {
        int64_t lo = 0, hi = n_feat;   // last f with foff[f] <= start
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (foff[mid] <= start) lo = mid;
            else hi = mid;
        }
        c.f = lo;
        c.next = foff[lo + 1];
    }
    while (start >= c.next && c.f + 1 < n_feat) {
        c.f++;
        c.next = foff[c.f + 1];
    }
    return start + k <= c.next;        // false: the k-mer runs into the next feature
}

