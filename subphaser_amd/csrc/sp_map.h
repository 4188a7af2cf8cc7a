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
struct map_bloom_probe {
    uint32_t word, bits;
};
__host__ __device__ __forceinline__ map_bloom_probe map_bloom(uint64_t x64, int nbits) {
    const uint32_t x = (uint32_t)x64 ^ (uint32_t)(x64 >> 32);
    const uint32_t h = x * 0x9E3779B1u;
    const uint32_t h2 = x * 0x85EBCA6Bu;
    map_bloom_probe p;
    p.word = h >> (32 - (nbits - 5));
    p.bits = (1u << ((h >> 7) & 31u)) | (1u << ((h >> 2) & 31u)) | (1u << (h2 >> 27));
    return p;
}
__device__ __forceinline__ bool map_bloom_test(const uint32_t *__restrict__ bloom, int nbits, uint64_t x) {
    const map_bloom_probe p = map_bloom(x, nbits);
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
                pairhit = map_bloom_test(bloom, nbits, (uint64_t)(mf < mr ? mf : mr));
            }
        }
        if (pairhit && valid_k) hit(start, fwd, rc);
    });
}

#define MAP_PAIR_MAX_SG 7
#ifndef MAP_BATCH
#define MAP_BATCH 4     // pairs whose probes / gathers are in flight together (16 pairs per 32-start unit)
#endif
#ifndef MAP_BATCH_COMPACT
#define MAP_BATCH_COMPACT 2   // with the compact table (8-byte bucket loads, more live registers per pair): 2: 46.0, 4: 47.1, 8: 59.8 ms
#endif

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

// ----------------------------------------------------------------- compact exact pair table (S <= 3, round 4)
// The direct pair table is 4^(k-1) words (1 GiB at k = 15) for a few million used entries, and every gather from it
// leaves the chip's caches (54 G/s).  With the gathers confined to <= 128 MB the same kernel ran 43-44 instead of
// 50.8 ms per wheat-like pass (profiles/r03_notes.md): the 256-MB Infinity Cache serves them.  The tagged
// open-addressing table of round 3 lost that again to its probe loops.  This one answers with ONE 8-byte load and no
// loop: 2^bb buckets of two 32-bit entries; the canonical (k-1)-mer x goes through a BIJECTIVE mix of its 2(k-1) bits,
// the top bb bits of the mix choose the bucket, the remaining tb <= 7 bits are the entry's tag -- bucket and tag
// determine x, so a tag match is an exact match.  Entry: tag << 25 | overflow flag << 24 | eight 3-bit fields (label
// 1..3 in the low two bits, "seen" in the third; the field order of the direct table).  0 = empty (an entry has at
// least one label).  A key that finds both entries of its bucket taken goes to a small open-addressing overflow table
// of {x + 1, fields} words and raises the flag in the bucket's first entry: only a look-up that matches neither
// entry of a flagged bucket (about one in a thousand at the load chosen) probes further.
#define MAP_CT_FIELD 3
#define MAP_CT_ANY 0x6DB6DBu          // the label bits of all eight fields
#define MAP_CT_PAYLOAD 0xFFFFFFu
#define MAP_CT_OVF (1u << 24)
#define MAP_CT_MAX_TAG_BITS 7
struct map_ptab {
    uint32_t *direct;                 // 4^(k-1) words, 4-bit fields (S <= 7), or NULL
    uint2 *buckets;                   // compact table
    unsigned long long *ovf;          // overflow table: (x + 1) << 32 | fields, 0 = empty
    uint32_t ovf_mask;                // its size - 1
    int kb, tb;                       // key bits 2(k-1), tag bits
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
// look x up: the eight fields (0: absent) and where they live (for the "seen" mark): bucket * 2 + entry, or
// 0x80000000 | overflow slot
struct map_ct_hit {
    uint32_t fields, loc;
};
__device__ __forceinline__ map_ct_hit map_ct_lookup(const map_ptab &T, uint32_t x, bool want) {
    map_ct_hit r;
    r.fields = 0;
    r.loc = 0;
    const uint32_t h = map_ct_mix(x, T.kb), b = h >> T.tb, tag = h & ((1u << T.tb) - 1u);
    uint2 e = make_uint2(0u, 0u);
    if (want) e = T.buckets[b];
    const bool m0 = (e.x >> 25) == tag && (e.x & MAP_CT_PAYLOAD), m1 = (e.y >> 25) == tag && (e.y & MAP_CT_PAYLOAD);
    r.fields = m0 ? (e.x & MAP_CT_PAYLOAD) : (m1 ? (e.y & MAP_CT_PAYLOAD) : 0u);
    r.loc = 2u * b + (m1 ? 1u : 0u);
    if (!m0 && !m1 && (e.x & MAP_CT_OVF)) {        // rare: the bucket overflowed and x is in neither entry
        uint32_t i = map_ct_ovf_home(x) & T.ovf_mask;
        for (;;) {
            const unsigned long long o = T.ovf[i];
            if (o == 0ULL) break;
            if ((uint32_t)(o >> 32) == x + 1u) {
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

// ----------------------------------------------------------------- K5
// One block-iteration covers MAP_UNITS_PER_BLOCK units of 64 starts.  Hits are
// accumulated in an LDS histogram over the (few) output slots the range
// touches and flushed with one global atomic per non-zero entry.
#ifndef MAP_GRID_MULT
#define MAP_GRID_MULT 16
#endif
#ifndef MAP_BLOCK
#define MAP_BLOCK 768   // measured: 256 -> 70.5 ms, 512 -> 68.3, 768 -> 66.2, 1024 -> 73.8 (640 / 896: 76-79)
#endif
#define MAP_RANGE (MAP_BLOCK * SP_UNIT)  // starts per block iteration
#ifndef MAP_LDS_ENTRIES
#define MAP_LDS_ENTRIES 4096
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

