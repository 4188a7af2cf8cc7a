// sp_c2batch.h -- one launch per kernel TYPE over all chromosomes of a small genome (round 5).
// A 20-Mb chromosome's counting chain is ten launches of 10-200 us with a dependency between each pair: 13 chains on four
// streams were 156 launches and ~3 ms per Arabidopsis-like pass, most of it launch latency and tails.  The batched path
// gives every kernel a second grid dimension: blockIdx.y = the chromosome, whose pointers and sizes come from a
// descriptor table -- nine launches per pass, every one with thirteen chromosomes' worth of blocks.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

struct c2_bdesc {
    const uint32_t *pk, *pm, *nm;
    int64_t n_units32, n_visit, n_tiles;      // units of 32 starts; units the histogram visits; part1 tiles
    int sample_shift, split;
    unsigned long long mult8, mult8_1, slack, slack1, pad1, sslack;
    unsigned long long *ghist, *off_fine, *off1, *tile_start, *cur1, *cur2;
    ulonglong2 *span;
    uint16_t *lo1;
    uint8_t *hi1;
    uint16_t *buf2;
    uint32_t *seg_base, *seg_cnt, *seg_off;
    uint2 *stage;                             // unordered (slot, count) pairs of the list counter
    unsigned long long stage_cap;
    unsigned long long *d_len4;               // [0] sum [1] n [2] pairs [3] region-overrun flag
    unsigned long long *out_keys;             // the chromosome's list
    uint32_t *out_cnts;
};
