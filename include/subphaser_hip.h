/*
 * subphaser_hip.h -- C-ABI of libsubphaser_hip.so: the MI355X (gfx950) native
 * implementation of SubPhaser's k-mer hot path.
 *
 * The reference has no FFI; its seam is five Python call sites in
 * Pipeline.run() (subphaser/__main__.py:403-404, 409-433, 484-486, 491,
 * 497-498).  Each entry point below names the reference interface it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SP_E* code on failure;
 *     the message is available from sp_last_error(ctx) (ctx may be NULL for
 *     errors raised before a context exists).  No C++ exception crosses.
 *   - pointers named d_* are DEVICE pointers (HIP), all others are host.
 *   - k-mers are uint64: 2 bits per base (A=0 C=1 G=2 T=3), first base in the
 *     most significant used bits, always the canonical orientation
 *     (min of the k-mer and its reverse complement).
 *   - the caller owns every output buffer; the library owns device memory
 *     inside the context; no caller pointer is retained after return.
 *   - one context per process per GPU; calls on a context are serialised by
 *     the caller.
 */
#ifndef SUBPHASER_HIP_H
#define SUBPHASER_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_EINVAL (-1)    /* bad argument / call order                     */
#define SP_EUNSUP (-2)    /* k outside the supported range                 */
#define SP_ENOMEM (-3)    /* hipMalloc failed                              */
#define SP_EHIP (-4)      /* any other HIP runtime error                   */
#define SP_ENODEV (-5)    /* no usable gfx950 device                       */
#define SP_ESTATE (-6)    /* reference-level precondition (message mirrors
                             the reference's ValueError text)              */
#define SP_EIO (-7)       /* write() to the caller's descriptor failed
                             (errno in sp_last_error(NULL))                */

typedef struct sp_ctx sp_ctx;

int sp_version(void);
const char *sp_last_error(const sp_ctx *ctx);

/* stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL
 * to let the library create its own. */
int sp_ctx_create(int device, void *stream, sp_ctx **out);
int sp_ctx_destroy(sp_ctx *ctx);
int sp_sync(sp_ctx *ctx);
/* the hipStream_t every kernel of this context is launched on */
void *sp_stream(sp_ctx *ctx);

/* ---- genome ingest (K0) -------------------------------------------------
 * Replaces the per-chromosome FASTA files written by Seqs.split_genomes
 * (Seqs.py:27-71) and read by jellyfish / chunk_chromfiles: the FASTA body of
 * chromosome `chrom` (newlines removed, any case, IUPAC allowed) is packed to
 * 2 bits/base + 1 validity bit/base in HBM.                                */
int sp_genome_reset(sp_ctx *ctx, int n_chrom);
int sp_genome_add(sp_ctx *ctx, int chrom, const uint8_t *ascii, int64_t len);
int sp_genome_add_device(sp_ctx *ctx, int chrom, const uint8_t *d_ascii, int64_t len);
int sp_genome_len(sp_ctx *ctx, int chrom, int64_t *len);
/* unpack back to upper-case ASCII ('N' for every invalid base) -- debugging / tests */
int sp_genome_unpack(sp_ctx *ctx, int chrom, uint8_t *ascii_out, int64_t len);

/* ---- k-mer counting (K1 + K2) -------------------------------------------
 * Replaces run_jellyfish_dumps (Jellyfish.py:671-704): for every chromosome,
 * canonical k-mer counts, kept when count >= lower_count (`jellyfish dump -L`).
 * engine: 0 = auto, 1 = global-atomic table, 2 = LDS radix-partition counter (byte tables), 3 = the same partition
 * chain ending in per-chromosome (slot, count) LISTS -- no byte tables; what auto picks for small genomes, where a
 * chromosome fills less than 1/3 of its dense table (k <= 15 with 2^17..2^31 slots, <= 64 chromosomes, whole-genome
 * sp_count calls only; sp_table_overflow / sp_filter_view / the table exchange are byte-table interfaces). */
int sp_count(sp_ctx *ctx, int k, int lower_count, int engine);
/* same for chromosomes [first, last) only (k <= 15): lets a multi-GPU caller ship the finished
 * table of chromosome i over xGMI while chromosome i+1 is being counted.          */
int sp_count_range(sp_ctx *ctx, int k, int lower_count, int engine, int first, int last);
/* Engine 2 sizes its partition buckets from a 1-in-16 sample of the chromosome and counts a chromosome again, with
 * exact sizes, when a bucket outgrows its region: *recounts = chromosomes counted twice since the context was
 * created (0 on ordinary genomes; results are identical either way -- this is a performance diagnostic). */
int sp_count_recounts(sp_ctx *ctx, int64_t *recounts);
/* number of slots of the dense count table for this k (2^(2k-1) for odd k, 4^k for even k) */
int sp_nslots(sp_ctx *ctx, int k, int64_t *nslots);
/* Count tables (k <= 15) are BYTE tables: one byte per dense slot holding the RAW count, saturated --
 * 0..254 = the count, 255 = "the count is >= 255: see the chromosome's overflow list" of
 * (uint32 slot, uint32 count) pairs, ascending slot.  lower_count (`jellyfish dump -L`,
 * Jellyfish.py:697) is applied wherever a table is read, so lengths, dumps and the matrix all see
 * m[key][c] = count if count >= lower_count else 0.  A byte table is also the multi-GPU wire format:
 * slot-range slices travel over xGMI as they are (a quarter of a u32 table).
 *   sp_tables_bind    use caller-owned device memory (nslots bytes, 16-byte aligned) as the table of
 *                     `chrom`, so that the caller (e.g. a torch tensor handed to RCCL) can exchange it;
 *                     NULL unbinds.  Call before sp_count.
 *   sp_table_overflow copy the overflow pairs of `chrom` into caller-owned DEVICE memory (capacity `cap`
 *                     pairs; d_pairs = NULL only queries *n_pairs); asynchronous on the context's stream. */
int sp_tables_bind(sp_ctx *ctx, int chrom, void *d_table);
int sp_table_overflow(sp_ctx *ctx, int chrom, void *d_pairs, int64_t cap, int64_t *n_pairs);
/* Multi-GPU, a chromosome counted in pieces by several ranks (the reference cuts chromosomes into 10-Mb
 * chunks with a k-1 overlap for the same purpose, Seqs.py:121-139): byte slices of the same slot range
 * [slot_base, slot_base + n) are added up exactly on the rank that filters that range.
 *   sp_table_merge    dst += src (device byte slices, dst may be read and written in place).  A slot whose sum
 *                     reaches 255 or whose summands were saturated becomes 255 and its exact sum (looked up in
 *                     the two overflow lists, absolute slots) goes to d_out_ovf -- a NEW list of the slots of this
 *                     range only, ascending (capacity `cap` pairs; SP_ENOMEM with *n_out = the number needed).
 *   sp_table_lengths  sum and number of the counts >= lower_count of a slice (the chromosome's contribution of this
 *                     slot range to `lengths`, Jellyfish.py:97,449, and to the dump size).                         */
int sp_table_merge(sp_ctx *ctx, void *d_dst_u8, const void *d_dst_ovf, int64_t n_dst_ovf, const void *d_src_u8,
                   const void *d_src_ovf, int64_t n_src_ovf, int64_t slot_base, int64_t n, void *d_out_ovf, int64_t cap,
                   int64_t *n_out);
int sp_table_lengths(sp_ctx *ctx, const void *d_tab_u8, const void *d_ovf, int64_t n_ovf, int64_t slot_base, int64_t n,
                     int lower_count, int64_t *sum, int64_t *n_dump);
/* lengths[c] = sum of the dumped counts of chromosome c (Jellyfish.py:97,449) */
int sp_lengths(sp_ctx *ctx, int64_t *lengths /*C*/);
/* jellyfish-dump equivalent of one chromosome.  Two calls: sp_dump_size then
 * sp_dump with buffers of that size.  Entries come in ascending order of the
 * internal dense slot (deterministic; jellyfish's own order is hash order);
 * the Python binding sorts them by canonical key.                           */
int sp_dump_size(sp_ctx *ctx, int chrom, int64_t *n);
int sp_dump(sp_ctx *ctx, int chrom, uint64_t *keys, uint32_t *counts, int64_t cap, int64_t *n);

/* ---- matrix + differential filter (K3) ----------------------------------
 * Replaces JellyfishDumps.to_matrix (Jellyfish.py:439-460) and .filter /
 * _filter_kmer (:462-512, :611-648).  Homoeologous sets in CSR form: set s
 * owns units [set_off[s], set_off[s+1]); unit u owns the chromosome indices
 * unit_chrom[unit_off[u] .. unit_off[u+1]).  min_freq / max_freq are the
 * already-resolved thresholds (min_prop/max_prop applied by the caller with
 * sp_lengths).  Outputs: n_union = len(d_mat), n_rows = differential k-mers,
 * n_hist = len(tot_freqs) (fold-passing k-mers, in or out of the freq range). */
/* Multi-GPU: make sp_filter / sp_filter_fetch work on a slot-range VIEW instead of the local
 * tables: C device pointers (16-byte aligned), each to the nslots_view table bytes of slots [slot_base,
 * slot_base + nslots_view) of one chromosome (all chromosomes of the genome, gathered from their
 * owner ranks), the chromosomes' overflow lists (d_ovf[c]: n_ovf[c] pairs with ABSOLUTE slots,
 * ascending; may cover more than the slot range; d_ovf = NULL: none) and the global `lengths`.
 * d_tabs = NULL returns to the local tables.                                                   */
int sp_filter_view(sp_ctx *ctx, int C, const void *const *d_tabs, int64_t slot_base,
                   int64_t nslots_view, const int64_t *lengths, int k, int lower_count,
                   const void *const *d_ovf, const int64_t *n_ovf);
int sp_filter(sp_ctx *ctx, int n_sets, const int32_t *set_off, const int32_t *unit_off,
              const int32_t *unit_chrom, double min_fold, int baseline, double min_freq,
              double max_freq, double ratio, int64_t *n_union, int64_t *n_rows, int64_t *n_hist);
/* rows in ascending dense-slot order (deterministic; the Python binding sorts by
 * canonical key); counts is row-major n_rows x C
 * (thresholded counts: 0 where count < lower_count); freqs = count/length in
 * fp64 exactly as Jellyfish.py:647 (may be NULL); tot = row sums (may be NULL) */
int sp_filter_fetch(sp_ctx *ctx, uint64_t *keys, uint32_t *counts, double *freqs, uint64_t *tot,
                    int64_t cap_rows);
/* the same rows, copied to PAGE-LOCKED host buffers (sp_host_alloc) by a copy stream while the calling stream is
 * free for the next stage; the buffers are valid after sp_filter_fetch_wait (k > 15: copies synchronously).
 * The emit buffers on the device are reused by the next sp_filter_fetch*: wait before calling it again.     */
int sp_filter_fetch_async(sp_ctx *ctx, uint64_t *keys, uint32_t *counts, uint64_t *tot, int64_t cap_rows);
int sp_filter_fetch_wait(sp_ctx *ctx);
/* same rows written to caller-owned DEVICE buffers (any may be NULL): multi-GPU callers gather
 * them over xGMI without a host round trip.  d_keys/d_tot: uint64 x n_rows, d_counts: uint32 x n_rows x C. */
int sp_filter_fetch_device(sp_ctx *ctx, void *d_keys, void *d_counts, void *d_tot, int64_t cap_rows);
/* tot of every fold-passing k-mer (the reference's tot_freqs histogram input) */
int sp_filter_hist(sp_ctx *ctx, uint64_t *tot, int64_t cap);

/* ---- subgenome-specific k-mer labels (K4) + bin mapping (K5) ------------
 * sp_labels_set replaces the d_kmers dict handed to Seqs.map_kmer3
 * (Cluster.py:174-175): canonical key -> subgenome index in [0, n_sg).
 * sp_map_bins replaces Seqs.map_kmer3 / map_kmer_each4 (Seqs.py:74-119,
 * 209-237) for one chromosome: counts by k-mer START position into
 *   slot(s) = s / bin_size + chunk(s)
 * where chunk(s) reproduces the reference's 10-Mb chunking (Seqs.py:121-139;
 * chunk_size = 0 disables it), so that a bin straddling a chunk boundary is
 * reported on two lines like the reference does.
 * slot_counts: nslots x n_sg int32 (overwritten).                          */
int sp_labels_set(sp_ctx *ctx, const uint64_t *keys, const uint8_t *sg, int64_t n, int n_sg);
/* the same with `keys` and `sg` already in DEVICE memory (e.g. the rows Cluster.output_kmers selected, kept where
 * sp_kmer_ttest tested them: Cluster.py:186-194 hands the reference's d_kmers dict to Seqs.map_kmer3 in memory too):
 * no host round trip; labels >= n_sg are detected on the device (SP_EINVAL)                                   */
int sp_labels_set_device(sp_ctx *ctx, const uint64_t *d_keys, const uint8_t *d_sg, int64_t n, int n_sg);
int sp_map_nslots(sp_ctx *ctx, int chrom, int64_t bin_size, int64_t chunk_size, int64_t *nslots);
int sp_map_bins(sp_ctx *ctx, int chrom, int64_t bin_size, int64_t chunk_size, int32_t *slot_counts,
                int64_t nslots, int64_t *n_mapped);
/* every chromosome in one call: chromosome c owns slots [slot_off[c], slot_off[c+1]) of
 * slot_counts (total x n_sg int32), sized with sp_map_nslots; n_mapped: C int64 (may be NULL).
 * One launch per chromosome back to back, one device->host copy.               */
int sp_map_bins_all(sp_ctx *ctx, int64_t bin_size, int64_t chunk_size, const int64_t *slot_off,
                    int32_t *slot_counts, int64_t *n_mapped);
/* window stack of the slot counts left on the device by the last sp_map_bins_all
 * (replaces Circos.stack_matrix, Circos.py:734-742, 831-842, without the text round trip):
 * window = (bin start) / window_size; chromosome c owns windows [win_off[c], win_off[c+1]),
 * at least len/window_size + 2 each; win_counts: total x n_sg int64 (overwritten).          */
int sp_stack_windows(sp_ctx *ctx, int64_t bin_size, int64_t chunk_size, int64_t window_size,
                     const int64_t *slot_off, const int64_t *win_off, int64_t *win_counts);
/* same into a caller-owned DEVICE table (total x n_sg uint64, ACCUMULATED: clear it first), for callers that
 * hold pieces of chromosomes: seg_start[i] = position of local chromosome i's base 0 inside the chromosome it is
 * a piece of (NULL = 0), win_off[i] = first window row of THAT chromosome.  Several ranks' tables add up
 * (all-reduce) to the whole-genome window table.                                                        */
int sp_stack_windows_dev(sp_ctx *ctx, int64_t bin_size, int64_t chunk_size, int64_t window_size,
                         const int64_t *slot_off, const int64_t *win_off, const int64_t *seg_start, void *d_win);
/* feature mode (map_kmer3(..., chunk=False), __main__.py:509-511): n_feat
 * sequences lying back to back in `ascii`, feature f = [off[f], off[f+1]).
 * Only k-mers that lie entirely inside one feature count (the kernel rejects
 * k-mers running across a boundary; no separator copy on the host).
 * counts: n_feat x n_sg int64 (whole-feature totals).                      */
int sp_map_features(sp_ctx *ctx, const uint8_t *ascii, const int64_t *off, int64_t n_feat,
                    int64_t *counts);
/* interval mode: BED-style features over the RESIDENT genome -- what `-custom_features` (__main__.py:509-517;
 * Seqs.map_kmer3(chunk=False), Seqs.py:228-244) computes from a FASTA of the feature sequences, without uploading
 * sequence that is already in HBM.  counts[i * n_sg + sg] = number of k-mer starts s with start[i] <= s and
 * s + k <= end[i] on chromosome chrom[i] whose canonical k-mer is labelled sg (0-based, half-open like BED).
 * Intervals may overlap or nest.  A labelled k-mer counts as "seen" (sp_labels_hit) only where some interval
 * contains it -- the same k-mers a FASTA of the features would contain.                                   */
int sp_map_intervals(sp_ctx *ctx, const int32_t *chrom, const int64_t *start, const int64_t *end, int64_t n,
                     int64_t *counts);
/* number of distinct labelled k-mers seen by sp_map_bins/sp_map_features since
 * sp_labels_set (the reference's "mapped kmers" log line, Seqs.py:109-117)  */
int sp_labels_hit(sp_ctx *ctx, int64_t *n_hit);

/* ---- per-window enrichment (K6) -----------------------------------------
 * Replaces Stats.enrich / _enrich / fisher_test / Pvalues.get_enriched
 * (Stats.py:14-31, 140-192).  counts: W x S int64 window rows (host).
 * Outputs (host): pvals W x S; argmin W; sig W; ratios W x S.              */
int sp_enrich(sp_ctx *ctx, const int64_t *counts, int64_t W, int S, double max_pval,
              double min_ratio, double *pvals, int32_t *argmin, uint8_t *sig, double *ratios);
/* counts in DEVICE memory (column totals are reduced on the device) */
int sp_enrich_dev(sp_ctx *ctx, const void *d_counts, int64_t W, int S, double max_pval, double min_ratio,
                  double *pvals, int32_t *argmin, uint8_t *sig, double *ratios);
/* sp_stack_windows + sp_enrich fused on the device (Circos.stack_matrix -> Stats.enrich without the text and
 * host round trips): every window row of win_off (empty rows included -- they change no total) is tested;
 * outputs are host arrays of win_off[C] rows.                                                            */
int sp_stack_enrich(sp_ctx *ctx, int64_t bin_size, int64_t chunk_size, int64_t window_size, const int64_t *slot_off,
                    const int64_t *win_off, double max_pval, double min_ratio, int64_t *win_counts, double *pvals,
                    int32_t *argmin, uint8_t *sig, double *ratios);

/* ---- subgenome-specific k-mer test ------------------------------------------
 * Replaces Cluster.output_kmers / _output_kmers for test_method = ttest_ind (Cluster.py:151-194): per
 * differential k-mer (row of `counts`, M x C thresholded counts as sp_filter_fetch returns them) the
 * frequencies count / lengths[c] are grouped by subgenome (group g owns the chromosome indices
 * group_chrom[group_off[g] .. group_off[g+1]), groups in sorted subgenome-name order), the groups are ordered
 * by mean (descending, ties in group order) and scipy.stats.ttest_ind(top, second) -- pooled variance,
 * two-sided -- gives pvals[row].  means: M x n_groups.  The caller keeps rows with !(p > max_pval).
 * `counts` may also be a device pointer (rows staged with sp_dev_copy_from_host earlier).                   */
int sp_kmer_ttest(sp_ctx *ctx, const uint32_t *counts, int64_t M, int C, const int64_t *lengths, int n_groups,
                  const int32_t *group_off, const int32_t *group_chrom, int32_t *top, int32_t *second,
                  double *pvals, double *means);

/* ---- multi-GPU, k > 15 ----------------------------------------------------------------------
 * Twin of sp_tables_bind / sp_filter_view for 64-bit keys ("for k > 16 the exchange
 * becomes key-partitioned").  After sp_count (k > 15) every local chromosome is a sorted list of
 * (canonical key, count >= lower_count) -- what jellyfish dump holds, Jellyfish.py:697-699.
 *   sp_sparse_sizes   n[c] = entries of local chromosome c.
 *   sp_sparse_sample  up to n_samples evenly spaced keys of one list (to choose common splitters).
 *   sp_sparse_split   bounds[0..n_split+1]: bounds[i+1] = first index whose key >= splitters[i];
 *                     bounds[0] = 0, bounds[n_split+1] = n.
 *   sp_sparse_export  copy entries [first, first+count) into caller-owned DEVICE buffers (uint64
 *                     keys, uint32 counts); asynchronous on the context's stream (sp_sync).
 *   sp_sparse_view    point sp_filter / sp_filter_fetch(_device) at caller-owned DEVICE lists, one per
 *                     chromosome of the whole genome (sorted keys of ONE key range, counts >= lower),
 *                     with the global `lengths` (Jellyfish.py:449); d_keys = NULL returns to the local
 *                     chromosomes.  to_matrix/filter semantics are unchanged (Jellyfish.py:439-512). */
int sp_sparse_sizes(sp_ctx *ctx, int64_t *n);
int sp_sparse_sample(sp_ctx *ctx, int chrom, int64_t n_samples, uint64_t *keys, int64_t *n_out);
int sp_sparse_split(sp_ctx *ctx, int chrom, const uint64_t *splitters, int n_split, int64_t *bounds);
int sp_sparse_export(sp_ctx *ctx, int chrom, int64_t first, int64_t count, void *d_keys, void *d_counts);
int sp_sparse_view(sp_ctx *ctx, int C, const void *const *d_keys, const void *const *d_counts, const int64_t *n,
                   const int64_t *lengths, int k, int lower_count);

/* ---- profiling: per-kernel HIP-event timing on the context's stream ------ */
int sp_prof_enable(sp_ctx *ctx, int on);
int sp_prof_reset(sp_ctx *ctx);
/* writes a JSON object {"kernel": {"calls": n, "ms": total}, ...} */
int sp_prof_report(sp_ctx *ctx, char *buf, int64_t cap);

/* ---- bench support (not part of the reference's interface) --------------
 * Deterministic synthetic chromosome written as ASCII into a device buffer. */
int sp_synth_chrom(sp_ctx *ctx, uint8_t *d_out, int64_t len, uint64_t seed, int set_id,
                   int sg_id, int n_sg, int chrom_id, int exchange);
/* positions [start, start + n) of the same chromosome (every base is a pure function of its position) */
int sp_synth_chrom_range(sp_ctx *ctx, uint8_t *d_out, int64_t len, int64_t start, int64_t n, uint64_t seed, int set_id,
                         int sg_id, int n_sg, int chrom_id, int exchange);
/* page-locked host memory: device->host copies into it run at PCIe speed; plain
 * (pageable) buffers are accepted everywhere but copy several times slower.     */
int sp_host_alloc(sp_ctx *ctx, int64_t bytes, void **h_ptr);
int sp_host_free(sp_ctx *ctx, void *h_ptr);
/* page-lock memory the caller already owns (e.g. a POSIX shared-memory segment the ranks of one node
 * assemble the matrix in: every rank copies ITS rows over ITS PCIe link); undone by sp_host_unregister */
int sp_host_register(sp_ctx *ctx, void *h_ptr, int64_t bytes);
int sp_host_unregister(sp_ctx *ctx, void *h_ptr);
/* device memory helpers so a non-torch caller can stage buffers */
int sp_dev_alloc(sp_ctx *ctx, int64_t bytes, void **d_ptr);
int sp_dev_free(sp_ctx *ctx, void *d_ptr);
int sp_dev_copy_to_host(sp_ctx *ctx, void *dst, const void *d_src, int64_t bytes);
int sp_dev_copy_from_host(sp_ctx *ctx, void *d_dst, const void *src, int64_t bytes);


/* ---- host-side FASTA scanner (no GPU work; replaces the Bio.SeqIO record loops of Seqs.py:27-71, 121-153) ----
 * `data`/`n`: the file image (mmap or decompressed bytes), which must stay valid until sp_fasta_close.  A record
 * starts at a '>' that is the first byte of a line; bytes <= 0x20 are dropped from the sequence lines.
 * sp_fasta_fetch: hdr_start[r] .. hdr_end[r] = header text of record r (without '>' and the line break),
 * seq_off[r] .. seq_off[r + 1] = its bases inside `cat` (n_bases bytes, caller-owned, may be page-locked). */
typedef struct sp_fasta sp_fasta;
int sp_fasta_open(const void *data, int64_t n, int threads, sp_fasta **out);
int sp_fasta_counts(const sp_fasta *h, int64_t *n_records, int64_t *n_bases);
int sp_fasta_fetch(const sp_fasta *h, int64_t *hdr_start, int64_t *hdr_end, int64_t *seq_off, void *cat);
void sp_fasta_close(sp_fasta *h);


/* ---- host-side text writers (no GPU work) --------------------------------------------------------------------
 * Rows of `.kmer.mat` (Jellyfish.py:515-520) and `.sig.kmer-subgenome.tsv` (Cluster.py:165-176) formatted by
 * threads of the calling process and written to the open file descriptor `fd` in row order; floats are printed as
 * Python's repr() prints them.  The caller writes the header line (and flushes) first.  names: n_names
 * '\0'-separated strings.  sp_text_repr: repr of x[i] back to back in out (>= 40 bytes per value), off[n + 1]. */
int sp_text_kmer_matrix(const uint64_t *keys, int k, const double *freqs, int64_t M, int C, int threads, int fd,
                        int64_t *bytes);
int sp_text_sig_kmers(const uint64_t *keys, int k, const int32_t *top, const char *names, int n_names,
                      const double *pvals, const double *means, int G, int64_t M, int threads, int fd, int64_t *bytes);
int sp_text_repr(const double *x, int64_t n, char *out, int64_t *off);

/* Generic tab-separated rows for the feature-scale outputs: `.custom.enrich` / `.ltr.enrich` (Stats.py:59-70) and
 * the feature-mode `.bin.count` lines (Seqs.py:228-244).  Row i = the columns joined by '\t' + '\n'.
 *   SP_COL_STR   data = char blob, off[M + 1]: the string of row i is data[off[i] .. off[i + 1])
 *   SP_COL_I64   data = int64 [M x width], printed in decimal, joined by `join`
 *   SP_COL_F64   data = double [M x width], printed as Python's repr(), joined by `join`
 *   SP_COL_NAME  data = int32 [M] indices into `names` (`width` strings; string j = names[off[j] .. off[j + 1]))
 *   SP_COL_IVAL  data = int64 [M x 3] (name index, start, end), printed `name:start-end` (ids of BED intervals) */
#define SP_COL_STR 0
#define SP_COL_I64 1
#define SP_COL_F64 2
#define SP_COL_NAME 3
#define SP_COL_IVAL 4
typedef struct sp_text_col {
    int kind;
    int width;
    char join;
    const void *data;
    const int64_t *off;
    const char *names;
} sp_text_col;
int sp_text_table(const sp_text_col *cols, int n_cols, int64_t M, int threads, int fd, int64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif
