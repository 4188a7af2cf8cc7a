#!/usr/bin/env python3
"""Feature-mode mapping + enrichment at scale (BASELINE config 5): N features of length
U[0.2, 10] kb cut from the synthetic genome, written as a FASTA with ids `chrom:start-end`, then
Seqs.map_kmer3(chunk=False) + Stats.enrich_ltr exactly as the CLI's -custom_features step runs them.
usage: feat_bench.py [config=peanut] [n_features=200000] [workdir=/tmp/sp_feat]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from subphaser_amd import _native, cluster, runtime, seqs as Seqs, stats as Stats
from subphaser_amd.hotpath import HotPath
from subphaser_amd.synth import SynthGenome

cfg = sys.argv[1] if len(sys.argv) > 1 else "peanut"
n_feat = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200000
work = sys.argv[3] if len(sys.argv) > 3 else "/tmp/sp_feat"
os.makedirs(work, exist_ok=True)
gen = SynthGenome(cfg)
ctx = _native.Context(0)
runtime.set_context(ctx)
C, S, k = len(gen.chroms), gen.S, 15
d_ascii, host = [], []
for c in gen.chroms:
    p = ctx.dev_alloc(c["length"])
    ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], S, c["chrom_id"], c["exchange"])
    d_ascii.append(p)
hp = HotPath(ctx, gen.labels, [c["length"] for c in gen.chroms], gen.sgs, k=k)
r1 = hp.count_and_filter(d_ascii)


class _Mat:
    pass
mat = _Mat()
mat.labels, mat.keys, mat.k = gen.labels, r1.keys, k
mat.freqs = r1.counts.astype(np.float64) / np.asarray(r1.kmer_lengths, np.float64)
cl = cluster.Cluster(mat, n_clusters=S, sg_assigned=gen.sg_assigned)
labels = cl.output_kmers(open(os.devnull, "w"), max_pval=0.05)
print("labels: %d significant k-mers" % len(labels.keys))

# ---- features (untimed): intervals of U[0.2, 10] kb, seed 4, round-robin over chromosomes
t0 = time.perf_counter()
rng = np.random.RandomState(4)
fa = os.path.join(work, "features.fa")
bed = os.path.join(work, "features.bed")
per = -(-n_feat // C)
total_bp = 0
with open(fa, "wb") as out, open(bed, "wb") as outb:
    for ci, c in enumerate(gen.chroms):
        seq = ctx.dev_to_host(d_ascii[ci], c["length"])
        n = min(per, n_feat - ci * per)
        if n <= 0:
            break
        ln = rng.randint(200, 10001, size=n)
        st = rng.randint(0, c["length"] - 10001, size=n)
        parts = []
        for s_, l_ in zip(st.tolist(), ln.tolist()):
            parts.append(b">%s:%d-%d\n" % (c["label"].encode(), s_ + 1, s_ + l_))
            parts.append(seq[s_:s_ + l_].tobytes())
            parts.append(b"\n")
        out.write(b"".join(parts))
        outb.write(b"".join(b"%s\t%d\t%d\n" % (c["label"].encode(), s_, s_ + l_) for s_, l_ in zip(st.tolist(), ln.tolist())))
        total_bp += int(ln.sum())
print("features: %d records, %.2f Gbases, FASTA %.1f MB written in %.1f s"
      % (n_feat, total_bp / 1e9, os.path.getsize(fa) / 1e6, time.perf_counter() - t0))

d_sg = {c["label"]: gen.sg_assigned[c["label"]] for c in gen.chroms} if isinstance(gen.sg_assigned, dict) else {}
sg_names = sorted(set(d_sg.values())) or ["SG%d" % (i + 1) for i in range(S)]
t0 = time.perf_counter()
feat_map = os.path.join(work, "custom.bin.count")
rows = []
with open(feat_map, "w") as fout:
    Seqs.map_kmer3([fa], labels, fout=fout, k=k, bin_size=10000000, sg_names=sg_names, chunk=False, ctx=ctx, collect=rows)
t1 = time.perf_counter()
from subphaser_amd import circos as Circos
names, code = Circos.factorize_first([x for part in rows for x in part[0]])      # as Pipeline.stage_features does
bins, counts = Circos.stack_arrays(names, code, np.concatenate([p[1] for p in rows]),
                                   np.concatenate([p[2] for p in rows], axis=0), window_size=100000000)
t15 = time.perf_counter()
with open(os.path.join(work, "custom.enrich"), "w") as fout:
    d_enriched, _ = Stats.enrich_ltr(fout, d_sg, counts, colnames=sg_names, rownames=bins, max_pval=0.05)
t2 = time.perf_counter()
print("stack_matrix: %.2f s, %d rows, %d significant" % (t15 - t1, len(bins), len(d_enriched)))
print("map_kmer3(chunk=False): %.2f s   stack+enrich_ltr: %.2f s   -> %.3f M features/s, %.3f Gbases/s end to end"
      % (t1 - t0, t2 - t1, n_feat / (t2 - t0) / 1e6, total_bp / (t2 - t0) / 1e9))
print("outputs: %d + %d bytes" % (os.path.getsize(feat_map), os.path.getsize(os.path.join(work, "custom.enrich"))))

# ---- the same features as BED intervals over the resident genome (sp_map_intervals): no sequence upload
t0 = time.perf_counter()
rows_b = []
with open(os.path.join(work, "custom_bed.bin.count"), "w") as fout:
    Seqs.map_intervals([bed], labels, {c["label"]: i for i, c in enumerate(gen.chroms)}, fout=fout, k=k, bin_size=10000000,
                       sg_names=sg_names, ctx=ctx, collect=rows_b)
t1 = time.perf_counter()
rows_iv = Seqs.IntervalRows.concat([p_[0] for p_ in rows_b])
counts_b = np.concatenate([p_[1] for p_ in rows_b], axis=0)
with open(os.path.join(work, "custom_bed.enrich"), "w") as fout:
    sg_idx, _ = Stats.enrich_ltr(fout, d_sg, counts_b, colnames=sg_names, rownames=rows_iv, max_pval=0.05, as_arrays=True)
t2 = time.perf_counter()
same = len(counts_b) == len(counts) and (np.asarray(counts_b) == np.asarray(counts)).all()
print("BED intervals: map_intervals %.2f s   enrich_ltr %.2f s   -> %.3f M features/s, %.2f s end to end; %d rows, %d "
      "significant; per-feature totals equal to the FASTA path: %s"
      % (t1 - t0, t2 - t1, n_feat / (t2 - t0) / 1e6, t2 - t0, len(rows_iv), int((sg_idx >= 0).sum()), same))
