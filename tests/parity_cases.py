"""Parity checks shared by the CPU suite (oracle behind the host layer, `-m "not gpu"`)
and the GPU suite (libsubphaser_hip.so through the C-ABI, `-m gpu`).

Every check takes a context object (`OracleContext` or `_native.Context`) and
compares what the host-side mirror modules produce with it against the golden
vectors generated from the imported reference (tests/golden/gen_golden.py)."""
import io
import math

import numpy as np

import pyoracle as po
from subphaser_amd import circos, cluster, jellyfish, seqs, stats
from subphaser_amd import kmer as kmerlib

K, L = 15, 3


def register_toy(toy, prefix="/virtual/toy/"):
    """Make the toy chromosomes known to the host layer without touching disk."""
    files = []
    for lab in toy["labels"]:
        path = "%s%s.fasta" % (prefix, lab)
        if path not in seqs._REG:      # keep the record (it remembers which GPU slot holds it)
            seqs._REG[path] = seqs.ChromRecord(lab, toy["seqs"][lab].encode())
        files.append(path)
    return files


def count_toy(ctx, toy, engine=0):
    files = register_toy(toy)
    dumps = jellyfish.run_jellyfish_dumps(files, k=K, lower_count=L, ctx=ctx, engine=engine)
    return files, dumps


# ------------------------------------------------------------------ G1
def check_toy_dumps(ctx, golden, toy, engine=0):
    files, dumps = count_toy(ctx, toy, engine)
    g1 = golden["G1_toy_dumps"]
    assert g1["k"] == K and g1["lower"] == L
    for lab, d in zip(toy["labels"], dumps):
        keys, cnts = d.fetch()
        exp = g1["dumps"][lab]
        assert len(keys) == exp["n"], lab
        assert int(cnts.astype(np.int64).sum()) == exp["sum"], lab
        assert (int(np.bitwise_xor.reduce(keys)) if len(keys) else 0) == exp["xor"], lab
        assert [int(x) for x in keys[:8]] == exp["head_keys"], lab
        assert [int(x) for x in cnts[:8]] == exp["head_counts"], lab
        assert int(cnts.max()) == exp["max_count"], lab
        assert (np.diff(keys.astype(np.int64)) > 0).all()
        assert (cnts >= L).all()
        assert str(d).endswith("%s.fasta_%d.fa" % (lab, K))


# ------------------------------------------------------------------ G2 / G3
def check_filter_cases(ctx, golden, toy, only=None):
    files, dumps = count_toy(ctx, toy)
    for name, case in golden["G2_filter"].items():
        if only and name not in only:
            continue
        jd = jellyfish.JellyfishDumps(dumps, toy["labels"])
        d_mat = jd.to_matrix()
        exp = case["res"]
        assert jd.lengths == exp["lengths"], name
        try:
            d2 = jd.filter(d_mat, jd.lengths, case["sgs"], outfig="hist.png", **case["kw"])
        except ValueError as e:
            assert "error" in exp, (name, str(e))
            assert str(e) == exp["error"], name
            continue
        assert "error" not in exp, name
        assert len(d_mat) == exp["n_union"], name
        assert len(d2) == exp["n_rows"], name
        rows = [[km] + [repr(float(x)) for x in fr] for km, fr in d2.items()]
        assert rows == exp["rows"], name
        # counts / tot consistency
        assert (d2.counts.sum(axis=1) == d2.tot.astype(np.int64)).all()
        assert jd.n_hist >= len(d2)
        hist = np.sort(jd.hist_tot())
        assert len(hist) == jd.n_hist


def check_kmer_mat_text(ctx, golden, toy):
    files, dumps = count_toy(ctx, toy)
    case = golden["G2_filter"]["default"]
    jd = jellyfish.JellyfishDumps(dumps, toy["labels"])
    d2 = jd.filter(jd.to_matrix(), None, case["sgs"], outfig="h.png", **case["kw"])
    buf = io.StringIO()
    jd.write_matrix(d2, buf)
    assert buf.getvalue() == golden["G3_kmer_mat_text"]
    return d2


# ------------------------------------------------------------------ G7
def check_output_kmers(ctx, golden, toy):
    d2 = check_kmer_mat_text(ctx, golden, toy)
    g7 = golden["G7_output_kmers"]
    cl = cluster.Cluster(d2, n_clusters=2, sg_prefix="SG", sg_assigned=g7["sg_assigned"])
    assert dict(cl.d_sg) == g7["d_sg"]
    assert cl.sg_names == g7["sg_names"]
    buf = io.StringIO()
    labels = cl.output_kmers(buf, max_pval=0.05)
    assert len(labels) == g7["n_dkmers"]
    exp_lines = sorted(g7["text"].strip().split("\n")[1:])
    got_lines = sorted(buf.getvalue().strip().split("\n")[1:])
    assert len(exp_lines) == len(got_lines)
    for e, g in zip(exp_lines, got_lines):
        e, g = e.split("\t"), g.split("\t")
        assert e[:2] == g[:2]
        assert math.isclose(float(e[2]), float(g[2]), rel_tol=1e-9, abs_tol=1e-300)
        for a, b in zip(e[3].split(","), g[3].split(",")):
            assert math.isclose(float(a), float(b), rel_tol=1e-14, abs_tol=0)
    return cl, labels


# ------------------------------------------------------------------ G4
def check_map_cases(ctx, golden, toy):
    cl, labels = check_output_kmers(ctx, golden, toy)
    files = register_toy(toy)
    for name, case in golden["G4_map_kmer3"].items():
        kw = dict(case["kw"])
        buf = io.StringIO()
        seqs.map_kmer3(files, labels, fout=buf, k=K, sg_names=cl.sg_names, ctx=ctx, **kw)
        assert buf.getvalue() == case["text"], name
    return cl, labels


def check_map_features(ctx, golden, toy, tmp_path):
    cl, labels = check_output_kmers(ctx, golden, toy)
    case = golden["G4_map_features"]
    ff = tmp_path / "features.fa"
    with open(ff, "w") as f:
        for fid, s in case["features"]:
            f.write(">%s\n%s\n" % (fid, s))
    buf, rows = io.StringIO(), []
    seqs.map_kmer3([str(ff)], labels, fout=buf, k=K, bin_size=10000000, sg_names=cl.sg_names, chunk=False,
                   log=False, ctx=ctx, collect=rows)
    assert buf.getvalue() == case["text"]
    _check_collected(rows, buf.getvalue(), tmp_path, 100000000)


def _check_collected(rows, text, tmp_path, window_size):
    """the arrays map_kmer3 hands the CLI stack to the same windows as the text it wrote, parsed back"""
    import numpy as np
    from subphaser_amd import circos
    f = tmp_path / "collected.bin.count"
    f.write_text(text)
    names, code = circos.factorize_first([x for part in rows for x in part[0]])
    got = circos.stack_arrays(names, code, np.concatenate([p_[1] for p_ in rows]),
                              np.concatenate([p_[2] for p_ in rows], axis=0), window_size=window_size)
    assert got == circos.stack_matrix(str(f), window_size=window_size)
    assert len(got[0]) > 0


def check_dict_labels(ctx, golden, toy):
    """d_kmers given as the reference's dict (both orientations -> SG name)."""
    cl, labels = check_output_kmers(ctx, golden, toy)
    kmers = kmerlib.decode_many(labels.keys, K)
    rcs = kmerlib.decode_many(kmerlib.revcomp(labels.keys, K), K)
    d = {}
    for km, rc, s in zip(kmers, rcs, labels.sg_idx):
        d[km] = labels.sg_names[s]
        d[rc] = labels.sg_names[s]
    files = register_toy(toy)
    case = golden["G4_map_kmer3"]["chunk2000_bin100"]
    buf = io.StringIO()
    seqs.map_kmer3(files, d, fout=buf, k=K, sg_names=cl.sg_names, ctx=ctx, **case["kw"])
    assert buf.getvalue() == case["text"]


# ------------------------------------------------------------------ G5 (pure host)
def check_stack_matrix(golden, tmp_path):
    p = tmp_path / "toy.bin.count"
    p.write_text(golden["G4_map_kmer3"]["chunk2000_bin100"]["text"])
    for ws, exp in golden["G5_stack_matrix"].items():
        coords, counts = circos.stack_matrix(str(p), window_size=int(ws))
        assert [[c, s, e] for c, s, e in coords] == exp["coords"], ws
        assert [[int(x) for x in row] for row in counts] == exp["counts"], ws
    return p


# ------------------------------------------------------------------ G6
P_TOL = 1e-6      # BASELINE.json north_star: enrichment p-values within 1e-6


def _cmp_enrich_text(got, exp, float_cols, float_list_cols):
    g_lines, e_lines = got.strip("\n").split("\n"), exp.strip("\n").split("\n")
    assert g_lines[0] == e_lines[0]
    assert len(g_lines) == len(e_lines)
    for g, e in zip(g_lines[1:], e_lines[1:]):
        g, e = g.split("\t"), e.split("\t")
        assert len(g) == len(e)
        for i, (a, b) in enumerate(zip(g, e)):
            if i in float_cols:
                assert abs(float(a) - float(b)) <= P_TOL and math.isclose(float(a), float(b), rel_tol=1e-6, abs_tol=1e-290), (g, e)
            elif i in float_list_cols:
                for x, y in zip(a.split(","), b.split(",")):
                    assert abs(float(x) - float(y)) <= P_TOL and math.isclose(float(x), float(y), rel_tol=1e-6, abs_tol=1e-290), (g, e)
            else:
                assert a == b, (i, g, e)


def check_enrich_bin(ctx, golden, tmp_path):
    g6 = golden["G6_enrich"]
    p = check_stack_matrix(golden, tmp_path)
    tb = g6["toy_bin"]
    coords, counts = circos.stack_matrix(str(p), window_size=tb["window_size"])
    f1, f2 = io.StringIO(), io.StringIO()
    stats.enrich_bin(f1, f2, tb["d_sg"], counts, colnames=tb["sg_names"], rownames=coords, max_pval=0.05, ctx=ctx)
    # cols: 4 p_value, 8 pvals, 10 p_corrected are floating point; ratios (6) must match exactly
    _cmp_enrich_text(f1.getvalue(), tb["enrich_text"], {4, 10}, {8})
    assert f2.getvalue() == tb["group_text"]
    for name, t in g6["tables"].items():
        rows = [tuple(r) for r in t["rows"]]
        f1, f2 = io.StringIO(), io.StringIO()
        with np.errstate(all="ignore"):
            stats.enrich_bin(f1, f2, t["d_sg"], t["counts"], colnames=t["sg_names"], rownames=rows,
                             max_pval=0.05, ctx=ctx)
        _cmp_enrich_text(f1.getvalue(), t["enrich_text"], {4, 10}, {8})
        assert f2.getvalue() == t["group_text"], name


def check_enrich_features(ctx, golden, tmp_path):
    g6 = golden["G6_enrich"]
    p = tmp_path / "feat.bin.count"
    p.write_text(golden["G4_map_features"]["text"])
    fc, fcounts = circos.stack_matrix(str(p), window_size=100000000)
    tb = g6["toy_bin"]
    ok = [(c, n) for c, n in zip(fc, fcounts) if ":" in c[0]]
    f3 = io.StringIO()
    d_enriched, d_exchange = stats.enrich_ltr(f3, tb["d_sg"], [n for _, n in ok], colnames=tb["sg_names"],
                                              rownames=[c for c, _ in ok], max_pval=0.05, ctx=ctx)
    _cmp_enrich_text(f3.getvalue(), g6["toy_features"]["enrich_text"], {2, 5}, set())
    # ids that do not look like chrom:start-end: reference crashes, we report 'none'
    f4 = io.StringIO()
    d_enriched, d_exchange = stats.enrich_ltr(f4, tb["d_sg"], fcounts, colnames=tb["sg_names"], rownames=fc,
                                              max_pval=0.05, ctx=ctx)
    assert d_exchange.get("weird_id_without_coords", "none") == "none"


def check_fisher_cells_and_tails(ctx, golden):
    """fisher_test margins (x22 quirk, clamps) and hypergeometric tails vs 30-digit values."""
    import mpmath as mp
    for v in golden["G6_hypergeom_mp"]:
        a, b, c, d = v["cells"]
        # build (each, total) that produce exactly these cells when no clamp is active, else call via cells
        # a two-subgenome row [a, b] with totals [T0, T1] gives x12 = b, x21 = T0 - a,
        # x22 = T1 + a - b; clamped cells (== MAX_INT) are reached with any larger total
        each = [a, b]
        T0 = a + (300000000 if c == stats.MAX_INT else c)
        T1 = 300000000 if d == stats.MAX_INT else d + b - a
        if T1 < b:
            continue
        total = [T0, T1]
        assert po.fisher_cells(each, total, 0) == (a, b, c, d)
        p = stats.fisher_test(each, total, ctx=ctx)[0]
        exact = float(mp.mpf(v["p"]))
        assert math.isclose(p, exact, rel_tol=1e-8, abs_tol=1e-300), (v, p)
    for name, t in golden["G6_enrich"]["tables"].items():
        counts = np.array(t["counts"], np.int64)
        total = counts.sum(axis=0)
        for r, exp in zip(counts[:6], t["cells_first6"]):
            got = [list(po.fisher_cells(r, total, j)) for j in range(len(r))]
            assert got == exp, name


# ------------------------------------------------------------------ array fast path (bench / pipeline)
def check_hotpath_stack(ctx, golden, toy):
    """HotPath.map_and_enrich (no text round trip) == map_kmer3 text -> stack_matrix of the reference."""
    from subphaser_amd.hotpath import HotPath
    cl, labels = check_output_kmers(ctx, golden, toy)
    lens = [len(toy["seqs"][lab]) for lab in toy["labels"]]
    for ws in (1000, 1500, 2500, 100000):
        hp = HotPath(ctx, toy["labels"], lens, toy["sgs"], k=K, lower_count=L, bin_size=100, chunk_size=2000,
                     window_size=ws)
        r = hp.map_and_enrich(labels, len(cl.sg_names))
        exp = golden["G5_stack_matrix"][str(ws)]
        assert [[c, s, e] for c, s, e in r.coords] == exp["coords"], ws
        assert r.window_counts.tolist() == exp["counts"], ws
        assert r.pvals.shape == r.window_counts.shape


# ------------------------------------------------------------------ end-to-end CLI (modules 1-2)
def check_pipeline_cli(ctx, golden, toy, tmp_path):
    """`subphaser -i genome.fa -c sg.config -sg_assigned ...` on the toy genome: the files the
    reference would write for modules 1-2, compared with the goldens."""
    import gzip
    from subphaser_amd import pipeline, runtime
    fa = tmp_path / "toy.fa.gz"
    with gzip.open(fa, "wt") as f:
        for lab in toy["labels"]:
            s = toy["seqs"][lab]
            f.write(">%s some description\n" % lab)
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")
        f.write(">unplaced_scaffold\nACGTACGTNNNN\n")
    cfg = tmp_path / "sg.config"
    cfg.write_text("# toy\n" + "\n".join("\t".join(",".join(u) for u in sg) for sg in toy["sgs"]) + "\n")
    asg = tmp_path / "assigned.tsv"
    asg.write_text("".join("%s\t%s\n" % kv for kv in toy["sg_assigned"].items()))
    feats = tmp_path / "features.fa"
    with open(feats, "w") as f:
        for fid, sq in golden["G4_map_features"]["features"]:
            f.write(">%s\n%s\n" % (fid, sq))
    out, tmpd = tmp_path / "out", tmp_path / "tmp"
    old = runtime._ctx
    runtime.set_context(ctx)
    try:
        pipeline.main(["-i", str(fa), "-c", str(cfg), "-sg_assigned", str(asg), "-q", "30", "-k", str(K),
                       "-o", str(out), "-tmpdir", str(tmpd), "-window_size", "2500", "-custom_features", str(feats),
                       "-disable_ltr", "-disable_circos", "-figfmt", "png"])
    finally:
        runtime._ctx = old
    base = out / ("k%d_q30_f2" % K)
    assert open(str(base) + ".kmer.mat").read() == golden["G3_kmer_mat_text"]
    assert open(str(base) + ".subgenome.bin.count").read() == golden["G4_map_kmer3"]["chunk10M_bin10k"]["text"]
    assert open(str(base) + ".custom.bin.count").read() == golden["G4_map_features"]["text"]
    sig = open(str(base) + ".sig.kmer-subgenome.tsv").read().strip().split("\n")
    assert len(sig) - 1 == golden["G7_output_kmers"]["n_dkmers"] // 2
    assert open(str(base) + ".chrom-subgenome.tsv").read().startswith("#chrom\tsubgenome\tbootstrap\n")
    for ext in (".bin.enrich", ".bin.group", ".custom.enrich"):
        assert len(open(str(base) + ext).read().strip().split("\n")) > 1, ext
    assert (tmpd / "split.ok").exists() and (tmpd / ("k%d_q30_f2.kmer.mat.ok" % K)).exists()
    assert (tmpd / "chromosomes" / "A1.fasta").exists()


def check_pipeline_cli_bed(ctx, golden, toy, tmp_path):
    """`-custom_features x.bed`: the features as intervals over the resident genome must give the files the FASTA of
    the same sub-sequences gives (`.custom.bin.count`, `.custom.enrich` byte for byte), and the `.custom.bin.count`
    lines must be the reference's (G4) for those features."""
    import re
    from subphaser_amd import pipeline, runtime
    fa = tmp_path / "toy.fa"
    with open(fa, "w") as f:
        for lab in toy["labels"]:
            f.write(">%s\n%s\n" % (lab, toy["seqs"][lab]))
    cfg = tmp_path / "sg.config"
    cfg.write_text("\n".join("\t".join(",".join(u) for u in sg) for sg in toy["sgs"]) + "\n")
    asg = tmp_path / "assigned.tsv"
    asg.write_text("".join("%s\t%s\n" % kv for kv in toy["sg_assigned"].items()))
    feats = [(fid, sq) for fid, sq in golden["G4_map_features"]["features"]
             if re.match(r"(\S+?):(\d+)-(\d+)$", fid) and fid.split(":")[0] in toy["seqs"]]
    assert len(feats) >= 10
    ffa, fbed = tmp_path / "features.fa", tmp_path / "features.bed"
    with open(ffa, "w") as f, open(fbed, "w") as b:
        b.write("# toy features\ntrack name=toy\n")
        for fid, sq in feats:
            c, a, e = re.match(r"(\S+?):(\d+)-(\d+)$", fid).groups()
            assert toy["seqs"][c][int(a):int(e)] == sq
            f.write(">%s\n%s\n" % (fid, sq))
            b.write("%s\t%s\t%s\tfeature_%s\t0\t+\n" % (c, a, e, a))
        b.write("unplaced_scaffold\t0\t10\n")        # not a target chromosome: skipped
    res = {}
    old = runtime._ctx
    runtime.set_context(ctx)
    try:
        for tag, ff in (("fasta", ffa), ("bed", fbed)):
            out, tmpd = tmp_path / ("out_" + tag), tmp_path / ("tmp_" + tag)
            pipeline.main(["-i", str(fa), "-c", str(cfg), "-sg_assigned", str(asg), "-q", "30", "-k", str(K), "-o", str(out),
                           "-tmpdir", str(tmpd), "-window_size", "2500", "-custom_features", str(ff), "-disable_ltr",
                           "-disable_circos", "-figfmt", "png"])
            base = out / ("k%d_q30_f2" % K)
            res[tag] = (open(str(base) + ".custom.bin.count").read(), open(str(base) + ".custom.enrich").read())
    finally:
        runtime._ctx = old
    assert res["bed"][0] == res["fasta"][0]
    assert res["bed"][1] == res["fasta"][1] and len(res["bed"][1].strip().split("\n")) > 3
    ref_lines = [l for l in golden["G4_map_features"]["text"].split("\n") if l.split("\t")[0] in dict(feats)]
    assert res["bed"][0].strip().split("\n")[1:] == ref_lines and len(ref_lines) >= 5


def check_long_feature(ctx, golden, toy, tmp_path):
    """A feature longer than the bin size is reported per bin, like a chunk-less chromosome."""
    cl, labels = check_output_kmers(ctx, golden, toy)
    seq = toy["seqs"]["B1"][:12345]
    ff = tmp_path / "long.fa"
    ff.write_text(">B1:0-12345\n%s\n>short:1-40\n%s\n" % (seq, seq[5000:5040]))
    buf, rows = io.StringIO(), []
    seqs.map_kmer3([str(ff)], labels, fout=buf, k=K, bin_size=1000, sg_names=cl.sg_names, chunk=False, log=False,
                   ctx=ctx, collect=rows)
    _check_collected(rows, buf.getvalue(), tmp_path, 5000)
    exp, _, _ = po.map_bins(seq, K, labels.keys, labels.sg_idx, len(cl.sg_names), 1000, 0)
    lines = [l.split("\t") for l in buf.getvalue().strip().split("\n")[1:] if l.startswith("B1:")]
    got = {int(l[1]) // 1000: [int(x) for x in l[3:]] for l in lines}
    for b in range(13):
        assert got.get(b, [0] * len(cl.sg_names)) == exp[b].tolist(), b
    assert all(int(l[2]) == min(int(l[1]) + 1000, 12345) for l in lines)


# ------------------------------------------------------------------ G10: BASELINE chromosome/set structures
def _sha(txt):
    import hashlib
    return hashlib.sha256(txt.encode()).hexdigest()


def check_shape(ctx, golden, shape, engine=0, k=None, ent=None):
    """Whole path on a toy with the structure of the wheat (21 / 7 x 3), peanut (20 / 10 x 2) or
    Arabidopsis suecica (13, comma-grouped units) config, against what the imported reference produced.
    k / ent: another k-mer length with its own fixture entry (G14: k = 17, 21)."""
    from toygenome import make_shape_genome
    K = k or globals()["K"]
    tg = make_shape_genome(shape)
    ent = ent or golden["G10_shapes"][shape]
    files = []
    for lab in tg["labels"]:
        path = "/virtual/shape_%s/%s.fasta" % (shape, lab)
        if path not in seqs._REG:
            seqs._REG[path] = seqs.ChromRecord(lab, tg["seqs"][lab].encode())
        files.append(path)
    dumps = jellyfish.run_jellyfish_dumps(files, k=K, lower_count=L, ctx=ctx, engine=engine)
    for cname, case in ent["cases"].items():
        jd = jellyfish.JellyfishDumps(dumps, tg["labels"])
        d_mat = jd.to_matrix()
        assert jd.lengths == case["lengths"], (shape, cname)
        d2 = jd.filter(d_mat, jd.lengths, tg["sgs"], outfig="h.png", **case["kw"])
        assert len(d_mat) == case["n_union"], (shape, cname)
        assert len(d2) == case["n_rows"], (shape, cname)
        rows = ["\t".join([km] + [repr(float(x)) for x in fr]) for km, fr in d2.items()]
        assert rows[:5] == case["rows_head"], (shape, cname)
        assert _sha("\n".join(rows)) == case["rows_sha256"], (shape, cname)
        hist = np.sort(jd.hist_tot()).astype(np.int64)
        assert len(hist) == case["hist_n"] and int(hist.sum()) == case["hist_sum"], (shape, cname)
        assert _sha(",".join(map(str, hist.tolist()))) == case["hist_sha256"], (shape, cname)
        if "kmer_mat_sha256" not in case:
            continue
        buf = io.StringIO()
        jd.write_matrix(d2, buf)
        assert _sha(buf.getvalue()) == case["kmer_mat_sha256"], shape
        cl = cluster.Cluster(d2, n_clusters=tg["n_sg"], sg_prefix="SG", sg_assigned=tg["sg_assigned"])
        assert cl.sg_names == case["sg_names"] and dict(cl.d_sg) == case["d_sg"], shape
        buf = io.StringIO()
        labels = cl.output_kmers(buf, max_pval=0.05)
        assert len(labels) == case["n_dkmers"], shape
        sig = sorted(buf.getvalue().strip().split("\n")[1:])
        assert _sha("\n".join("\t".join(l.split("\t")[:2]) for l in sig)) == case["sig_kmers_sha256"], shape
        buf = io.StringIO()
        seqs.map_kmer3(files, labels, fout=buf, k=K, sg_names=cl.sg_names, ctx=ctx, window_size=4000, bin_size=500,
                       chunk=True)
        assert buf.getvalue() == case["bin_count_text"], shape
        # array fast path (what bench.py times) against the reference's stack_matrix + enrich_bin
        from subphaser_amd.hotpath import HotPath
        lens = [len(tg["seqs"][lab]) for lab in tg["labels"]]
        hp = HotPath(ctx, tg["labels"], lens, tg["sgs"], k=K, lower_count=L, bin_size=500, chunk_size=4000,
                     window_size=case["window_size"])
        r = hp.map_and_enrich(labels, len(cl.sg_names))
        assert [[c, s, e] for c, s, e in r.coords] == case["coords"], shape
        assert r.window_counts.tolist() == case["counts"], shape
        f1, f2 = io.StringIO(), io.StringIO()
        stats.enrich_bin(f1, f2, dict(cl.d_sg), r.window_counts.tolist(), colnames=cl.sg_names, rownames=r.coords,
                         max_pval=0.05, ctx=ctx)
        _cmp_enrich_text(f1.getvalue(), case["enrich_text"], {4, 10}, {8})
        assert f2.getvalue() == case["group_text"], shape


# ------------------------------------------------------------------ G11 / G12 / G13 host rows pinned by the reference
def check_split_genomes(golden, tmp_path):
    """Seqs.split_genomes (Seqs.py:27-71): `new|old` renames, label prefixes, d_targets, missing ids."""
    from collections import OrderedDict
    g = golden["G11_split_genomes"]
    paths = []
    for name, txt in sorted(g["genomes"].items()):
        p = tmp_path / name
        p.write_text(txt)
        paths.append(str(p))
    for cname, c in g["cases"].items():
        od = str(tmp_path / cname) + "/"
        (tmp_path / cname).mkdir()
        dt = OrderedDict(c["d_targets"]) if c["d_targets"] else None
        outfas, labs, dt2, dsz = seqs.split_genomes(paths, c["prefixes"], c["targets"], od, d_targets=dt, sep="|")
        assert [x[len(od):] for x in outfas] == c["files"], cname
        assert labs == c["labels"], cname
        assert sorted([list(kv) for kv in dt2.items()]) == c["d_targets2"], cname
        assert dsz == c["d_size"], cname
        for f in outfas:
            assert "".join(open(f).read().split("\n")[1:]) == c["seqs"][f[len(od):]], (cname, f)
            assert open(f).read().split("\n")[0] == ">" + f[len(od):-len(".fasta")], (cname, f)


def check_stat_enrich(golden, tmp_path):
    from subphaser_amd.stat_enrich import summarize
    g = golden["G12_stat_enrich"]
    p = tmp_path / "e4.tsv"
    p.write_text(g["input"])
    buf = io.StringIO()
    summarize(str(p), buf)
    assert buf.getvalue() == g["output"]


def check_dump_roundtrip(ctx, golden, toy, tmp_path):
    """KmerDump.write_text writes the text the reference's own dump parser was fed when the fixture was
    generated (it parsed `ref_lengths` / `ref_n_union` out of exactly these bytes, Jellyfish.py:46-98,439-460)."""
    g = golden["G13_dump_roundtrip"]
    files = []
    for lab in g["labels"]:
        path = str(tmp_path / ("rt_%s.fasta" % lab))
        seqs._REG[path] = seqs.ChromRecord(lab, toy["seqs"][lab].encode())
        files.append(path)
    dumps = jellyfish.run_jellyfish_dumps(files, k=K, lower_count=L, ctx=ctx, write_dumps=True)
    for d, exp in zip(dumps, g["text_sha256"]):
        assert _sha(open(str(d)).read()) == exp
        assert (tmp_path / (str(d).split("/")[-1] + ".ok")).exists()
    jd = jellyfish.JellyfishDumps(dumps, g["labels"])
    assert len(jd.to_matrix()) == g["ref_n_union"]
    assert jd.lengths == g["ref_lengths"]
