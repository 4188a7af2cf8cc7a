#!/usr/bin/env python3
"""G14: the reference's whole hot path on the wheat-structured toy (21 chromosomes / 7 sets x 3) at k = 17 and 21
(BASELINE config 5's k values) -- the same steps and fields as G10 of gen_golden.py, which has k = 15 only.

Run in the authoring container only (needs /root/reference):    python tests/golden/gen_golden_k17.py
Writes tests/golden/golden_k17.json.gz (data: inputs are a seeded toy genome, outputs are what the imported,
unmodified reference produced; the dump files it parses come from oracle/pyoracle.count_bruteforce, like G10's)."""
import gzip
import hashlib
import io
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (stand-ins for the four absent third-party packages, write_dump)
import pyoracle as po  # noqa: E402
from toygenome import make_shape_genome  # noqa: E402


def sha(txt):
    return hashlib.sha256(txt.encode()).hexdigest()


def main():
    tmp = tempfile.mkdtemp(prefix="sp_golden17_")
    gg.install_standins(tmp)
    import logging
    logging.disable(logging.CRITICAL)
    from subphaser import Jellyfish, Seqs, Circos, Stats  # noqa
    from subphaser.Cluster import Cluster  # noqa
    captured = []
    Jellyfish.plot_histogram = lambda data, outfig, **kw: captured.append(sorted(int(x) for x in data))
    L = 3
    out = {}
    for shape, k in (("wheat", 17), ("wheat", 21), ("peanut", 17)):
        tg = make_shape_genome(shape)
        sdir = os.path.join(tmp, "shape_%s_k%d" % (shape, k))
        os.makedirs(sdir)
        chromfiles, dumpfiles = [], []
        for lab in tg["labels"]:
            cf = os.path.join(sdir, lab + ".fasta")
            with open(cf, "w") as f:
                f.write(">%s\n%s\n" % (lab, tg["seqs"][lab]))
            chromfiles.append(cf)
            keys, cnts = po.count_bruteforce(tg["seqs"][lab], k, L)
            df = "%s_%d.fa" % (cf, k)
            gg.write_dump(df, keys, cnts, k)
            dumpfiles.append(df)
        ent = {"seed": 11, "k": k, "shape": shape, "cases": {}}
        for cname, kw in {"q20_f2": dict(min_freq=20, max_freq=1e9, min_fold=2, baseline=1, ratio=1),
                          "q10_f3_last_r06": dict(min_freq=10, max_freq=500, min_fold=3, baseline=-1, ratio=0.6)}.items():
            jd = Jellyfish.JellyfishDumps(dumpfiles, tg["labels"], ncpu=2, method="map", chunksize=None)
            d_mat = jd.to_matrix()
            del captured[:]
            d2 = jd.filter(d_mat, jd.lengths, tg["sgs"], outfig=os.path.join(tmp, "h.png"), **kw)
            rows = sorted(d2.items())
            rows_txt = "\n".join("\t".join([km] + [repr(float(x)) for x in fr]) for km, fr in rows)
            case = {"kw": kw, "n_union": len(d_mat), "lengths": [int(x) for x in jd.lengths], "n_rows": len(rows),
                    "rows_sha256": sha(rows_txt), "rows_head": rows_txt.split("\n")[:5],
                    "hist_n": len(captured[0]), "hist_sum": int(sum(captured[0])),
                    "hist_sha256": sha(",".join(map(str, captured[0])))}
            ent["cases"][cname] = case
            if cname != "q20_f2":
                continue
            buf = io.StringIO()
            jd.write_matrix(dict(rows), buf)
            case["kmer_mat_sha256"] = sha(buf.getvalue())
            matfile = os.path.join(sdir, "x.kmer.mat")
            open(matfile, "w").write(buf.getvalue())
            cl = Cluster(matfile, n_clusters=tg["n_sg"], sg_prefix="SG", sg_assigned=tg["sg_assigned"], bootstrap=False)
            buf = io.StringIO()
            dk = cl.output_kmers(buf, max_pval=0.05, ncpu=2, test_method="ttest_ind")
            sig_lines = sorted(buf.getvalue().strip().split("\n")[1:])
            case["sg_names"], case["d_sg"] = cl.sg_names, dict(cl.d_sg)
            case["n_dkmers"] = len(dk)
            case["sig_kmers_sha256"] = sha("\n".join("\t".join(l.split("\t")[:2]) for l in sig_lines))
            buf = io.StringIO()
            Seqs.map_kmer3(chromfiles, dk, fout=buf, k=k, sg_names=cl.sg_names, ncpu=2, method="map",
                           window_size=4000, bin_size=500, chunk=True)
            case["bin_count_text"] = buf.getvalue()
            binf = os.path.join(sdir, "x.bin.count")
            open(binf, "w").write(buf.getvalue())
            coords, counts = Circos.stack_matrix(binf, window_size=2000)
            case["window_size"] = 2000
            case["coords"] = [[c, int(s), int(e)] for c, s, e in coords]
            case["counts"] = [[int(x) for x in row] for row in counts]
            f1, f2 = io.StringIO(), io.StringIO()
            Stats.enrich_bin(f1, f2, dict(cl.d_sg), counts, colnames=cl.sg_names, rownames=coords, max_pval=0.05, ncpu=2)
            case["enrich_text"], case["group_text"] = f1.getvalue(), f2.getvalue()
        out["%s_k%d" % (shape, k)] = ent
        print(shape, k, {c: (v["n_union"], v["n_rows"]) for c, v in ent["cases"].items()})
    with gzip.open(os.path.join(HERE, "golden_k17.json.gz"), "wt") as f:
        json.dump({"G14_shapes_k17_k21": out}, f, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden_k17.json.gz"))


if __name__ == "__main__":
    main()
