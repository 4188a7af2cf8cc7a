"""CPU suite: oracle pinned against the golden vectors, through the host layer."""
import numpy as np
import pytest

import parity_cases as pc
import pyoracle as po


def test_oracle_count_matches_bruteforce():
    rng = np.random.RandomState(0)
    alphabet = list("ACGTacgtNR")
    for k in (1, 2, 5, 15, 16, 17, 21, 31, 32):
        s = "".join(rng.choice(alphabet, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02], size=2500))
        for lower, nt in ((1, 1), (2, 3)):
            a, b = po.count(s, k, lower, nthreads=nt), po.count_bruteforce(s, k, lower)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), k
    # dense-table path of the oracle (k small relative to the sequence)
    s = "".join(rng.choice(list("ACGTn"), size=20000))
    for k in (3, 6, 7):
        a, b = po.count(s, k, 1, nthreads=4), po.count_bruteforce(s, k, 1)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), k


def test_oracle_count_edges():
    assert len(po.count("", 15)[0]) == 0
    assert len(po.count("ACGT", 15)[0]) == 0              # shorter than k
    assert len(po.count("N" * 100, 5)[0]) == 0
    keys, cnts = po.count("A" * 50, 15)                    # hot key, AAAA == rc(TTTT)
    assert keys.tolist() == [0] and cnts.tolist() == [36]
    keys, cnts = po.count("T" * 50, 15)
    assert keys.tolist() == [0] and cnts.tolist() == [36]
    keys, cnts = po.count("ACGT" * 10, 4)                  # even k: palindromes ACGT, CGTA ...
    bf = po.count_bruteforce("ACGT" * 10, 4)
    assert keys.tolist() == bf[0].tolist() and cnts.tolist() == bf[1].tolist()
    # window broken exactly at k-1 / k valid bases
    assert len(po.count("ACGTACGTACGTACnACGTACGTACGTAC", 15)[0]) == 0
    assert len(po.count("ACGTACGTACGTACGnACGTACGTACGTAC", 15)[0]) == 1


def test_toy_dumps(oracle_ctx, golden, toy):
    pc.check_toy_dumps(oracle_ctx, golden, toy)


def test_filter_cases(oracle_ctx, golden, toy):
    pc.check_filter_cases(oracle_ctx, golden, toy)


def test_kmer_mat_text(oracle_ctx, golden, toy):
    pc.check_kmer_mat_text(oracle_ctx, golden, toy)


def test_output_kmers(oracle_ctx, golden, toy):
    pc.check_output_kmers(oracle_ctx, golden, toy)


def test_map_cases(oracle_ctx, golden, toy):
    pc.check_map_cases(oracle_ctx, golden, toy)


def test_map_features(oracle_ctx, golden, toy, tmp_path):
    pc.check_map_features(oracle_ctx, golden, toy, tmp_path)


def test_long_feature(oracle_ctx, golden, toy, tmp_path):
    pc.check_long_feature(oracle_ctx, golden, toy, tmp_path)


def test_kmeans_assignment(oracle_ctx, golden, toy):
    """No -sg_assigned: scikit-learn KMeans on the Z-normalised matrix, as the reference does,
    separates the two planted subgenomes of the toy genome."""
    from subphaser_amd import cluster
    d2 = pc.check_kmer_mat_text(oracle_ctx, golden, toy)
    cl = cluster.Cluster(d2, n_clusters=2)
    groups = {}
    for c, sg in cl.d_sg.items():
        groups.setdefault(sg, set()).add(c[0])
    assert sorted(cl.sg_names) == ["SG1", "SG2"]
    assert sorted(tuple(sorted(v)) for v in groups.values()) == [("A",), ("B",)]


def test_map_dict_labels(oracle_ctx, golden, toy):
    pc.check_dict_labels(oracle_ctx, golden, toy)


def test_hotpath_stack(oracle_ctx, golden, toy):
    pc.check_hotpath_stack(oracle_ctx, golden, toy)


def test_pipeline_cli(oracle_ctx, golden, toy, tmp_path):
    pc.check_pipeline_cli(oracle_ctx, golden, toy, tmp_path)


@pytest.mark.parametrize("shape", ["wheat", "peanut", "ara"])
def test_baseline_shapes(oracle_ctx, golden, shape):
    pc.check_shape(oracle_ctx, golden, shape)


def test_split_genomes_reference_cases(golden, tmp_path):
    pc.check_split_genomes(golden, tmp_path)


def test_stat_enrich_reference_output(golden, tmp_path):
    pc.check_stat_enrich(golden, tmp_path)


def test_dump_roundtrip(oracle_ctx, golden, toy, tmp_path):
    pc.check_dump_roundtrip(oracle_ctx, golden, toy, tmp_path)


def test_stack_matrix(golden, tmp_path):
    pc.check_stack_matrix(golden, tmp_path)


def test_enrich_bin(oracle_ctx, golden, tmp_path):
    pc.check_enrich_bin(oracle_ctx, golden, tmp_path)


def test_enrich_features(oracle_ctx, golden, tmp_path):
    pc.check_enrich_features(oracle_ctx, golden, tmp_path)


def test_fisher_cells_and_tails(oracle_ctx, golden):
    pc.check_fisher_cells_and_tails(oracle_ctx, golden)


def test_oracle_tail_vs_scipy():
    from scipy.stats import hypergeom
    rng = np.random.RandomState(3)
    for _ in range(300):
        a, b = rng.randint(0, 60), rng.randint(0, 400)
        c, d = rng.randint(0, 5000), rng.randint(0, 50000)
        p = po.hypergeom_right_tail(a, b, c, d)
        q = float(hypergeom.sf(a - 1, a + b + c + d, a + b, a + c))
        assert abs(p - q) <= 1e-9 * max(q, 1e-300) + 1e-15, (a, b, c, d, p, q)


def test_bh_hand_vector():
    # hand-computed Benjamini-Hochberg: p*n/rank, then reverse cumulative minimum
    p = [0.01, 0.04, 0.03, 0.005, 0.5]
    exp = [0.025, 0.05, 0.05, 0.025, 0.5]
    from subphaser_amd.stats import correct_pvals
    assert np.allclose(po.bh_correct(p), exp, rtol=0, atol=1e-15)
    assert np.allclose(correct_pvals(p), exp, rtol=0, atol=1e-15)


def test_circos_tracks_from_bin_counts(golden, tmp_path):
    """G9: per-subgenome circos tracks derived from the bin counts (Circos.stack_bed_density, f-3 row):
    byte-identical files for trimmed and untrimmed windows; sg_ratio / sg_enrich writer."""
    import contextlib
    import io
    from subphaser_amd import circos
    g = golden["G9_stack_bed_density"]
    binfile = tmp_path / "toy.bin.count"
    binfile.write_text(g["bin_count_text"])
    for name, exp in g["tracks"].items():
        ws, mode = name.split("_")
        with contextlib.redirect_stderr(io.StringIO()):
            files = circos.stack_bed_density(str(binfile), str(tmp_path / name), g["sg_names"], window_size=int(ws),
                                             trim=(mode == "trim"))
        assert sorted(files) == sorted(exp)
        for key, path in files.items():
            assert open(path).read() == exp[key], (name, key)
    lines = [["chrA", 0, 2500, "SG1", 0.01, "5,6", "0.4,0.6", "1,0,0", "0.1,0.2", "no"],
             ["chrB", 2500, 5000, None, 1.0, "0,0", "nan,nan", "0,0,1", "1.0,1.0", "none"]]
    (tmp_path / "d").mkdir()
    rf, ef = circos.out_sg_lines(lines, str(tmp_path / "d"))
    assert open(rf).read() == "chrA\t0\t2500\t0.4,0.6\nchrB\t2500\t5000\tnan,nan\n"
    assert open(ef).read() == "chrA\t0\t2500\t1,0,0\nchrB\t2500\t5000\t0,0,1\n"


def test_pipeline_cli_bed_features(oracle_ctx, golden, toy, tmp_path):
    pc.check_pipeline_cli_bed(oracle_ctx, golden, toy, tmp_path)


@pytest.mark.parametrize("name", ["wheat_k17", "wheat_k21", "peanut_k17"])
def test_baseline_shapes_k17_k21(oracle_ctx, golden, golden_k17, name):
    """G14 pins the oracle (and the host layer) at k = 17 / 21 to the imported reference"""
    ent = golden_k17[name]
    pc.check_shape(oracle_ctx, golden, ent["shape"], k=ent["k"], ent=ent)
