"""CPU suite: the C-ABI library loads and exports every declared symbol (no compute
without a GPU), and the pure-host pieces (config grammar, codec, writers) match the goldens."""
import io
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "subphaser_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from subphaser_amd import _native
    lib = _native.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libsubphaser_hip.so lacks %s" % name
    assert sorted(_native.SYMBOLS) == declared     # the binding covers the whole header
    assert lib.sp_version() >= 100


def test_no_gpu_means_loud_failure():
    """There is no CPU fallback: without a device the context cannot be created."""
    import ctypes
    from subphaser_amd import _native
    lib = _native.load()
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        have_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        have_gpu = False
    if have_gpu:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_native.NativeError, match="no HIP device"):
        _native.Context(0)


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "subphaser_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "sp_oracle" not in txt and "oracle_ctx" not in txt, f


def test_sgconfig_matches_reference(golden, tmp_path):
    from subphaser_amd.config import SGConfig
    for name, ent in golden["G8_sgconfig"].items():
        p = tmp_path / name
        p.write_text(ent["text"])
        for parse in ent["parses"]:
            cfg = SGConfig(str(p), prefix=parse["prefix"], sep="|")
            assert cfg.sgs == parse["sgs"], (name, parse["prefix"])
            assert cfg.chrs == parse["chrs"], name
            assert cfg.nsg == parse["nsg"], name


def test_idmap_and_duplicates(tmp_path):
    from subphaser_amd.config import check_duplicates, parse_idmap
    p = tmp_path / "t.map"
    p.write_text("# c\nold1 new1\n3|old2  # trailing\n\nold3\n")
    d = parse_idmap(str(p))
    assert list(d.items()) == [("old1", "new1"), ("3|old2", "old2"), ("old3", "old3")]
    assert parse_idmap(None) is None
    with pytest.raises(ValueError, match="Duplicates"):
        check_duplicates(["a", "b", "a"])


def test_kmer_codec_roundtrip():
    from subphaser_amd import kmer
    rng = np.random.RandomState(0)
    for k in (1, 5, 15, 21, 31, 32):
        keys = rng.randint(0, 2 ** 62, size=50, dtype=np.int64).astype(np.uint64) & np.uint64((1 << (2 * k)) - 1 if k < 32 else 2 ** 64 - 1)
        s = kmer.decode_many(keys, k)
        assert (kmer.encode_many(s) == keys).all()
        rc = kmer.revcomp(keys, k)
        assert (kmer.revcomp(rc, k) == keys).all()
        comp = str.maketrans("ACGT", "TGCA")
        assert kmer.decode_many(rc, k) == [x.translate(comp)[::-1] for x in s]
        assert (kmer.canonical(keys, k) == np.minimum(keys, rc)).all()
    assert kmer.encode("ACGT") == 0b00011011 and kmer.decode(0b00011011, 4) == "ACGT"
    with pytest.raises(ValueError):
        kmer.encode("ACNT")


def test_fasta_reader_and_split(tmp_path):
    from subphaser_amd import seqs
    fa = tmp_path / "g.fa"
    fa.write_text(">chr1 desc here\nACGT\nacgtNN\n\n>chr2\nTTTT\n>other\nGG\n")
    recs = list(seqs.read_fasta(str(fa)))
    assert recs == [("chr1", b"ACGTacgtNN"), ("chr2", b"TTTT"), ("other", b"GG")]
    import gzip
    gz = tmp_path / "g.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(fa.read_bytes())
    assert list(seqs.read_fasta(str(gz))) == recs
    out = str(tmp_path) + "/chrom_"
    files, labels, d_targets, d_size = seqs.split_genomes([str(fa)], [""], ["A|chr1", "chr2"], out)
    assert labels == ["A", "chr2"] and d_size == {"A": 10, "chr2": 4}
    assert list(d_targets.items()) == [("A|chr1", "A"), ("chr2", "chr2")]
    assert open(files[0]).read() == ">A\nACGTacgtNN\n"
    assert bytes(seqs.load_chromfile(files[1]).seq) == b"TTTT"   # seq is bytes-like (bytes or a uint8 view of the mmap)


def test_stat_enrich_summary(tmp_path):
    from subphaser_amd.stat_enrich import summarize
    p = tmp_path / "e.tsv"
    p.write_text("#id\tsubgenome\tp_value\tcounts\tpotential_exchange\tp_corrected\n"
                 "LTR-1\tSG1\t0.01\t5,1\tno\t0.02\nLTR-2\tSG1\t0.01\t7,0\tno\t0.02\n"
                 "LTR-3\tSG2\t0.01\t0,9\tno\t0.02\nGENE-1\tNone\t0.5\t1,1\tnone\t0.5\n")
    buf = io.StringIO()
    summarize(str(p), buf)
    assert buf.getvalue() == "GENE\t1\t0\t0\t1\t1\nLTR\t0\t2\t1\t12\t10\n"


def test_cli_parses_reference_flags():
    from subphaser_amd.pipeline import makeArgparse
    a = makeArgparse("-i g.fa -c sg.cfg -pre x/y -k 15 -q 200 -f 2 -window_size 1000000 -disable_ltr "
                     "-disable_circos -sg_assigned t.tsv -custom_features te.fa -p 8".split())
    assert a.prefix == "x_y" and a.outdir == "x_yphase-results" and a.tmpdir == "x_ytmp"
    assert a.k == 15 and a.min_freq == 200 and a.min_fold == 2.0 and a.lower_count == 3
    assert a.max_freq == 1e9 and a.baseline == 1 and a.ratio == 1 and a.max_pval == 0.05
    assert a.custom_features == ["te.fa"] and a.ncpu == 8


def test_fasta_bulk_reader_matches_record_reader(tmp_path):
    """read_fasta_bulk (feature sets with millions of records) == read_fasta, across worker-span cuts."""
    import numpy as np
    from subphaser_amd import seqs
    rng = np.random.RandomState(1)
    recs, txt = [], ["leading junk\n"]
    for i in range(400):
        L = int(rng.randint(0, 900))
        w = int(rng.choice([60, 80, 10 ** 6]))
        sq = "".join(rng.choice(list("ACGTNacgt"), L))
        recs.append(("r%d" % i, sq.encode()))
        txt.append(">r%d some desc\n" % i + "\n".join(sq[j:j + w] for j in range(0, L, w)) + ("\n" if L else ""))
    txt.append(">last\r\nAC GT\r\n>empty")
    recs += [("last", b"ACGT"), ("empty", b"")]
    fa = tmp_path / "f.fa"
    fa.write_text("".join(txt))
    assert list(seqs.read_fasta(str(fa))) == recs
    old = seqs._BULK_STEP
    try:
        for step in (old, 4096, 17, 1001):
            seqs._BULK_STEP = step
            ids, cat, off = seqs.read_fasta_bulk(str(fa))
            assert [(i, bytes(cat[off[j]:off[j + 1]])) for j, i in enumerate(ids)] == recs, step
    finally:
        seqs._BULK_STEP = old
    empty = tmp_path / "e.fa"
    empty.write_text("")
    assert seqs.read_fasta_bulk(str(empty))[0] == []


def test_fasta_scanner_matches_numpy_twin(tmp_path, monkeypatch):
    """sp_fasta_* (host code of the library) against the numpy readers it replaces: CRLF, blank lines, blanks inside
    lines, empty records, text before the first record, no final line break, records larger than a work piece."""
    import numpy as np
    from subphaser_amd import _native, seqs
    rng = np.random.RandomState(5)
    alpha = np.frombuffer(b"ACGTNacgtn", np.uint8)
    parts = [b"; a comment line before any record\n"]
    want = []
    for r in range(300):
        n = int(rng.choice([0, 1, 59, 60, 61, 500, 20000]))
        if r == 7:
            n = 9_500_000       # spans two 8-MiB work pieces
        seq = alpha[rng.randint(0, len(alpha), size=n)].tobytes()
        rid = "rec%d" % r
        want.append((rid, seq))
        eol = b"\r\n" if r % 3 == 0 else b"\n"
        parts.append(b">" + rid.encode() + (b" some description" if r % 2 else b"") + eol)
        w = int(rng.choice([60, 70, 80, 100000]))
        for i in range(0, n, w):
            line = seq[i:i + w]
            if r % 5 == 0 and len(line) > 4:
                line = line[:2] + b" " + line[2:4] + b"\t" + line[4:]
            parts.append(line + eol)
        if r % 11 == 0:
            parts.append(eol)
    blob = b"".join(parts)
    blob = blob[:-1] if blob.endswith(b"\n") else blob        # no final line break
    path = tmp_path / "messy.fa"
    path.write_bytes(blob)
    for threads in (1, 3, 16):
        ids, cat, off = _native.fasta_scan(np.frombuffer(blob, np.uint8), threads=threads)
        assert ids == [w_[0] for w_ in want]
        assert off.tolist() == np.concatenate(([0], np.cumsum([len(w_[1]) for w_ in want]))).tolist()
        assert bytes(cat) == b"".join(w_[1] for w_ in want)
    got_native = list(seqs.read_fasta(str(path)))
    bulk_native = seqs.read_fasta_bulk(str(path))
    monkeypatch.setenv("SP_FASTA_NUMPY", "1")
    got_numpy = list(seqs.read_fasta(str(path)))
    bulk_numpy = seqs.read_fasta_bulk(str(path))
    assert got_native == got_numpy == want
    assert bulk_native[0] == bulk_numpy[0] and bytes(bulk_native[1]) == bytes(bulk_numpy[1])
    assert bulk_native[2].tolist() == bulk_numpy[2].tolist()
    ids, cat, off = _native.fasta_scan(np.empty(0, np.uint8))
    assert ids == [] and cat.size == 0 and off.tolist() == [0]


def test_text_writers_match_python_formatting(tmp_path):
    """sp_text_* (host code of the library): repr(float) digit for digit, and the `.kmer.mat` / sig-k-mer rows
    byte for byte against the Python formatting the mirror uses for non-file objects."""
    import io
    import numpy as np
    from subphaser_amd import _native, cluster, jellyfish
    rng = np.random.default_rng(11)
    xs = [0.0, -0.0, 1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1e15, 1e16, 9007199254740992.0, 1e22, 5e-324,
          1.7976931348623157e308, float("nan"), float("inf"), -float("inf"), 2.2250738585072014e-308, 1 / 3, 100.0]
    xs += list(rng.random(20000)) + list(np.exp(rng.uniform(-700, 700, 20000)))
    xs += [c / l for c, l in zip(rng.integers(0, 5000, 20000), rng.integers(1, 10 ** 9, 20000))]
    for e in range(-1070, 1023, 13):
        xs += [2.0 ** e, np.nextafter(2.0 ** e, 0), np.nextafter(2.0 ** e, np.inf)]
    xs = np.array(xs, np.float64)
    assert _native.text_repr(xs) == [repr(float(v)) for v in xs]
    # .kmer.mat rows
    M, Cn, k = 30011, 7, 15

    class _Mat:
        pass
    mat = _Mat()
    mat.keys = rng.integers(0, 4 ** k, M, dtype=np.int64).astype(np.uint64)
    mat.freqs = rng.integers(0, 3000, (M, Cn)) / rng.integers(10 ** 6, 10 ** 9, Cn)
    mat.k = k
    dumps = jellyfish.JellyfishDumps.__new__(jellyfish.JellyfishDumps)
    dumps.labels = ["c%d" % i for i in range(Cn)]
    buf = io.StringIO()
    dumps.write_matrix(mat, buf)                     # Python formatting (no file descriptor)
    with open(tmp_path / "m.mat", "w") as f:
        dumps.write_matrix(mat, f)                   # library writer
    assert open(tmp_path / "m.mat").read() == buf.getvalue()
    # sig k-mer rows
    top = rng.integers(0, 3, M).astype(np.int32)
    pv = rng.random(M)
    pv[::97] = np.nan
    means = rng.random((M, 3)) * 1e-5
    buf = io.StringIO()
    with open(tmp_path / "s.tsv", "w") as f:
        assert _native.text_sig_kmers(f, mat.keys, k, top, ["SG1", "SG2", "SG10"], pv, means)
    from subphaser_amd import kmer as kmerlib
    exp = "".join("\t".join([km, ["SG1", "SG2", "SG10"][t], repr(p), ",".join(map(repr, mv))]) + "\n"
                  for km, t, p, mv in zip(kmerlib.decode_many(mat.keys, k), top.tolist(), pv.tolist(), means.tolist()))
    assert open(tmp_path / "s.tsv").read() == exp


def test_gz_and_bgzf_input(tmp_path):
    """gz FASTA: plain gzip (one stream) and BGZF (bgzip blocks, inflated in parallel) give the same records"""
    import gzip
    import struct
    import zlib
    import numpy as np
    from subphaser_amd import seqs
    rng = np.random.default_rng(9)
    recs = [("c%d" % i, np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, int(n))].tobytes().decode())
            for i, n in enumerate([0, 17, 200000, 70001])]
    raw = "".join(">%s\n%s\n" % (i, "\n".join(s[j:j + 60] for j in range(0, len(s), 60))) for i, s in recs).encode()
    blocks = []
    for i in range(0, len(raw), 65280):
        blk = raw[i:i + 65280]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(blk) + c.flush()
        blocks.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                      struct.pack("<II", zlib.crc32(blk), len(blk)))
    blocks.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))     # BGZF EOF block
    (tmp_path / "b.fa.gz").write_bytes(b"".join(blocks))
    with gzip.open(tmp_path / "p.fa.gz", "wb") as f:
        f.write(raw)
    assert gzip.open(tmp_path / "b.fa.gz").read() == raw          # a valid multi-member gzip file
    want = [(i, s.encode()) for i, s in recs]
    for name in ("b.fa.gz", "p.fa.gz"):
        assert bytes(seqs.read_gz(str(tmp_path / name))) == raw
        assert list(seqs.read_fasta(str(tmp_path / name))) == want
        ids, cat, off = seqs.read_fasta_bulk(str(tmp_path / name))
        assert ids == [i for i, _ in recs] and bytes(cat) == b"".join(s for _, s in want)


def test_text_table_and_enrich_ltr_file_vs_stream(tmp_path):
    """sp_text_table (generic TSV rows, host code of the library) against Python formatting, and stats.enrich_ltr
    written to a real file (library threads) against the same call on a StringIO (Python formatting): byte-identical.
    The enrichment itself is stubbed -- this is the writer's test and runs without a GPU."""
    import io
    import numpy as np
    from subphaser_amd import _native, stats
    rng = np.random.default_rng(5)
    n = 50021
    ids = ["chr%d:%d-%d" % (i % 7, i * 10, i * 10 + 9) for i in range(n)]
    ids[17] = "weird_id_without_coords"
    ids[18] = "a:b:3-4"
    ints = rng.integers(-5, 10 ** 12, (n, 3))
    fl = np.exp(rng.uniform(-300, 300, n))
    fl[::101] = np.nan
    idx = rng.integers(0, 3, n).astype(np.int32)
    blob, off = _native.str_blob(ids)
    with open(tmp_path / "t.tsv", "w") as f:
        f.write("#head\n")
        assert _native.text_table(f, n, [("str", blob, off), ("name", idx, ["x", "", "long name"]), ("f64", fl, ","),
                                         ("i64", ints, ","), ("i64", ints[:, :1], "\t")])
    exp = "#head\n" + "".join("%s\t%s\t%s\t%s\t%d\n" % (ids[i], ["x", "", "long name"][idx[i]], repr(float(fl[i])),
                                                      ",".join(map(str, ints[i].tolist())), ints[i, 0]) for i in range(n))
    assert open(tmp_path / "t.tsv").read() == exp
    assert _native.text_table(io.StringIO(), 0, []) is False      # no descriptor: the caller formats

    class _Ctx:
        def enrich(self, arr, max_pval, min_ratio):
            r = np.random.default_rng(9)
            p = r.random(arr.shape) ** 8
            return p, p.argmin(axis=1), p.min(axis=1) <= max_pval, None
    counts = rng.integers(0, 500, (n, 3))
    d_sg = {"chr%d" % i: "SG%d" % (1 + i % 3) for i in range(6)}          # chr6 unknown
    d_sg["a:b"] = "SGx"                                                  # a subgenome outside colnames
    rows = [(i,) for i in ids]
    buf = io.StringIO()
    e1, x1 = stats.enrich_ltr(buf, d_sg, counts, colnames=["SG1", "SG2", "SG3"], rownames=rows, ctx=_Ctx())
    with open(tmp_path / "e.tsv", "w") as f:
        e2, x2 = stats.enrich_ltr(f, d_sg, counts, colnames=["SG1", "SG2", "SG3"], rownames=rows, ctx=_Ctx())
    text = open(tmp_path / "e.tsv").read()
    assert text == buf.getvalue() and e1 == e2 and x1 == x2
    lines = text.splitlines()
    assert lines[0].split("\t") == ["#id", "subgenome", "p_value", "counts", "potential_exchange", "p_corrected"]
    assert x1["weird_id_without_coords"] == "none" and len(lines) == n + 1
    # the row-by-row rule of the reference (Stats.py:42-57, 133-138) on a sample
    import re
    p, am, sig, _ = _Ctx().enrich(counts, 0.05, 0.5)
    for i in list(range(0, 40)) + list(range(n - 40, n)):
        m = re.compile(r"(\S+?):\d+\-\d+").match(ids[i])
        obs = d_sg.get(m.groups()[0]) if m else None
        expc = ["SG1", "SG2", "SG3"][am[i]] if sig[i] else None
        want = "none" if (not expc or not obs) else ("no" if obs == expc else "yes")
        col = lines[1 + i].split("\t")
        assert col[0] == ids[i] and col[1] == str(expc) and col[4] == want, (i, col)
        assert col[3] == ",".join(map(str, counts[i].tolist())) and col[2] == repr(float(p[i, am[i]]))
    # an unwritable descriptor surfaces errno, not "failed (-1)"
    import os
    import pytest
    with open(tmp_path / "ro.tsv", "w") as f:
        pass
    fd = os.open(tmp_path / "ro.tsv", os.O_RDONLY)
    try:
        with pytest.raises(OSError, match="errno"):
            _native.text_table(os.fdopen(fd, "r", closefd=False), 10, [("i64", np.arange(10), ",")])
    finally:
        os.close(fd)


def test_output_kmers_group_larger_than_kernel_limit():
    """A subgenome of more than 64 chromosomes (scaffold-level assemblies): sp_kmer_ttest refuses it (SP_EUNSUP),
    so Cluster.output_kmers must take the numpy test instead of aborting -- and must still use the device at 64."""
    import io
    import numpy as np
    from scipy import stats as st
    from subphaser_amd import cluster

    class _Ctx:
        calls = 0

        def kmer_ttest(self, counts, lengths, groups):
            _Ctx.calls += 1
            if max(len(g) for g in groups) > cluster.TTEST_MAX_GROUP:
                raise RuntimeError("sp_kmer_ttest: a subgenome with 65 chromosomes (1..64 supported)")
            raise AssertionError("stub: not reached in this test")

    rng = np.random.default_rng(3)
    for na, expect_device in ((65, False), (64, True)):
        C, M = na + 5, 200

        class _Mat:
            pass
        mat = _Mat()
        mat.labels = ["c%03d" % i for i in range(C)]
        mat.k = 15
        mat.keys = rng.integers(0, 4 ** 15, M, dtype=np.int64).astype(np.uint64)
        mat.counts = rng.integers(0, 50, (M, C)).astype(np.uint32)
        mat.counts[:, :na] += rng.integers(0, 30, (M, 1)).astype(np.uint32)
        mat.lengths = rng.integers(10 ** 6, 10 ** 7, C)
        mat.freqs = mat.counts / mat.lengths.astype(np.float64)
        mat.ctx = _Ctx()
        sg = {c: ("SG1" if i < na else "SG2") for i, c in enumerate(mat.labels)}
        cl = cluster.Cluster(mat, n_clusters=2, sg_assigned=sg)
        before = _Ctx.calls
        buf = io.StringIO()
        if expect_device:
            try:
                cl.output_kmers(buf, max_pval=1.0)
            except AssertionError:
                pass
            assert _Ctx.calls == before + 1
            continue
        labels = cl.output_kmers(buf, max_pval=1.0)
        assert _Ctx.calls == before and len(labels.keys) == M
        rows = buf.getvalue().strip().split("\n")[1:]
        assert len(rows) == M
        for line in rows[:25]:
            km, sgname, p, ratios = line.split("\t")
            from subphaser_amd import kmer as kmerlib
            r = int(np.flatnonzero(mat.keys == kmerlib.encode_many([km])[0])[0])
            a, b = mat.freqs[r, :na], mat.freqs[r, na:]
            if b.mean() > a.mean():
                a, b = b, a
            assert np.isclose(float(p), st.ttest_ind(a, b)[1], rtol=1e-9, atol=1e-300)


def test_bench_selfchecks_multi_rank_runs_by_default(monkeypatch):
    """`bench.py --gpus N --steps K --warmup W` (what the driver runs) must validate its collectives without extra
    flags: the self-check is on whenever there is more than one rank, off for one rank unless asked for."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def parse(argv):
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        return bench.parse()
    assert bench.want_selfcheck(parse(["--gpus", "8", "--steps", "20", "--warmup", "5"]), 8) is True
    assert bench.want_selfcheck(parse(["--gpus", "2"]), 2) is True
    assert bench.want_selfcheck(parse(["--gpus", "8", "--no-dist-selfcheck"]), 8) is False
    assert bench.want_selfcheck(parse([]), 1) is False
    assert bench.want_selfcheck(parse(["--force-dist", "--dist-selfcheck"]), 1) is True
    assert "exchange wait" in bench.EXCHANGE_KEYS and "exchange+lengths" in bench.EXCHANGE_KEYS


def test_bench_jellyfish_leg(tmp_path, monkeypatch):
    """bench.py's cpu_baseline leg runs the reference's exact jellyfish commands (Jellyfish.py:697-699) when the
    binary exists and diffs its dump with the oracle's; this image has none -> "absent".  With a stand-in executable
    that answers `count` / `histo` / `dump` from the oracle, the leg must parse the dump and report equality."""
    import importlib.util
    import stat
    import sys
    import numpy as np
    import pyoracle as po
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(2)
    seqs = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3000)] for _ in range(2)]
    seqs[0] = np.concatenate([seqs[0], seqs[0][:1500]])
    k, L = 15, 2
    dumps = [po.count(s, k, L) for s in seqs]
    monkeypatch.setenv("PATH", str(tmp_path / "nobin"))
    assert bench.jellyfish_leg(seqs, ["c0", "c1"], k, L, 2, dumps) == "absent"
    fake = tmp_path / "bin" / "jellyfish"
    fake.parent.mkdir()
    fake.write_text("""#!%s
import sys
sys.path.insert(0, %r)
import numpy as np, pyoracle as po
a = sys.argv[1:]
if a[0] == "--version":
    print("jellyfish stand-in 0.0")
elif a[0] == "count":
    k = int(a[a.index("-m") + 1]); out = a[a.index("-o") + 1]
    seq = "".join(l.strip() for l in sys.stdin if not l.startswith(">"))
    open(out, "w").write("%%d\\n%%s" %% (k, seq))
elif a[0] == "histo":
    open(a[a.index("-o") + 1], "w").write("")
elif a[0] == "dump":
    jf = [x for x in a[1:] if x.endswith(".jf")][0]
    k, seq = open(jf).read().split("\\n", 1)
    keys, cnts = po.count(np.frombuffer(seq.encode(), np.uint8), int(k), int(a[a.index("-L") + 1]))
    order = np.random.default_rng(1).permutation(len(keys))       # hash order, like the real dump
    with open(a[a.index("-o") + 1], "w") as f:
        for i in order:
            f.write("%%s %%d\\n" %% (po.decode(int(keys[i]), int(k)), int(cnts[i])))
""" % (sys.executable, os.path.join(ROOT, "oracle")))
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake.parent) + os.pathsep + "/usr/bin:/bin")
    r = bench.jellyfish_leg(seqs, ["c0", "c1"], k, L, 2, dumps)
    assert r["dumps_equal_oracle_and_hip"] is True and r["dump_kmers"] == sum(len(d[0]) for d in dumps) > 0
    bad = [(d[0], d[1] + 1) for d in dumps]
    assert bench.jellyfish_leg(seqs, ["c0", "c1"], k, L, 2, bad)["dumps_equal_oracle_and_hip"] is False


def test_read_bed_errors_and_interval_merge(tmp_path, monkeypatch):
    """advisor r03: a short BED line must raise the 'not a BED line' error with or without pandas; identical
    intervals are one row (summed), in order of first appearance, like same-id FASTA lines in stack_matrix"""
    import builtins
    from subphaser_amd import seqs
    good = tmp_path / "g.bed"
    good.write_text("#c\nchr1\t5\t10\tx\nchr2 0 3\nchr1\t5\t10\n")
    bad = tmp_path / "b.bed"
    bad.write_text("chr1\t5\t10\nchr2\t5\n")
    names, code, st, en = seqs.read_bed(str(good))
    assert names == ["chr1", "chr2"] and code.tolist() == [0, 1, 0] and st.tolist() == [5, 0, 5] and en.tolist() == [10, 3, 10]
    with pytest.raises(ValueError, match="not a BED line"):
        seqs.read_bed(str(bad))
    real_import = builtins.__import__

    def no_pandas(name, *a, **kw):
        if name == "pandas":
            raise ImportError(name)
        return real_import(name, *a, **kw)
    monkeypatch.setattr(builtins, "__import__", no_pandas)
    assert seqs.read_bed(str(good))[1].tolist() == [0, 1, 0]
    with pytest.raises(ValueError, match="b.bed:2: not a BED line"):
        seqs.read_bed(str(bad))
    monkeypatch.undo()
    rows = seqs.IntervalRows(names, code, st, en)
    m, c = rows.merged(np.array([[1, 2], [3, 4], [10, 20]], np.int64))
    assert m.ids() == ["chr1:5-10", "chr2:0-3"] and c.tolist() == [[11, 22], [3, 4]]
    same, c2 = m.merged(c)
    assert same is m and c2 is c


def test_rank_tests_in_array_form_equal_scipy_row_by_row():
    """-test_method kruskal / mannwhitneyu (Cluster.py:178-194): the array forms in cluster._scipy_rows must give what
    the reference's per-k-mer scipy call gives, bit for bit -- ties, zeros, all-identical rows, unequal group sizes"""
    from scipy import stats
    from subphaser_amd.cluster import _scipy_rows
    rng = np.random.default_rng(3)
    for n1, n2 in ((7, 7), (3, 9), (1, 5), (10, 2), (9, 12)):
        M = 400
        a = np.round(rng.random((M, n1)) * 4) / 4 * (rng.random((M, n1)) < 0.7)
        b = np.round(rng.random((M, n2)) * 6) / 6 * (rng.random((M, n2)) < 0.5)
        a[::7] = 0.25
        b[::7] = 0.25
        a[1::7] = rng.random(a[1::7].shape)
        b[1::7] = rng.random(b[1::7].shape)
        for name in ("kruskal", "mannwhitneyu"):
            got = _scipy_rows(name, a, b)
            f = getattr(stats, name)
            exp = np.empty(M)
            for i in range(M):
                try:
                    exp[i] = f(a[i], b[i])[1]
                except ValueError as e:
                    assert "identical" in str(e) or "zero" in str(e)
                    exp[i] = 1.0
            assert np.array_equal(got, exp, equal_nan=True), (name, n1, n2, np.nanmax(np.abs(got - exp)))
    # wilcoxon (paired): scipy picks exact / permutation (ties or zeros, <= 13 pairs: all 2^n sign assignments, 3 ms to
    # 0.6 s per call) / asymptotic per call; the array form splits the rows the same way and enumerates the subset sums
    import warnings
    for n, M in ((2, 60), (5, 60), (7, 60), (10, 24), (13, 4), (14, 40), (60, 20)):
        a = np.round(rng.random((M, n)) * 5) / 5 * (rng.random((M, n)) < 0.8)
        b = np.round(rng.random((M, n)) * 5) / 5 * (rng.random((M, n)) < 0.6)
        a[::6] = rng.random(a[::6].shape)
        b[::6] = rng.random(b[::6].shape)
        a[1::6] = b[1::6]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = _scipy_rows("wilcoxon", a, b)
            exp = np.array([stats.wilcoxon(a[i], b[i])[1] for i in range(M)])
        assert np.array_equal(got, exp, equal_nan=True), ("wilcoxon", n, np.nanmax(np.abs(got - exp)))


def test_cli_entry_point_leaves_without_teardown(tmp_path):
    """`python -m subphaser_amd` / the `subphaser` script go through pipeline.cli: main(), flush, os._exit(0) -- output written before
    the call returns is on disk, buffered stdout / stderr are flushed, argparse's own exits (-h) keep their codes, and SP_SLOW_EXIT=1
    takes the interpreter's ordinary way out (atexit handlers run)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-m", "subphaser_amd", "-h"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "-sg_cfgs" in out.stdout or "sg_cfgs" in out.stdout
    marker, atexit_marker = tmp_path / "written", tmp_path / "atexit"
    prog = ("import atexit, sys\n"
            "from subphaser_amd import pipeline\n"
            "atexit.register(lambda: open(%r, 'w').write('x'))\n"
            "def fake_main(argv=None):\n"
            "    open(%r, 'w').write('done')\n"
            "    sys.stdout.write('buffered-out'); sys.stderr.write('buffered-err')\n"
            "pipeline.main = fake_main\n"
            "pipeline.cli([])\n"
            "print('not reached')\n") % (str(atexit_marker), str(marker))
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("SP_SLOW_EXIT", None)
    fast = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120, env=env)
    assert fast.returncode == 0 and fast.stdout == "buffered-out" and "buffered-err" in fast.stderr
    assert marker.read_text() == "done" and not atexit_marker.exists()
    slow = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120, env=dict(env, SP_SLOW_EXIT="1"))
    assert slow.returncode == 0 and "not reached" in slow.stdout and atexit_marker.exists()
