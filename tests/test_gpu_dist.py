"""The multi-GPU host path driving the REAL library, with 2 and 3 processes sharing the one GPU of the test box
(collectives staged through the host over gloo, tests/staged_dist.py): position pieces with a k-1 halo, byte
tables + overflow lists on the wire, sp_table_merge of a chromosome counted by two ranks, slot-range filter,
matrix assembled in the shared page-locked segment, device window table all-reduce -- against the single-process
oracle on the same genome."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, out_dir, k, engine, shape, toy_kw):
    for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle_ctx import OracleContext
    from staged_dist import StagedDist
    from subphaser_amd import _native, cluster
    from subphaser_amd.dist import DistHotPath
    from subphaser_amd.hotpath import HotPath
    from test_dist_gloo import _Gen
    from toygenome import make_shape_genome, make_toy_genome

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        toy = make_shape_genome(shape) if shape else make_toy_genome(seed=7, **(toy_kw or {}))
        n_sg = toy.get("n_sg", 2)
        gen = _Gen(toy)
        L = 3
        kw = dict(min_freq=30, bin_size=100, chunk_size=2000, window_size=2500)
        ctx = _native.Context(0)
        runner = DistHotPath(ctx, gen, StagedDist(dist), torch, k=k, lower_count=L, engine=engine,
                             device=torch.device("cuda", 0), shared_rows=True, **kw)
        d_pieces = []
        for pc in runner.local_pieces:
            a = np.frombuffer(toy["seqs"][gen.labels[pc["chrom"]]].encode(), np.uint8)[pc["start"]:pc["stop"]]
            p = ctx.dev_alloc(len(a))
            ctx.host_to_dev(p, np.ascontiguousarray(a))
            d_pieces.append(p)
        ctx.sync()
        a = runner.count_and_filter(d_pieces)
        a0 = runner.count_and_filter(d_pieces, host_rows_on_all_ranks=False)   # the bench path: shared segment
        a0.wait()      # (asynchronous since round 5: copy stream + barrier; every rank calls it)
        if rank == 0:
            assert (np.asarray(a0.keys) == a.keys).all() and (np.asarray(a0.counts) == a.counts).all()
        a0 = None
        ref = OracleContext()
        ref.genome_reset(len(gen.labels))
        for i, l in enumerate(gen.labels):
            ref.genome_add(i, toy["seqs"][l])
        ref.count(k, L)
        hp = HotPath(ref, gen.labels, [c["length"] for c in gen.chroms], gen.sgs, k=k, lower_count=L, **kw)
        nu, nr, nh = ref.filter(*hp.csr, hp.min_fold, hp.baseline, hp.min_freq, hp.max_freq, hp.ratio)
        rk, rc, _, rt = ref.filter_fetch(nr)
        assert (a.n_union, a.n_rows, a.n_hist) == (nu, nr, nh), (rank, a.n_union, a.n_rows, a.n_hist, nu, nr, nh)
        assert a.kmer_lengths.tolist() == ref.lengths().tolist()
        o = np.argsort(a.keys, kind="stable")
        assert (a.keys[o] == rk).all() and (a.counts[o] == rc).all()
        assert nr > 0

        class _Mat:
            pass
        mat = _Mat()
        mat.labels, mat.keys, mat.k = gen.labels, rk, k
        mat.freqs = rc.astype(np.float64) / ref.lengths().astype(np.float64)
        cl = cluster.Cluster(mat, n_clusters=n_sg, sg_assigned=toy["sg_assigned"])
        labels = cl.output_kmers(open(os.devnull, "w"), max_pval=0.05)
        b = runner.map_and_enrich(labels, n_sg, gather_bins=True)
        rb = hp.map_and_enrich(labels, n_sg)
        assert b.coords == rb.coords and len(rb.coords) > 0
        assert (b.window_counts == rb.window_counts).all()
        assert b.n_mapped == rb.n_mapped
        assert np.allclose(b.pvals, rb.pvals, rtol=1e-6, atol=1e-300)
        assert (b.sig == rb.sig).all() and (b.argmin == rb.argmin).all()
        if rank == 0:
            for x, y in zip(b.bins, rb.bins):
                assert x.shape == y.shape and (x == y).all()
        runner.close()
        for p in d_pieces:
            ctx.dev_free(p)
        ctx.close()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _spawn(tmp_path, world, k, engine=0, shape=None, toy_kw=None):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), k, engine, shape, toy_kw), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / ("ok%d" % r)).exists()


def test_two_processes_one_gpu_dense(gpu_ctx, tmp_path):
    _spawn(tmp_path, 2, 11)


def test_three_processes_one_gpu_wheat_shape_engine2(gpu_ctx, tmp_path):
    """21 chromosomes / 7 sets x 3 on three ranks, LDS partition engine, k = 13 (2^25-slot byte tables)"""
    _spawn(tmp_path, 3, 13, engine=2, shape="wheat")


def test_two_processes_one_gpu_overflow_lists(gpu_ctx, tmp_path):
    """k = 6 on a larger toy genome: saturated bytes + overflow pairs merged by sp_table_merge on the device"""
    _spawn(tmp_path, 2, 6, toy_kw=dict(chrom_len=150000, copies=200))


def test_two_processes_one_gpu_key_range(gpu_ctx, tmp_path):
    """k = 17: sorted (k-mer, count) lists, splitters, uneven all_to_all of key-range pieces"""
    _spawn(tmp_path, 2, 17)


def test_eight_processes_one_gpu_wheat_shape_position_pieces(gpu_ctx, tmp_path):
    """Rehearsal of the first 8-GPU lease (round 6): EIGHT ranks on the wheat shape (21 chromosomes / 7 sets x 3), byte-table
    engine: eight position pieces = seven chromosomes cut inside, every cut chromosome merged by sp_table_merge on the rank
    that owns the slot range, slot-range filter on all eight, shared-segment rows, window all-reduce -- against the
    single-process oracle.  (k = 13: the same code as k = 15 with 2^25-slot tables; 28 pieces x 512 MiB through the
    host-staged gloo all_to_all of this test would take minutes.)"""
    _spawn(tmp_path, 8, 13, engine=2, shape="wheat")


def test_eight_processes_one_gpu_wheat_shape_key_range(gpu_ctx, tmp_path):
    """... and k = 17 on eight ranks: whole chromosomes dealt by LPT, sorted lists, splitters, the uneven all_to_all of
    key-range pieces."""
    _spawn(tmp_path, 8, 17, shape="wheat")


def test_bench_line_single_and_forced_dist(tmp_path):
    """bench.py end to end on the `small` genome: the JSON contract (metric / value / roofline / cpu_baseline with the
    jellyfish leg / verified), then the multi-GPU code path through RCCL with one rank (`--force-dist`) including
    `--dist-selfcheck` -- the checks the first 8-GPU lease will rely on (rccl_ranks, pieces_per_rank, selfcheck ok)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "small", "--steps", "2", "--warmup", "1",
                          "--cpu-sample-mb", "5"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype",
                "data", "config", "roofline", "cpu_baseline", "stage_roofline", "traffic_commit"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["verified"] is True and d["unit"] == "Gbases/s"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["traffic"] is None
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["jellyfish"] in ("absent",) or isinstance(d["cpu_baseline"]["jellyfish"], dict)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "small", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--force-dist", "--dist-selfcheck"], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d2 = json.loads(out.stdout.strip().splitlines()[-1])
    assert d2["rccl_ranks"] == 1 and d2["dist_selfcheck"]["ok"] is True and d2["pieces_per_rank"][0]["rank"] == 0
    assert d2["exchange_ms_per_rank"][0]["rank"] == 0 and d2["exchange_ms_per_rank"][0].get("exchange wait", 0.0) >= 0
    assert d2["stage_ms_per_rank"][0]["rank"] == 0 and d2["stage_ms_per_rank"][0]["labels"] > 0 and d2["stage_ms_per_rank"][0]["map+stack"] > 0
    assert d2["pieces_per_rank"][0]["bases"] > 0 and d2["config"]["differential_kmers"] == d["config"]["differential_kmers"]
    assert d2["config"]["mapped_positions"] == d["config"]["mapped_positions"]
