"""Multi-GPU path on CPU: world_size 2 / 3 / 8 over gloo, oracle-backed contexts.  Checks that the
position-sharded count (chromosomes cut into pieces with a k-1 halo) + slot-range-sharded filter
(all_to_all of byte-table slices, exact merge of the pieces of one chromosome) + position-sharded map
+ all-reduced window table reproduce the single-process result exactly."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class _Gen:
    """Minimal stand-in for subphaser_amd.synth.SynthGenome built from the toy genome."""

    def __init__(self, toy):
        self.labels = toy["labels"]
        self.sgs = toy["sgs"]
        self.chroms = [dict(label=l, length=len(toy["seqs"][l])) for l in self.labels]


def _worker(rank, world, port, out_dir, k=9, toy_kw=None, run_kw=None, shape=None):
    for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle_ctx import OracleContext, OracleDistContext
    from subphaser_amd import cluster
    from subphaser_amd.dist import DistHotPath
    from subphaser_amd.hotpath import HotPath
    from toygenome import make_shape_genome, make_toy_genome

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        toy = make_shape_genome(shape) if shape else make_toy_genome(seed=7, **(toy_kw or {}))
        n_sg = toy.get("n_sg", 2)
        gen = _Gen(toy)
        L = 3              # k = 9 -> 2^17 dense slots: small enough to ship as CPU tensors; k = 17 -> key-range path
        kw = dict(min_freq=30, bin_size=100, chunk_size=2000, window_size=2500)
        ctx = OracleDistContext()
        runner = DistHotPath(ctx, gen, dist, torch, k=k, lower_count=L, device=torch.device("cpu"), **kw, **(run_kw or {}))
        ascii_ = [np.frombuffer(toy["seqs"][gen.labels[pc["chrom"]]].encode(), np.uint8)[pc["start"]:pc["stop"]]
                  for pc in runner.local_pieces]
        if not runner.sparse and world > 1:     # pieces are cut inside chromosomes and cover every start exactly once
            allp = [p for r_ in runner.pieces for p in r_]
            assert sum(b_ - a_ for _, a_, b_ in allp) == sum(c["length"] for c in gen.chroms)
            assert len(allp) > len({c for c, _, _ in allp}) or world > len(gen.chroms)
        a = runner.count_and_filter(ascii_)
        if (run_kw or {}).get("shared_rows"):     # the bench path: rank 0 reads the matrix from the shared segment
            a0 = runner.count_and_filter(ascii_, host_rows_on_all_ranks=False)
            if rank == 0:
                assert (a0.keys == a.keys).all() and (a0.counts == a.counts).all()
            a0 = None
            runner.close()
        # single-process reference on the same genome
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
        if k > 15:      # key ranges are ordered by rank: the gathered matrix is already in ascending key order
            assert (a.keys[1:] > a.keys[:-1]).all()

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
        assert np.allclose(b.pvals, rb.pvals, rtol=0, atol=0)      # every rank holds every window and its test
        assert (b.sig == rb.sig).all() and (b.argmin == rb.argmin).all()
        if rank == 0:
            for x, y in zip(b.bins, rb.bins):
                assert x.shape == y.shape and (x == y).all()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_lpt_assign_balances():
    from subphaser_amd.dist import lpt_assign
    lens = [700, 650, 600, 580, 560, 500, 480, 450, 440, 400, 390, 380, 370, 360, 350, 340, 330, 320, 310, 300, 290]
    for n in (1, 2, 4, 8):
        owned = lpt_assign(lens, n)
        assert sorted(i for o in owned for i in o) == list(range(len(lens)))
        loads = [sum(lens[i] for i in o) for o in owned]
        assert max(loads) <= sum(lens) / n * 1.15 + 1


def _spawn(tmp_path, world, k, toy_kw=None, run_kw=None, shape=None):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), k, toy_kw, run_kw, shape), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / ("ok%d" % r)).exists()


def test_two_rank_hot_path_over_gloo(tmp_path):
    _spawn(tmp_path, 2, 9)


def test_two_rank_key_range_exchange_over_gloo(tmp_path):
    """k = 17: 64-bit keys, uneven all_to_all of key-range pieces instead of table slices."""
    _spawn(tmp_path, 2, 17)


def test_three_rank_key_range_exchange_over_gloo(tmp_path):
    _spawn(tmp_path, 3, 21)


def test_three_rank_hot_path_over_gloo(tmp_path):
    """three ranks, six chromosomes: two chromosomes are cut, their pieces' tables are merged at the filter;
    the matrix is also assembled in the shared host segment (every rank writes its own rows)"""
    _spawn(tmp_path, 3, 9, run_kw=dict(shared_rows=True))


def test_two_rank_overflow_lists_over_gloo(tmp_path):
    """k = 6 on a larger toy genome: counts >= 255 exist, so the byte tables travel with overflow pairs and the
    merge of the cut chromosome's two pieces has to add saturated bytes exactly."""
    _spawn(tmp_path, 2, 6, dict(chrom_len=150000, copies=200))


def test_eight_rank_wheat_shape_over_gloo(tmp_path):
    """21 chromosomes / 7 sets x 3 (the wheat structure) on 8 ranks: seven chromosomes are cut"""
    _spawn(tmp_path, 8, 9, shape="wheat")


@pytest.mark.parametrize("world", [2, 4])
def test_wheat_shape_two_and_four_ranks_over_gloo(tmp_path, world):
    """the wheat structure at the other rank counts the driver's scaling run uses (N = 1, 2, 4, 8)"""
    _spawn(tmp_path, world, 9, shape="wheat")


def test_plan_pieces_wheat_every_rank_count():
    """position plan of the wheat-like genome at N = 1, 2, 4, 8: every base owned exactly once, cuts inside a
    chromosome on multiples of the chunk size, slices balanced to within one chunk"""
    from subphaser_amd.dist import plan_pieces
    from subphaser_amd.synth import SynthGenome
    lens = [c["length"] for c in SynthGenome("wheat").chroms]
    align = 10_000_000
    for n in (1, 2, 4, 8):
        plan = plan_pieces(lens, n, align)
        assert len(plan) == n
        cover = [[] for _ in lens]
        for pieces in plan:
            for c, a, b in pieces:
                assert 0 <= a < b <= lens[c] and (a % align == 0) and (b % align == 0 or b == lens[c])
                cover[c].append((a, b))
        for c, iv in enumerate(cover):
            iv.sort()
            assert iv[0][0] == 0 and iv[-1][1] == lens[c] and all(x[1] == y[0] for x, y in zip(iv, iv[1:]))
        loads = [sum(b - a for _, a, b in pieces) for pieces in plan]
        assert max(loads) - min(loads) <= 2 * align, (n, loads)
