#!/usr/bin/env python3
"""bench.py -- whole hot path (k-mer count + differential filter + window map + enrichment)
on a synthetic genome resident in HBM.  One "step" = one full pass over the genome.

    python bench.py --gpus 1 --steps K --warmup W [--config wheat|peanut|ara|small|tiny]

Prints ONE JSON line (rank 0).  metric = genome Gbases/s (BASELINE.json); dtype names
the arithmetic type of the dominant work (u32 counters); `roofline` prices the dominant
kernel against 8 TB/s HBM using the ALGORITHMIC bytes of SURVEY.md section 8(d);
`cpu_baseline` is the oracle (a C port, oracle/sp_oracle.c) timed on this box's host
cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("SP_BENCH_CONFIG", "wheat"))
    ap.add_argument("--scale", type=float, default=1.0, help="scale chromosome lengths (debugging only)")
    ap.add_argument("--engine", type=int, default=int(os.environ.get("SP_ENGINE", "0")))
    ap.add_argument("-k", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-GPU code path (RCCL process group, table exchange) even with one rank")
    ap.add_argument("--dist-selfcheck", action="store_true",
                    help="before the timed run, every rank takes part in one distributed pass over the `small` synthetic "
                         "genome (9 chromosomes, cut inside chromosomes) and rank 0 compares tallies, lengths, matrix rows, "
                         "windows and calls with a single-process pass over the same genome.  ON BY DEFAULT with more than "
                         "one rank (the line then carries dist_selfcheck.ok); this flag turns it on for --force-dist runs")
    ap.add_argument("--no-dist-selfcheck", action="store_true", help="skip the self-check of a multi-rank run")
    ap.add_argument("--cpu-sample-mb", type=float, default=float(os.environ.get("SP_CPU_SAMPLE_MB", "1000")))
    return ap.parse_args()


# host wall-clock phases of DistHotPath that contain a collective (reported per rank as `exchange_ms_per_rank`)
EXCHANGE_KEYS = ["count(+exchange issue)", "exchange wait", "exchange+lengths", "merge+lengths", "rows to shared host memory",
                 "gather rows", "windows all-reduce+enrich"]
# ... and the phases without one, so that a scaling curve explains itself: what each rank spent packing, filtering, building the
# label tables (on every rank: the part that does not shrink with N) and mapping (`stage_ms_per_rank`)
LOCAL_KEYS = ["pack", "count", "split+export", "filter+fetch", "labels", "map+stack"]


def want_selfcheck(args, world):
    """A run with more than one rank validates its own collectives before anything is timed unless told not to: the
    driver's `bench.py --gpus N --steps K --warmup W` passes no extra flag, and its line must carry dist_selfcheck.ok."""
    return bool(args.dist_selfcheck or (world > 1 and not args.no_dist_selfcheck))


def algorithmic_bytes(kernel, bases, nslots, C, S, extra):
    """SURVEY.md 8(d) per-unit figures x the units one launch processes."""
    k = extra.get("k", 15)
    if k > 16 and kernel.startswith("s3_"):
        return 16.25 * bases                      # 0.25 B read + 8 B key read + 4 B counter read + 4 B write per base
    if kernel.startswith("sps_"):                 # matrix + filter over lists: (Kb + 4) B per dumped k-mer + the rows
        return extra.get("sum_dump", 0) * (12.0 if k > 15 else 8.0) + extra.get("M", 0) * C * 8.0
    if kernel == "k5_map_sparse":
        return 9.25 * bases + extra.get("nbins", 0) * S * 4
    if kernel.startswith("k1_") or kernel.startswith("c2_"):
        return 8.25 * bases                       # 0.25 B read + 4 B counter read + 4 B counter write per base
    if kernel == "k5_map":
        return 1.25 * bases + extra.get("nbins", 0) * S * 4
    if kernel == "k3_eval":
        return extra.get("sum_dump", 0) * 8.0 + extra.get("M", 0) * C * 8.0
    if kernel == "k0_pack":
        return 1.375 * bases                      # 1 B ASCII read + 0.375 B packed write (not in 8(d): input there is 2-bit)
    if kernel == "k2_lengths":
        return 4.0 * nslots
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU,
        # RCCL rendezvous on the loopback address) and hand over to them
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from subphaser_amd import _native, cluster
    from subphaser_amd.hotpath import HotPath
    from subphaser_amd.synth import SynthGenome
    rccl_ranks = None
    if dist is not None:     # how many ranks RCCL really connects: a 1-element all-reduce of ones
        one = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        if rccl_ranks != world:
            raise SystemExit("RCCL all-reduce over %d ranks returned %d" % (world, rccl_ranks))

    gen = SynthGenome(args.config, args.scale)
    ctx = _native.Context(local_rank)
    C, S = len(gen.chroms), gen.S

    selfcheck = None
    if dist is not None:
        from subphaser_amd.dist import DistHotPath
        if want_selfcheck(args, world):
            selfcheck = dist_selfcheck(ctx, dist, torch, args, rank)
        runner = DistHotPath(ctx, gen, dist, torch, k=args.k, engine=args.engine)
    else:
        runner = None

    # ---- synthetic genome straight into HBM (untimed) ----------------------------------------
    t0 = time.perf_counter()
    d_ascii = []
    if runner is None:
        pieces = [dict(chrom=i, start=0, end=c["length"], stop=c["length"]) for i, c in enumerate(gen.chroms)]
    else:
        pieces = runner.local_pieces     # contiguous slices of the concatenated genome (k-1 halo inside a chromosome)
    for pc in pieces:
        c = gen.chroms[pc["chrom"]]
        n = pc["stop"] - pc["start"]
        p = ctx.dev_alloc(max(n, 1))
        ctx.synth_chrom_range(p, c["length"], pc["start"], n, gen.seed, c["set_id"], c["sg_id"], S, c["chrom_id"], c["exchange"])
        d_ascii.append(p)
    ctx.sync()
    t_synth = time.perf_counter() - t0

    if runner is None:
        hp = HotPath(ctx, gen.labels, [c["length"] for c in gen.chroms], gen.sgs, k=args.k, engine=args.engine)

        def first_half(overlap=False):
            return hp.count_and_filter(d_ascii, overlap=overlap)

        def second_half(lab):
            return hp.map_and_enrich(lab, S)
    else:
        def first_half(everywhere=False):
            return runner.count_and_filter(d_ascii, host_rows_on_all_ranks=everywhere)

        def second_half(lab):
            return runner.map_and_enrich(lab, S)

    # ---- labels: the reference's unchanged Cluster step, computed once, untimed ---------------
    r1 = first_half(True) if runner is not None else first_half()

    class _Mat:
        pass
    mat = _Mat()
    mat.labels, mat.keys, mat.k = gen.labels, r1.keys, args.k
    mat.freqs = r1.counts.astype(np.float64) / np.asarray(r1.kmer_lengths, np.float64)
    mat.counts, mat.lengths, mat.ctx = r1.counts, np.asarray(r1.kmer_lengths, np.int64), ctx    # t-test on the device
    if r1.n_rows == 0:
        raise SystemExit("synthetic genome produced 0 differential k-mers")
    cl = cluster.Cluster(mat, n_clusters=S, sg_assigned=gen.sg_assigned)
    kmer_labels = cl.output_kmers(open(os.devnull, "w"), max_pval=0.05)
    del mat

    def step():
        # the M x C matrix goes to the host on a copy stream while the map stage runs; it is complete (wait) before
        # the step counts as done
        a = first_half(overlap=True) if runner is None else first_half()
        b = second_half(kmer_labels)
        a.wait()
        return a, b

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        step()
    (hp if runner is None else runner).wall.clear()
    ctx.prof_reset()
    ctx.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a, b = step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    prof = ctx.prof_report()
    # the same steps once more WITHOUT the two HIP events per launch that the roofline figures are made of (they cost
    # ~1 % of a wheat-like pass and ~10 % of an Arabidopsis-like one): reported next to the contract's figure, never as it
    n_plain = min(args.steps, 3)
    wall_timed = dict((hp if runner is None else runner).wall)      # (the stage walls of the timed steps only)
    barrier()
    t1 = time.perf_counter()
    for _ in range(n_plain):
        a, b = step()
    barrier()
    dt_plain = time.perf_counter() - t1
    if dist is not None:
        t = torch.tensor([dt, dt_plain], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_plain = float(t[0].item()), float(t[1].item())
    ms_per_step = dt / args.steps * 1e3
    ms_per_step_no_events = dt_plain / n_plain * 1e3
    gbases = gen.total_bases / (dt / args.steps) / 1e9
    # what every rank spent waiting for / issuing the table (or key-range) exchange and the other collectives
    rank_wall = rank_local = None
    if runner is not None:
        mine = torch.tensor([wall_timed.get(k_, 0.0) / args.steps * 1e3 for k_ in EXCHANGE_KEYS], dtype=torch.float64, device="cuda")
        allw = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allw, mine)
        rank_wall = [{"rank": r_, **{k_: round(float(v_), 3) for k_, v_ in zip(EXCHANGE_KEYS, w_.tolist()) if v_}}
                     for r_, w_ in enumerate(allw)]
        mine = torch.tensor([wall_timed.get(k_, 0.0) / args.steps * 1e3 for k_ in LOCAL_KEYS], dtype=torch.float64, device="cuda")
        allw = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allw, mine)
        rank_local = [{"rank": r_, **{k_: round(float(v_), 3) for k_, v_ in zip(LOCAL_KEYS, w_.tolist()) if v_}}
                      for r_, w_ in enumerate(allw)]

    # sum over chromosomes of D_c = distinct k-mers with count >= L (the lines of the jellyfish dumps)
    n_dumped = sum(ctx.dump_size(i) for i in range(len(pieces)))
    if dist is not None and world > 1:
        tdump = torch.tensor([n_dumped], dtype=torch.int64, device="cuda")
        dist.all_reduce(tdump)
        n_dumped = int(tdump.item()) // world      # one rank's slot / key range

    if rank != 0:
        if runner is not None:
            a = b = None
            runner.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        return

    # ---- roofline of the dominant kernel (HIP events per launch, on the context's stream) ------
    nslots = (1 << (2 * args.k - 1) if args.k % 2 else 1 << (2 * args.k)) if args.k <= 15 else 0
    n_local = len(pieces)
    local_bases = sum(pc["stop"] - pc["start"] for pc in pieces)
    extra = dict(nbins=sum(len(x) for x in b.bins) / max(1, len(b.bins)), M=a.n_rows,
                 sum_dump=int(n_dumped), k=args.k)
    # physical HBM bytes per launch from the committed PMC passes (profiles/, tools/pmc_round.sh).  They describe the
    # kernels as they were when the passes ran: the file carries the hash of csrc/ and the commit, and the numbers are
    # quoted only while the sources still hash to it -- otherwise `traffic` is null (stale evidence is worse than none)
    tj, traffic_commit = {}, None
    try:
        from subphaser_amd._native import csrc_fingerprint
        # one table per measured (genome, k): profiles/r04_wheat_pmc.json, r04_wheat_k17_pmc.json, r04_peanut_pmc.json ...
        pmc = "%s_pmc.json" % args.config if args.k == 15 else "%s_k%d_pmc.json" % (args.config, args.k)
        for rnd in ("r06", "r05", "r04", "r03"):
            path = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, pmc))
            if os.path.exists(path):
                pj = json.load(open(path))
                if world == 1 and pj.get("csrc_sha16") == csrc_fingerprint():
                    tj, traffic_commit = pj["kernels"], pj.get("commit")
                break
    except (OSError, ValueError, KeyError):
        pass

    def price(names, label, per_chrom=True):
        """achieved algorithmic GB/s of one kernel, or of a chain of kernels launched once per chromosome"""
        sts = [prof[n] for n in names if n in prof]
        if not sts:
            return None
        # a chain runs once per local chromosome, a whole-genome kernel (k5_map since round 3, the filter) once per step
        # (a chain holds kernels launched several times per chromosome -- scans -- and some launched only where needed:
        # the unit is the chromosome whenever any of them runs at least once per chromosome)
        most = max(prof[n]["calls"] for n in names if n in prof) / args.steps
        units = (n_local if most >= n_local else max(1, int(round(most)))) if per_chrom else 1
        per_launch = local_bases / units
        alg = algorithmic_bytes(names[0], per_launch, nslots, C, S, extra)
        if not alg:
            return None
        avg_ev = sum(st["ms"] for st in sts) / args.steps / units / 1e3    # HIP-event time of one pass of the chain
        # chains that run side by side on several streams (round 4): the events of one chain also clock the time it
        # shares the chip with its neighbours, so the duration that prices the chain is its share of the stage's wall
        # time (lane_scale = stage wall / sum of the stage's event times); both are on the line
        avg_s = avg_ev * (lane_scale if names[0] in LANE_KERNELS else 1.0)
        traffic = None
        # the profiler's labels -> the kernel symbols rocprofv3 reports (one kernel launched under two labels)
        sym = {"k5_map": "k5_map2", "k5_map_sparse": "k5_map_sparse2", "sps_join": "sps_join_blk",
               "c2_hist_sample": "c2_hist_fine", "ovf_place_list": "ovf_place",
               "sps_emit_hist": "sps_emit", "k3_emit_hist": "k3_emit", "s3_hist1_sample": "s3_hist1",
               "s3_hist2_sample": "s3_hist2"}
        def _sym(n):       # the symbol a label's kernel has in the PMC table (batched wrappers carry a _b suffix)
            # (round 6: the list counter of small genomes is c2_count_list16 / c2_count_list16_b under the label c2_count_list)
            for cand in (sym.get(n, n), n, sym.get(n, n) + "_b", n + "_b", n + "16_b", n + "16"):
                if cand in tj:
                    return cand
            return None
        if tj and all(_sym(n) for n in names if n in prof):
            # PMC bytes are per launch; launches per pass of the chain come from THIS run's launch counts
            traffic = int(sum((tj[_sym(n)].get("read_bytes", 0) + tj[_sym(n)].get("write_bytes", 0))
                              * (prof[n]["calls"] / args.steps / units) for n in names if n in prof))
        ach = alg / avg_s
        return {"bound": "hbm", "kernel": label, "achieved": round(ach / 1e9, 3), "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK, 5), "traffic": traffic,
                "avg_launch_ms": round(avg_s * 1e3, 4), "alg_bytes_per_launch": int(alg),
                **({"avg_launch_ms_events": round(avg_ev * 1e3, 4), "lane_scale": round(lane_scale, 4)}
                   if names[0] in LANE_KERNELS and lane_scale != 1.0 else {})}

    COUNT_CHAIN = [n for n in ("c2_hist_sample", "c2_hist_fine", "c2_offsets", "c2_part1", "c2_tiles", "c2_part2", "c2_spans",
                               "c2_count16", "c2_count", "c2_count_list", "ovf_scan", "ovf_place", "ovf_place_list", "k1_count_atomic",
                               "k1_narrow") if n in prof and prof[n]["calls"] >= args.steps]      # (batched small genomes: one launch per group of chromosomes)
    if args.k > 15:     # the MSD-partition engine for 64-bit keys: one chain of s3_* kernels per chromosome
        COUNT_CHAIN = sorted(n for n in prof if n.startswith("s3_"))
    # byte-table filter, or the list filter (k > 15, and engine 3 on small genomes at k <= 15)
    FILTER_CHAIN = ["k3_eval", "k3_slow"] if "k3_eval" in prof else \
        sorted((n for n in prof if n.startswith("sps_") and "hash" not in n and "pair" not in n), key=lambda n: n != "sps_join")
    # one process: the pack kernels and the count chains overlap on up to four streams (sp_count's lanes; round 4: also
    # the k > 15 chains, three chromosomes in flight)
    LANE_KERNELS, lane_scale = set(), 1.0
    wall_pc = (wall_timed.get("pack+count", 0.0) / args.steps * 1e3) if runner is None else 0.0
    if wall_pc > 0:
        lk = [n for n in COUNT_CHAIN + ["k0_pack"] if n in prof]
        ev = sum(prof[n]["ms"] for n in lk) / args.steps
        if ev > 0 and wall_pc / ev < 0.9:
            LANE_KERNELS, lane_scale = set(lk), wall_pc / ev
    dom = max(prof.items(), key=lambda kv: kv[1]["ms"] * (lane_scale if kv[0] in LANE_KERNELS else 1.0)) if prof else None
    roofline = None
    if dom:
        name = dom[0]
        if name in COUNT_CHAIN:
            # SURVEY 8(d) prices the COUNT (8.25 B/base) as one unit; the engine spreads it over a chain of
            # kernels launched once per chromosome, so the chain is priced together, never one link alone
            roofline = price(COUNT_CHAIN, "count engine: " + "+".join(COUNT_CHAIN))
        else:
            roofline = price([name], name, per_chrom=name not in FILTER_CHAIN)
        if roofline:
            # how "dominant" was decided (a reader of profiles/ sees c2_part1 lead by rocprof TOTAL: its 21 launches overlap
            # on the counting lanes, so their event times add up to more than the stage's wall, which is what is compared)
            roofline["dominant_by"] = ("stage wall: per-kernel event time per step, the overlapped count-chain kernels scaled by "
                                       "pack+count wall / their event sum (%.2f)" % lane_scale) if LANE_KERNELS else "kernel event time per step"
    # the three stages of the path, each against its own SURVEY 8(d) bytes (context for the line above)
    stage_roofline = {}
    for label, names in (("count", COUNT_CHAIN), ("filter", FILTER_CHAIN), ("map", ["k5_map"] if args.k <= 15 else ["k5_map_sparse"])):
        pr = price(names, "+".join(names), per_chrom=label != "filter") if names else None
        if pr:
            stage_roofline[label] = {"kernels": pr["kernel"], "achieved_GBps": pr["achieved"], "frac": pr["frac"],
                                     "chain_ms": pr["avg_launch_ms"], "traffic": pr["traffic"]}
    # the whole step against SURVEY 8(d): count + map bytes per base (9.5 at k <= 15, 25.5 with 64-bit keys) + the matrix
    # term (sum of the dump sizes x key + counter bytes, the M x C rows), over the wall time of a step -- the figure the
    # north star's ">= 60 % of the HBM roofline" is stated in
    extra_all = dict(extra, sum_dump=extra["sum_dump"] * world)       # (a rank's share of the dumps x the ranks)
    step_alg = ((algorithmic_bytes("c2_" if args.k <= 15 else "s3_", gen.total_bases, nslots, C, S, dict(extra_all, k=max(args.k, 17) if args.k > 15 else args.k)) or 0) +
                (algorithmic_bytes("k5_map" if args.k <= 15 else "k5_map_sparse", gen.total_bases, nslots, C, S, extra_all) or 0) +
                (algorithmic_bytes("sps_" if args.k > 15 else "k3_eval", gen.total_bases, nslots, C, S, extra_all) or 0))
    step_roofline = {"alg_bytes": int(step_alg), "achieved_GBps": round(step_alg / (ms_per_step / 1e3) / 1e9, 3),
                     "frac": round(step_alg / (ms_per_step / 1e3) / (HBM_PEAK * world), 5), "peak_GBps": HBM_PEAK * world / 1e9}
    stages = {k_: {"calls_per_step": v["calls"] / args.steps, "ms_per_step": round(v["ms"] / args.steps, 3)}
              for k_, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ---------------------
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(ctx, gen, d_ascii, kmer_labels, args)

    out = {
        "metric": "genome Gbases/s k-mer+enrichment, %s k=%d 1Mb windows" % (args.config, args.k),
        "value": round(gbases, 4), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "ms_per_step_no_events": round(ms_per_step_no_events, 3),
        "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s-like synthetic genome, %d chromosomes, %.3f Gbases, k=%d, lower_count=3, "
                               "q=200 f=2, 10-kb bins, 1-Mb windows" % (args.config, C, gen.total_bases / 1e9, args.k),
                   "engine": args.engine, "differential_kmers": int(a.n_rows), "union_kmers": int(a.n_union),
                   "sig_kmers": int(len(kmer_labels.keys)), "windows": int(len(b.window_counts)),
                   "mapped_positions": int(b.n_mapped), "parallelism": ("single GPU" if world == 1 else "genome-position-sharded count/map + slot-range-sharded filter x%d" % world)},
        "rccl_ranks": rccl_ranks, "dist_selfcheck": selfcheck,
        # how the M x C matrix reached rank 0
        "rows": (getattr(runner, "rows_handover", None) if runner is not None else "device->host copy stream"),
        "exchange_ms_per_rank": rank_wall, "stage_ms_per_rank": rank_local,
        "pieces_per_rank": ([{"rank": r_, "pieces": len(p_), "bases": int(sum(e_ - a_ for _, a_, e_ in p_))}
                             for r_, p_ in enumerate(runner.pieces)] if runner is not None else None),
        "traffic_commit": traffic_commit, "roofline": roofline, "step_roofline": step_roofline, "cpu_baseline": cpu, "verified": bool(cpu and cpu.get("verified")), "stage_roofline": stage_roofline, "stages": stages, "synth_s": round(t_synth, 2),
        "host_wall_ms_per_step": {k_: round(v / args.steps * 1e3, 2)
                                  for k_, v in wall_timed.items()},
    }
    if runner is not None:
        a = b = None
        runner.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL prints a banner through C stdio; flush it first so that the JSON is the LAST line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if world > 1:
        time.sleep(1.0)       # let the other ranks' buffered library output drain first
    print(json.dumps(out))
    sys.stdout.flush()


def cpu_baseline(ctx, gen, d_ascii, kmer_labels, args):
    """Oracle (C port of the same path) on the host cores over the first `cpu_sample_mb` Mb of
    every chromosome of the first homoeologous set.  Reported, never the target.  The oracle's results
    are not thrown away: the HIP path then redoes the same sample and dumps, matrix rows and bin counts
    must be identical (`verified`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from subphaser_amd.config import sets_to_csr
    cores = max(1, len(os.sched_getaffinity(0)) // 2)      # one thread per physical core: SMT siblings only add
    n = int(args.cpu_sample_mb * 1e6)                      # page-fault and memory-bus contention to this workload
    first = gen.sgs[0]
    labs = [c for unit in first for c in unit]
    idx = [gen.labels.index(l) for l in labs]
    seqs = [ctx.dev_to_host(d_ascii[i], min(n, gen.chroms[i]["length"])) for i in idx]
    bases = sum(len(s) for s in seqs)
    fkw = (2.0, 1, 200 * bases / gen.total_bases, 1e9, 1.0)
    t0 = time.perf_counter()
    dumps = [po.count(s, args.k, 3, nthreads=cores) for s in seqs]
    t_count = time.perf_counter() - t0
    try:
        filt = po.filter_dumps(dumps, [first], labs, *fkw)
    except ValueError:
        filt = None
    t_filter = time.perf_counter() - t0 - t_count
    S = gen.S
    obins = [po.map_bins(s, args.k, kmer_labels.keys, kmer_labels.sg_idx, S, 10000, 10_000_000, nthreads=cores)
             for s in seqs]
    t_map = time.perf_counter() - t0 - t_count - t_filter
    total = time.perf_counter() - t0
    # ---- the same sample through the HIP path (untimed): bit-exact or the bench fails ---------------
    ctx.genome_reset(len(seqs))
    for i, s in enumerate(seqs):
        ctx.genome_add(i, s)
    ctx.count(args.k, 3, args.engine)
    checked = {"dump_kmers": 0, "matrix_rows": 0, "bins": 0}
    for i, (ok, oc) in enumerate(dumps):
        gk, gc = ctx.dump(i)
        if gk.shape != ok.shape or not (gk == ok).all() or not (gc == oc).all():
            raise SystemExit("VERIFY FAILED: dump of sample chromosome %d differs from the oracle" % i)
        checked["dump_kmers"] += int(gk.size)
    if filt is not None:
        nu, nr, nh = ctx.filter(*sets_to_csr([first], labs), *fkw)
        gkeys, gcounts, _, gtot = ctx.filter_fetch(nr, want_freqs=False, sort=True)
        if (nu, nr, nh) != (filt.n_union, len(filt.keys), len(filt.hist)) or not (gkeys == filt.keys).all() \
                or not (gcounts == filt.counts).all() or not (gtot == filt.tot).all():
            raise SystemExit("VERIFY FAILED: matrix / filter rows of the sample differ from the oracle")
        checked["matrix_rows"] = int(nr)
    ctx.labels_set(kmer_labels.keys, kmer_labels.sg_idx, S)
    for i, ob in enumerate(obins):
        gb, gn = ctx.map_bins(i, 10000, 10_000_000)
        if gb.shape != ob[0].shape or not (gb == ob[0]).all() or gn != ob[2]:
            raise SystemExit("VERIFY FAILED: bin counts of sample chromosome %d differ from the oracle" % i)
        checked["bins"] += int(gb.shape[0])
    jf = jellyfish_leg(seqs, labs, args.k, 3, cores, dumps)
    return {"value": round(bases / total / 1e9, 5), "unit": "Gbases/s", "cores": cores, "kind": "port", "jellyfish": jf,
            "sample": "%s of the %d chromosomes of homoeologous set 1 (%.1f Mbases): "
                      "oracle count (thread-partitioned) %.2fs + matrix/filter %.2fs + map %.2fs"
                      % ("all" if all(len(s) == gen.chroms[i]["length"] for s, i in zip(seqs, idx))
                         else "first %.0f Mb of each" % args.cpu_sample_mb, len(seqs), bases / 1e6, t_count,
                         t_filter, t_map),
            "verified": True, "verified_items": checked}


def dist_selfcheck(ctx, dist, torch, args, rank):
    """One distributed pass over the `small` synthetic genome (9 chromosomes of ~30 Mb: at 8 ranks every rank gets a
    ~34-Mb slice, so chromosomes are cut and the halo / merge / slot-range code runs) against a single-process pass
    on rank 0.  Cheap (a fraction of a second) and loud: any difference ends the run with the first mismatch."""
    import numpy as np
    from subphaser_amd import cluster
    from subphaser_amd.dist import DistHotPath
    from subphaser_amd.hotpath import HotPath
    from subphaser_amd.synth import SynthGenome
    g = SynthGenome("small")
    kw = dict(k=args.k, lower_count=3, min_freq=20)

    def synth(pieces):
        out = []
        for pc in pieces:
            c = g.chroms[pc["chrom"]]
            n = pc["stop"] - pc["start"]
            p = ctx.dev_alloc(max(n, 1))
            ctx.synth_chrom_range(p, c["length"], pc["start"], n, g.seed, c["set_id"], c["sg_id"], g.S, c["chrom_id"], c["exchange"])
            out.append(p)
        ctx.sync()
        return out

    def labels_of(res):
        class _Mat:
            pass
        m = _Mat()
        m.labels, m.keys, m.k = g.labels, res.keys, args.k
        m.freqs = res.counts.astype(np.float64) / np.asarray(res.kmer_lengths, np.float64)
        return cluster.Cluster(m, n_clusters=g.S, sg_assigned=g.sg_assigned).output_kmers(open(os.devnull, "w"), max_pval=0.05)

    runner = DistHotPath(ctx, g, dist, torch, engine=args.engine, **kw)
    ptrs = synth(runner.local_pieces)
    a = runner.count_and_filter(ptrs, host_rows_on_all_ranks=True)
    lab = labels_of(a)
    b = runner.map_and_enrich(lab, g.S)
    runner.close()
    for p in ptrs:
        ctx.dev_free(p)
    ok, why = True, ""
    if rank == 0:
        whole = [dict(chrom=i, start=0, stop=c["length"]) for i, c in enumerate(g.chroms)]
        ptrs = synth(whole)
        hp = HotPath(ctx, g.labels, [c["length"] for c in g.chroms], g.sgs, engine=args.engine if args.k > 15 else 2, **kw)
        ra = hp.count_and_filter(ptrs, sort=True)
        rb = hp.map_and_enrich(labels_of(ra), g.S)
        for p in ptrs:
            ctx.dev_free(p)
        o = np.argsort(a.keys, kind="stable")
        checks = [("tallies", (a.n_union, a.n_rows, a.n_hist) == (ra.n_union, ra.n_rows, ra.n_hist)),
                  ("lengths", list(map(int, a.kmer_lengths)) == list(map(int, ra.kmer_lengths))),
                  ("matrix rows", a.n_rows == ra.n_rows and (a.keys[o] == ra.keys).all() and (a.counts[o] == ra.counts).all()),
                  ("mapped positions", b.n_mapped == rb.n_mapped),
                  ("window table", b.coords == rb.coords and (b.window_counts == rb.window_counts).all()),
                  ("calls", (b.sig == rb.sig).all() and (b.argmin == rb.argmin).all()
                   and bool(np.allclose(b.pvals, rb.pvals, rtol=1e-6, atol=1e-300)))]
        bad = [n_ for n_, v in checks if not v]
        ok, why = not bad, ", ".join(bad)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda")
    dist.broadcast(flag, 0)
    if not int(flag.item()):
        raise SystemExit("DIST SELFCHECK FAILED on rank 0: %s differ between the %d-rank pass and the single-process pass"
                         % (why or "results", dist.get_world_size()))
    return {"config": "small", "ranks": dist.get_world_size(), "pieces": [len(p) for p in runner.pieces], "ok": True,
            "compared": ["n_union/n_rows/n_hist", "lengths", "matrix rows", "mapped positions", "window table", "p-values/calls"]}


def jellyfish_leg(seqs, labs, k, lower, cores, dumps):
    """If the `jellyfish` binary is on PATH: the reference's exact count / dump commands (Jellyfish.py:697-699) on the
    CPU sample, timed, and its dump -- sorted -- compared with the oracle's (which the HIP path has just been verified
    against): the one route by which the count stage is pinned to jellyfish itself rather than to its definition.
    jellyfish 2.2.10 is not in this image (SubPhaser.yaml:66): the leg then reports "absent"."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("jellyfish")
    if not exe:
        return "absent"
    tmp = tempfile.mkdtemp(prefix="sp_jf_")
    try:
        t_all, same, n_lines = 0.0, True, 0
        for lab, s, (ok, oc) in zip(labs, seqs, dumps):
            fa = os.path.join(tmp, lab + ".fasta")
            with open(fa, "wb") as f:
                f.write(b">" + lab.encode() + b"\n")
                f.write(np.asarray(s, np.uint8).tobytes())
                f.write(b"\n")
            pre = "%s_%d" % (fa, k)
            cmd = ('cat {fa} | jellyfish count -t {t} -m {k} -s 100000000  --canonical /dev/stdin -o "{pre}.jf" && '
                   'jellyfish histo -h 100000 -t {t} -o {pre}.histo {pre}.jf && '
                   'jellyfish dump -c -o "{pre}.fa" "{pre}.jf" -L {L}').format(fa=fa, t=cores, k=k, pre=pre, L=lower)
            t0 = time.perf_counter()
            subprocess.check_call(cmd, shell=True)
            t_all += time.perf_counter() - t0
            code = {65: 0, 67: 1, 71: 2, 84: 3}
            got = {}
            with open(pre + ".fa") as f:
                for line in f:
                    km, c = line.split()
                    v = 0
                    for ch in km.encode():
                        v = (v << 2) | code[ch]
                    got[v] = int(c)
            n_lines += len(got)
            same = same and len(got) == len(ok) and all(got.get(int(a)) == int(b) for a, b in zip(ok.tolist(), oc.tolist()))
            os.remove(pre + ".jf")
        bases = sum(len(x) for x in seqs)
        return {"version": subprocess.check_output([exe, "--version"]).decode().strip(), "count_dump_s": round(t_all, 2),
                "Gbases_per_s": round(bases / t_all / 1e9, 5), "threads": cores, "dump_kmers": n_lines,
                "dumps_equal_oracle_and_hip": bool(same)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
