#!/usr/bin/env python3
"""End-to-end wall clock of the `subphaser` CLI (modules 1-2) on a synthetic genome written to disk as
FASTA: FASTA in the page cache -> every output file on disk.  Reported separately from bench.py's
device-resident throughput.  usage: e2e_cli.py [config=ara] [workdir=/tmp/sp_e2e]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from subphaser_amd import _native
from subphaser_amd.seqs import write_fasta
from subphaser_amd.synth import SynthGenome

cfg = sys.argv[1] if len(sys.argv) > 1 else "ara"
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/sp_e2e"
os.makedirs(work, exist_ok=True)
gen = SynthGenome(cfg)
ctx = _native.Context(0)
fa = os.path.join(work, "genome.fa")
t0 = time.perf_counter()
reuse = os.path.exists(fa) and os.path.getsize(fa) > gen.total_bases      # a previous call left it there
with open(os.devnull if reuse else fa, "wb") as out:
    for c in ([] if reuse else gen.chroms):
        p = ctx.dev_alloc(c["length"])
        ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
        seq = ctx.dev_to_host(p, c["length"]).tobytes()
        ctx.dev_free(p)
        tmp = os.path.join(work, "one.fa")
        write_fasta(tmp, c["label"], seq)
        out.write(open(tmp, "rb").read())
ctx.close()
with open(os.path.join(work, "sg.config"), "w") as f:
    for sg in gen.sgs:
        f.write("\t".join(",".join(u) for u in sg) + "\n")
with open(os.path.join(work, "assigned.tsv"), "w") as f:
    for k, v in gen.sg_assigned.items():
        f.write("%s\t%s\n" % (k, v))
print("synthetic FASTA: %.1f MB written in %.1f s" % (os.path.getsize(fa) / 1e6, time.perf_counter() - t0))
prof = ["-m", "cProfile", "-o", os.path.join(work, "cli.prof")] if os.environ.get("SP_E2E_PROFILE") else []
cmd = [sys.executable] + prof + ["-m", "subphaser_amd", "-i", fa, "-c", os.path.join(work, "sg.config"), "-sg_assigned",
       os.path.join(work, "assigned.tsv"), "-o", os.path.join(work, "out"), "-tmpdir", os.path.join(work, "tmp"),
       "-disable_ltr", "-disable_circos", "-overwrite", "-figfmt", "png"]
t0 = time.perf_counter()
# the CLI's log lines are stamped as they ARRIVE (its own timestamps have one-second resolution): the per-phase wall table
proc = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, bufsize=1,
                        env=dict(os.environ, SP_STACKS_AFTER=os.environ.get("SP_STACKS_AFTER", "240")))
stamped = []
for line in proc.stderr:
    stamped.append((time.perf_counter() - t0, line.rstrip("\n")))
proc.wait()
dt = time.perf_counter() - t0


class _R:
    returncode = proc.returncode
    stderr = "\n".join(l for _, l in stamped)


r = _R()
open(os.path.join(work, "cli.stderr"), "w").write(r.stderr)
keep = ("[exit]", "per-chromosome FASTA", "chromosomes, in config order", "Genome size", "###Step", "Counting", "matrix", "filter (K3)",
        "After filtering", "kmers in total", "bootstrap", "Bootstrap", "->", "significant subgenome", "Processed",
        "enrichment", "wrote", "writer", "Pipeline completed", "New check point")
prev = 0.0
print("   at s   +delta   (wall clock from process start; a line is printed when its phase BEGINS or an output is done)")
for t, line in stamped:
    if any(k in line for k in keep) and "Loading /" not in line:
        print("%7.2f  %+6.2f   %s" % (t, t - prev, line[:150]))
        prev = t
print("%7.2f  %+6.2f   (process exit: background writers joined, interpreter torn down)" % (dt, dt - prev))
if r.returncode:
    print(r.stderr[-3000:])
print("exit", r.returncode)
if prof:
    import pstats
    pstats.Stats(os.path.join(work, "cli.prof")).sort_stats("cumtime").print_stats(45)
print("END-TO-END %s: %.2f s wall for %.3f Gbases -> %.3f Gbases/s (FASTA parse + upload + all kernels + all output files)"
      % (cfg, dt, gen.total_bases / 1e9, gen.total_bases / dt / 1e9))
for f in sorted(os.listdir(os.path.join(work, "out"))):
    print("  %-50s %12d bytes" % (f, os.path.getsize(os.path.join(work, "out", f))))
