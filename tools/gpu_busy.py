#!/usr/bin/env python3
"""A second context that keeps the GPU busy (dev / test tool): counts a small synthetic genome over and over, cycling k and
the number of chains in flight, until its standard input reaches end of file -- i.e. until the process that started it closes
the pipe or dies.  Other chains competing for the CUs are what opened the window of the round-5 `s3_part1` race; the
stream-mode fuzz and the stress tests run with one of these next to them.

usage: gpu_busy.py [scale=0.01] [ks=19,15,17,22]        (started by `busy_neighbour()` below, not by hand)"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _main():
    from subphaser_amd import _native
    from subphaser_amd.synth import SynthGenome
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
    ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "19,15,17,22").split(",")]
    stop = threading.Event()

    def watch():
        try:
            while sys.stdin.buffer.read(4096):
                pass
        except Exception:
            pass
        stop.set()

    threading.Thread(target=watch, daemon=True).start()
    gen = SynthGenome("wheat", scale)
    ctx = _native.Context(0)
    ctx.genome_reset(len(gen.chroms))
    for i, c in enumerate(gen.chroms):
        p = ctx.dev_alloc(c["length"])
        ctx.synth_chrom(p, c["length"], gen.seed, c["set_id"], c["sg_id"], gen.S, c["chrom_id"], c["exchange"])
        ctx.genome_add_device(i, p, c["length"])
        ctx.dev_free(p)
    for name in ("SP_LANES_SPARSE", "SP_LANES_DENSE", "SP_LANES"):
        os.environ[name] = "3"
    print("busy: ready", flush=True)
    it = 0
    while not stop.is_set():
        ctx.count(ks[it % len(ks)], 3, 0)
        it += 1
    ctx.close()


class busy_neighbour:
    """context manager: `with busy_neighbour(): ...` runs the block with a gpu_busy.py process on the same GPU"""

    def __init__(self, scale=0.01, ks="19,15,17,22"):
        self.args = [sys.executable, os.path.abspath(__file__), str(scale), ks]
        self.proc = None

    def __enter__(self):
        import subprocess
        self.proc = subprocess.Popen(self.args, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        line = self.proc.stdout.readline()          # "busy: ready" -- its genome is resident and the loop has started
        if b"ready" not in line:
            self.__exit__(None, None, None)
            raise RuntimeError("gpu_busy.py did not start")
        return self

    def __exit__(self, *exc):
        p, self.proc = self.proc, None
        if p is None:
            return False
        try:
            p.stdin.close()                         # end of file: the loop ends after the count in flight
            p.wait(timeout=60)
        except Exception:
            p.kill()                                # (this exact child, nothing else)
            p.wait()
        return False


if __name__ == "__main__":
    _main()
