"""`subphaser` command line for modules 1-2: ingest -> count -> matrix/filter -> cluster -> bin map ->
window enrichment -> custom features, on the MI355X hot path.

What is kept from the reference is its CONTRACT: the flag surface (subphaser/__main__.py:29-248), the names
and formats of the files under `-o` / `-tmpdir` and the checkpoint files, so that
modules 3-4 of the reference (LTR, circos) and existing command lines keep working on these outputs.
How the run is organised is this build's own: a `Layout` that owns every path, five stages with explicit
inputs and outputs, and a second half that never leaves the device between the bin map, the window stack
and the Fisher tests -- `.subgenome.bin.count` is written FROM the device arrays as an output, it is not
parsed back (the reference re-reads it with Circos.stack_matrix before it can enrich).
"""
import argparse
import io
import os
import pickle
import shutil
import time
import sys
from collections import Counter, OrderedDict

import numpy as np

from . import REFERENCE_VERSION, __version__
from . import circos, seqs, stats
from .cluster import TEST_METHODS, Cluster
from .config import SGConfig, check_duplicates, parse_idmap
from .jellyfish import JellyfishDumps, plot_histogram, run_jellyfish_dumps
from .runtime import get_context, logger

NCPU = len(os.sched_getaffinity(0))
BIN_SIZE, CHUNK_SIZE, FEATURE_BIN = 10000, 10_000_000, 10_000_000   # Seqs.map_kmer3 defaults / __main__.py:509

# ---------------------------------------------------------------------------------------------------- CLI
# (group title, group description, [(flags, argparse keywords)]) -- the reference's options for modules 1-2;
# options of modules 3-4 are accepted (and ignored) so that one command line serves both programs.
_STR = dict(type=str, metavar="STR", default=None)
_FLAG = dict(action="store_true", default=False)
CLI = [
    ("Input", "Input genome and config files", [
        (("-i", "-genomes"), dict(dest="genomes", nargs="+", metavar="GENOME", required=True)),
        (("-c", "-sg_cfgs"), dict(dest="sg_cfgs", nargs="+", metavar="CFGFILE", required=True)),
        (("-labels",), dict(nargs="+", type=str, metavar="LABEL")),
        (("-no_label",), _FLAG),
        (("-target",), dict(type=str, metavar="FILE", default=None)),
        (("-sg_assigned",), dict(type=str, metavar="FILE", default=None)),
        (("-sep",), dict(type=str, metavar="STR", default="|")),
        (("-custom_features",), dict(nargs="+", metavar="FASTA|BED", default=None,
                                     help="feature sets to test for subgenome enrichment: FASTA files of feature sequences "
                                          "(ids chrom:start-end, as the reference takes them) and / or BED files of "
                                          "intervals on the target chromosomes (reduced over the resident genome: no "
                                          "sequence is uploaded)")),
    ]),
    ("Output", None, [
        (("-pre", "-prefix"), dict(dest="prefix", metavar="STR", default=None)),
        (("-o", "-outdir"), dict(dest="outdir", metavar="DIR", default="phase-results")),
        (("-tmpdir",), dict(type=str, metavar="DIR", default="tmp")),
        (("-colors",), dict(dest="colors", metavar="HEX,HEX[,...]", default=None)),
    ]),
    ("Kmer", "Options to count and filter kmers", [
        (("-k",), dict(type=int, metavar="INT", default=15)),
        (("-f", "-min_fold"), dict(dest="min_fold", type=float, metavar="FLOAT", default=2)),
        (("-q", "-min_freq"), dict(dest="min_freq", type=int, metavar="INT", default=200)),
        (("-baseline",), dict(type=int, default=1)),
        (("-ratio",), dict(type=float, default=1)),
        (("-lower_count",), dict(type=int, metavar="INT", default=3)),
        (("-min_prop",), dict(type=float, metavar="FLOAT", default=None)),
        (("-max_freq",), dict(type=int, metavar="INT", default=1e9)),
        (("-max_prop",), dict(type=float, metavar="FLOAT", default=None)),
        (("-low_mem",), dict(action="store_true", default=None)),
        (("-by_count",), dict(help="parsed, never used: as in the reference (__main__.py:98 vs :422-426)", **_FLAG)),
        (("-re_filter",), _FLAG),
    ]),
    ("Cluster", "Options for clustering to phase", [
        (("-nsg",), dict(type=int, metavar="INT", default=None)),
        (("-replicates",), dict(type=int, metavar="INT", default=1000)),
        (("-jackknife",), dict(type=float, metavar="FLOAT", default=50)),
        (("-max_pval",), dict(type=float, metavar="FLOAT", default=0.05)),
        (("-test_method",), dict(choices=list(TEST_METHODS), default="ttest_ind")),
        (("-figfmt",), dict(type=str, choices=["pdf", "png"], default="pdf")),
        (("-heatmap_colors",), dict(nargs="+", metavar="COLOR", default=("green", "black", "red"))),
        (("-heatmap_options",), dict(metavar="STR", default="")),
        (("-just_core",), _FLAG),
    ]),
    ("LTR / Circos", "accepted for command-line compatibility; modules 3-4 stay with the reference",
     [((o,), _STR) for o in ("-ltr_finder_options", "-ltr_harvest_options", "-tesorter_options", "-trimal_options",
                             "-tree_method", "-tree_options", "-ggtree_options", "-aligner", "-aligner_options",
                             "-chr_ordered")]
     + [((o,), dict(nargs="+", default=None)) for o in ("-ltr_detectors", "-ltr_domains", "-alt_cfgs")]
     + [((o,), _FLAG) for o in ("-disable_ltr", "-all_ltr", "-intact_ltr", "-exclude_exchanges", "-non_specific",
                                "-disable_ltrtree", "-disable_circos", "-disable_blocks")]
     + [(("-mu",), dict(type=float, default=13e-9)), (("-subsample",), dict(type=int, default=1000)),
        (("-window_size",), dict(type=int, metavar="INT", default=1000000)),
        (("-min_block",), dict(type=int, default=100000))]),
    ("Other options", None, [
        (("-p", "-ncpu"), dict(dest="ncpu", type=int, metavar="INT", default=NCPU)),
        (("-max_memory",), dict(type=str, metavar="MEM", default=None)),
        (("-cleanup",), _FLAG),
        (("-overwrite",), _FLAG),
        (("-engine",), dict(type=int, default=0, help="k-mer counting engine (0 auto, 1 atomic table, 2 LDS radix, 3 LDS radix into lists: small genomes)")),
        (("-write_dumps",), dict(help="also write jellyfish-style text dumps {chrom}_{k}.fa", **_FLAG)),
        (("-bootstrap_seed",), dict(type=int, default=None, help="seed of k-means and of the bootstrap resampling")),
    ]),
]


def makeArgparse(argv=None):
    parser = argparse.ArgumentParser(
        prog="subphaser", formatter_class=argparse.RawDescriptionHelpFormatter,
        description="Phase subgenomes of an allopolyploid or hybrid based on repetitive kmers "
                    "(MI355X-native k-mer counting / enrichment; modules 1-2 of SubPhaser).\n"
                    "Limits of the dense path (k <= 15): at most 560 chromosomes in the filter, 8 subgenome columns "
                    "per config line unless -baseline is 1 or -1, 32 subgenomes in the enrichment.")
    for title, desc, options in CLI:
        group = parser.add_argument_group(title, desc)
        for flags, kw in options:
            group.add_argument(*flags, **kw)
    parser.add_argument("-v", "-version", action="version",
                        version="subphaser_amd {} (interface of SubPhaser {})".format(__version__, REFERENCE_VERSION))
    args = parser.parse_args(argv)
    if args.prefix is not None:          # the prefix goes in front of directory names AND file names (__main__.py:242-245)
        args.prefix = args.prefix.replace("/", "_")
        args.outdir, args.tmpdir = args.prefix + args.outdir, args.prefix + args.tmpdir
    return args


# ------------------------------------------------------------------------------------------- paths, checkpoints
class Layout:
    """Every path of a run.  `out("x")` = <outdir>/<prefix>k15_q200_f2.x, `ckp(name)` = <tmpdir>/<prefix>name.ok"""

    def __init__(self, outdir, tmpdir, prefix, k, min_freq, min_fold):
        self.outdir, self.tmpdir = os.path.realpath(outdir) + "/", os.path.realpath(tmpdir) + "/"
        os.makedirs(self.outdir, exist_ok=True)
        os.makedirs(self.tmpdir, exist_ok=True)
        self.out_prefix = self.outdir + (prefix or "")
        self.tmp_prefix = self.tmpdir + (prefix or "")
        self.basename = "k{}_q{}_f{}".format(k, min_freq, min_fold)

    def out(self, ext):
        return "{}{}.{}".format(self.out_prefix, self.basename, ext)

    def ckp(self, file_or_name):
        return "{}{}.ok".format(self.tmp_prefix, os.path.basename(file_or_name))

    @property
    def chromdir(self):
        return self.tmp_prefix + "chromosomes/"


def mk_ckp(ckpfile, *data):
    """A checkpoint is the objects pickled one after the other (the reference's small_tools.mk_ckp)."""
    with open(ckpfile, "wb") as f:
        for obj in data:
            pickle.dump(obj, f)
    logger.info("New check point file: `{}`".format(ckpfile))


def check_ckp(ckpfile):
    """False: no checkpoint.  True: an empty one.  Otherwise the list of objects it holds."""
    if not os.path.exists(ckpfile):
        return False
    logger.info("Check point file: `{}` exists; skip this step".format(ckpfile))
    objs = []
    with open(ckpfile, "rb") as f:
        while True:
            try:
                objs.append(pickle.load(f))
            except EOFError:
                break
    return objs or True


# ------------------------------------------------------------------------------------------------- the run
def _bg_min_rows():
    """outputs with at least this many rows are written on a background thread (SP_BG_MIN_ROWS: tests force it)"""
    return int(os.environ.get("SP_BG_MIN_ROWS", "200000"))


class Pipeline:
    def __init__(self, genomes, sg_cfgs, labels=None, **opts):
        self.__dict__.update(opts)
        check_duplicates(genomes)
        check_duplicates(labels)
        self.genomes, self.sg_cfgs = genomes, sg_cfgs
        n = len(genomes)
        if self.no_label or (labels is None and n == 1):
            self.labels = [""] * n
        elif labels is None:
            self.labels = ["{}-".format(i) for i in range(1, n + 1)]
        else:
            self.labels = labels
        # one prefix per config file when the counts match, else none (__main__.py:274-281)
        cfg_prefix = self.labels if len(self.labels) == len(sg_cfgs) else [None] * len(sg_cfgs)
        cfgs = [SGConfig(f, prefix=p, sep=self.sep) for f, p in zip(sg_cfgs, cfg_prefix)]
        self.sgs = [line for cfg in cfgs for line in cfg.sgs]
        self.chrs = [c for cfg in cfgs for c in cfg.chrs]
        if not self.nsg or self.nsg < 2:
            self.nsg = sum(cfg.nsg for cfg in cfgs)

    # ---- stage 1: chromosomes ---------------------------------------------------------------------------
    def stage_ingest(self, lay):
        """Target chromosomes as per-chromosome FASTA (+ in-memory records); checkpoint `split.ok` holds
        (chromfiles, labels, d_targets, d_size) like the reference's."""
        logger.info("chromosomes named in the config: {}".format(self.chrs))
        logger.info("per-chromosome FASTA under `{}`".format(lay.chromdir))
        ckp = lay.ckp("split")
        saved = None if self.overwrite else check_ckp(ckp)
        reuse = isinstance(saved, list) and len(saved) == 4
        if reuse:
            chromfiles, labels, d_targets, d_size = saved
            if set(d_targets) != set(self.chrs):        # another target set: split again, and filter again
                reuse, self.re_filter = False, True
            elif not all(os.access(f, os.R_OK) for f in chromfiles):
                reuse = False
        if not reuse:
            # The per-chromosome copies are an OUTPUT for the reference's later modules; counting uses the records
            # in memory.  Their writer threads keep going while the GPU counts; `split.ok` is recorded once they
            # are done (_finish_background).
            os.makedirs(lay.chromdir, exist_ok=True)
            chromfiles, labels, d_targets, d_size, waiting = seqs.split_genomes(
                self.genomes, self.labels, self.chrs, lay.chromdir, d_targets=parse_idmap(self.target), sep=self.sep,
                defer=True)
            self._background.append(("chromosome files", waiting.wait,
                                     lambda: mk_ckp(ckp, chromfiles, labels, d_targets, d_size)))
        self.d_targets = dict(d_targets)      # ids as they are in the input -> labels (BED files may use either)
        # config order, not file order
        where = dict(zip(labels, chromfiles))
        labels = [lab for lab in d_targets.values() if lab in where]
        chromfiles = [where[lab] for lab in labels]
        if not chromfiles:
            raise ValueError("0 chromosome remained after filtering. Please check the inputs.")
        logger.info("{} chromosomes, in config order: {}".format(len(labels), labels))
        logger.info("Genome size: {:,} bp".format(sum(d_size.values())))
        rename = d_targets.get
        self.sgs = [[[rename(c, c) for c in unit] for unit in line] for line in self.sgs]
        logger.info("homoeologous sets after renaming: {}".format(self.sgs))
        assigned = {}
        if self.sg_assigned:
            for line in open(self.sg_assigned):
                if line.strip() and not line.startswith("#"):
                    c, sg = line.split()[:2]
                    assigned[rename(c, c)] = sg
        return chromfiles, labels, d_size, assigned

    # ---- stage 2: count + matrix + differential filter --------------------------------------------------
    def stage_count_filter(self, lay, chromfiles, labels):
        logger.info("###Step: Kmer Count")
        logger.info("Counting kmer on the GPU (replaces jellyfish)")
        dumpfiles = run_jellyfish_dumps(chromfiles, k=self.k, ncpu=self.ncpu, lower_count=self.lower_count,
                                        overwrite=self.overwrite, engine=self.engine, write_dumps=self.write_dumps)
        logger.info("chromosome x k-mer matrix: the count tables in HBM")
        dumps = JellyfishDumps(dumpfiles, labels, ncpu=self.ncpu)
        d_mat = dumps.to_matrix()
        logger.info("differential k-mer filter (K3)")
        histfig = lay.out("kmer_freq." + self.figfmt)
        d_mat2 = dumps.filter(d_mat, dumps.lengths, self.sgs, outfig=histfig, min_fold=self.min_fold,
                              baseline=self.baseline, min_freq=self.min_freq, max_freq=self.max_freq,
                              min_prop=self.min_prop, max_prop=self.max_prop, ratio=self.ratio)
        logger.info("{} kmers in total".format(len(d_mat)))
        if len(d_mat2) == 0:
            raise ValueError("0 kmer remained after filtering. Please reset the filter options.")
        # The matrix in memory is what every later stage uses, so the file always describes THIS run's filter:
        # it is rewritten whenever it is recomputed (an old file next to new calls would be inconsistent).
        # It is 0.5 GB of text at wheat scale: a writer thread (the library formats the rows on its own threads) works
        # while the clustering and mapping stages run; the checkpoint follows the writer's end.
        matfile = lay.out("kmer.mat")
        t0 = time.perf_counter()
        # the k-mer tests of the next stage read the rows on the device: copy them there now, while nothing else
        # competes for the host memory bus
        if getattr(d_mat2, "counts", None) is not None and getattr(d_mat2, "ctx", None) is not None \
                and hasattr(d_mat2.ctx, "stage_rows"):
            d_mat2.counts_dev = d_mat2.ctx.stage_rows(d_mat2.counts)
        self._write_matrix_in_background(dumps, d_mat2, matfile, histfig, lay)
        logger.info("`{}` + histogram figure: writer started in {:.2f} s".format(os.path.basename(matfile),
                                                                             time.perf_counter() - t0))
        return d_mat2

    def _write_matrix_in_background(self, dumps, d_mat2, matfile, histfig, lay):
        tot = dumps.hist_tot()          # device -> host here: the writer thread makes no GPU calls

        def work(fout):
            dumps.write_matrix(d_mat2, fout)

        def done():
            # The figure is drawn on the MAIN thread, after the writers: importing matplotlib's extension modules
            # (dlopen under the GIL) on a writer thread while scikit-learn's threadpoolctl walks the loaded
            # libraries (dl_iterate_phdr calling back into Python) on a bootstrap thread is a lock-order deadlock
            # -- one wheat-scale run in four hung there.
            mk_ckp(lay.ckp(matfile))
            try:
                plot_histogram(tot, histfig)
            except Exception as e:     # the figure is optional
                logger.warning("histogram not plotted: {}".format(e))
        self._write_in_background(matfile, work, len(d_mat2) >= _bg_min_rows(), done=done)

    def _write_in_background(self, path, write, big, done=None):
        """write(fout) into `path`: on a thread of this process when the output is large (the library's text writers
        release the GIL and use their own threads; no child processes: the GPU process stays single), inline otherwise;
        `done` (e.g. the checkpoint) runs on the main thread once the file is complete.  `write` must not import
        anything (see _write_matrix_in_background)."""
        import threading
        done = done or (lambda: None)
        failure = []

        def work():
            try:
                with open(path, "w") as fout:
                    write(fout)
            except BaseException as e:      # re-raised by wait()
                failure.append(e)

        if not big:
            work()
            if failure:
                raise failure[0]
            done()
            return
        th = threading.Thread(target=work, name="writer:" + os.path.basename(path), daemon=False)
        th.start()

        def wait():
            th.join()
            if failure:
                raise failure[0]
        self._background.append((path, wait, done))

    def _finish_background(self):
        """Wait for the writers started along the way, then record their checkpoints.  A checkpoint describes ITS
        file only: run() also calls this after a later stage raised, and a `.kmer.mat` that was written completely
        keeps its checkpoint then (the file is valid; a rerun recomputes whatever depends on the failed stage)."""
        pending, self._background = self._background, []
        err = None
        for name, wait, done in pending:
            try:
                wait()
                done()
            except Exception as e:      # keep waiting for the others; report the first failure
                logger.error("background writer of {} failed: {}".format(name, e))
                err = err or e
        if err is not None:
            raise err

    # ---- stage 3: subgenome assignment + subgenome-specific k-mers --------------------------------------
    def stage_cluster(self, lay, d_mat2, assigned):
        logger.info("###Step: Cluster")
        cl = Cluster(d_mat2, n_clusters=self.nsg, sg_prefix="SG", sg_assigned=assigned, bootstrap=True,
                     replicates=self.replicates, jackknife=self.jackknife, seed=self.bootstrap_seed)
        logger.info("Subgenome assignments: {}".format(dict(cl.d_sg)))
        with open(lay.out("chrom-subgenome.tsv"), "w") as fout:
            cl.output_subgenomes(fout)
        sg_kmers = lay.out("sig.kmer-subgenome.tsv")
        logger.info("subgenome-specific k-mers -> `{}`".format(sg_kmers))
        t0 = time.perf_counter()
        kmer_labels, write = cl.output_kmers(None, max_pval=self.max_pval, test_method=self.test_method, defer=True)
        t1 = time.perf_counter()
        self._write_in_background(sg_kmers, write, len(kmer_labels.keys) >= _bg_min_rows())
        logger.info("k-mer tests {:.2f} s, text writer started in {:.2f} s".format(t1 - t0, time.perf_counter() - t1))
        per_sg = np.bincount(kmer_labels.sg_idx, minlength=len(kmer_labels.sg_names))
        logger.info("{} significant subgenome-specific kmers".format(len(kmer_labels.keys)))
        for sg, n in zip(kmer_labels.sg_names, per_sg.tolist()):
            if n:
                logger.info("\t{} {}-specific kmers".format(n, sg))
        return cl, kmer_labels

    # ---- stage 4: bin map -> window stack -> enrichment, on the device -----------------------------------
    def stage_windows(self, lay, chromfiles, labels, d_size, cl, kmer_labels):
        ctx = get_context()
        S, names = len(cl.sg_names), cl.sg_names
        lengths = [d_size[lab] for lab in labels]
        sg_map = lay.out("subgenome.bin.count")
        logger.info("10-kb bin counts -> `{}`".format(sg_map))
        ctx.labels_set(kmer_labels.keys, kmer_labels.sg_idx, S)
        slots, n_mapped = ctx.map_bins_all(BIN_SIZE, CHUNK_SIZE)
        with open(sg_map, "w") as fout:     # an OUTPUT of the device arrays; nothing below reads it
            fout.write("\t".join(["#chrom", "start", "end"] + names) + "\n")
            for lab, n, sl in zip(labels, lengths, slots):
                seqs._write_lines(fout, lab, *seqs.bin_lines(lab, n, sl, BIN_SIZE, CHUNK_SIZE, self.k))
        mk_ckp(lay.ckp(sg_map))
        n_chunks = [max(1, -(-n // CHUNK_SIZE)) for n in lengths]
        hit = sum(c for c, m in zip(n_chunks, n_mapped.tolist()) if m)
        logger.info("Processed {} sequences".format(sum(n_chunks)))
        logger.info("{} ({:.2%}) sequences contain subgenome-specific kmers".format(hit, hit / max(1, sum(n_chunks))))
        if len(kmer_labels.keys):
            logger.info("{:.2%} of {} subgenome-specific kmers are mapped".format(
                ctx.labels_hit() / len(kmer_labels.keys), len(kmer_labels.keys)))
        logger.info("window enrichment, {}-bp windows (K6)".format(self.window_size))
        ws = int(self.window_size)
        win, woff, pvals, argmin, sig, ratios = ctx.stack_enrich(BIN_SIZE, CHUNK_SIZE, ws, lengths, self.max_pval, 0.5)
        nz = np.flatnonzero(win.any(axis=1))       # only windows that received a bin line exist (Circos.py:734-742)
        chrom = np.searchsorted(woff, nz, side="right") - 1
        rows = [(labels[c], int(w) * ws, int(w) * ws + ws) for c, w in zip(chrom.tolist(), (nz - woff[chrom]).tolist())]
        bin_enrich = lay.out("bin.enrich")
        with open(bin_enrich, "w") as f1, open(lay.out("bin.group"), "w") as f2:
            self.sg_lines = stats.enrich_bin(f1, f2, cl.d_sg, win[nz].astype(np.int64), colnames=names, rownames=rows,
                                             max_pval=self.max_pval,
                                             results=(pvals[nz], argmin[nz], sig[nz].astype(bool), ratios[nz]))
        logger.info("wrote {}".format(bin_enrich))

    # ---- stage 5: custom feature sets ----------------------------------------------------------------------
    def stage_features(self, lay, cl, kmer_labels):
        feat_map = lay.out("custom.bin.count")
        logger.info("feature sets: {}".format(self.custom_features))
        lines = []       # the written lines as arrays: `custom.bin.count` is an output, nothing below parses it back
        ivals = []       # BED feature sets: (IntervalRows, counts) per file, one row per feature
        beds = [f for f in self.custom_features if seqs.is_bed(f)]
        fastas = [f for f in self.custom_features if f not in beds]
        with open(feat_map, "w") as fout:
            if fastas or not beds:
                seqs.map_kmer3(fastas, kmer_labels, fout=fout, k=self.k, bin_size=FEATURE_BIN,
                               sg_names=cl.sg_names, chunk=False, log=False, collect=lines)
            if beds:
                # intervals: a reduction over the genome the GPU already holds; ids are synthesised as chrom:start-end
                # (what enrich_ltr's id rule expects); BED names may be the ids before or after renaming
                if fastas:
                    sub = io.StringIO()
                    seqs.map_intervals(beds, kmer_labels, {lab: i for i, lab in enumerate(self.labels)}, fout=sub, k=self.k,
                                       bin_size=FEATURE_BIN, sg_names=cl.sg_names, collect=ivals,
                                       aliases=dict(getattr(self, "d_targets", None) or {}))
                    fout.write(sub.getvalue().split("\n", 1)[1])      # one header line per file
                else:
                    seqs.map_intervals(beds, kmer_labels, {lab: i for i, lab in enumerate(self.labels)}, fout=fout, k=self.k,
                                       bin_size=FEATURE_BIN, sg_names=cl.sg_names, collect=ivals,
                                       aliases=dict(getattr(self, "d_targets", None) or {}))
        logger.info("feature enrichment")
        S = len(cl.sg_names)
        ids, counts = [], []
        if lines:
            names, code = circos.factorize_first([x for part in lines for x in part[0]])
            ids, counts = circos.stack_arrays(names, code, np.concatenate([part[1] for part in lines]),
                                              np.concatenate([part[2] for part in lines], axis=0), window_size=100000000)
        feat_enrich = lay.out("custom.enrich")
        with open(feat_enrich, "w") as fout:
            if ivals and not ids:       # intervals only: rows stay arrays end to end (millions of features)
                rows, rcounts = seqs.IntervalRows.concat([p_[0] for p_ in ivals]).merged(
                    np.concatenate([p_[1] for p_ in ivals], axis=0))
                sg_idx, _ = stats.enrich_ltr(fout, cl.d_sg, rcounts,
                                             colnames=cl.sg_names, rownames=rows, max_pval=self.max_pval, as_arrays=True)
                enriched = {i: cl.sg_names[j] for i, j in enumerate(sg_idx.tolist()) if j >= 0}
            else:
                for rows, cc in ivals:
                    ids = list(ids) + [(x, 0, 100000000) for x in rows.ids()]
                    counts = list(counts) + cc.tolist()
                if ivals:
                    # lines that share an id are ONE row whatever file they came from (Circos.stack_matrix,
                    # Circos.py:709-742, sums them): a BED interval listed twice, or listed next to a FASTA record of
                    # the same id, must give the row the intervals-only branch above gives (advisor r04)
                    where, m_ids, m_counts = {}, [], []
                    for rid, row in zip(ids, counts):
                        j = where.get(rid)
                        if j is None:
                            where[rid] = len(m_ids)
                            m_ids.append(rid)
                            m_counts.append(list(row))
                        else:
                            m_counts[j] = [x + y for x, y in zip(m_counts[j], row)]
                    ids, counts = m_ids, m_counts
                enriched, _ = stats.enrich_ltr(fout, cl.d_sg, np.asarray(counts, np.int64).reshape(len(ids), S),
                                               colnames=cl.sg_names, rownames=ids, max_pval=self.max_pval)
        logger.info("wrote {}".format(feat_enrich))
        logger.info("{} significant subgenome-specific features".format(len(enriched)))
        for sg, n in sorted(Counter(enriched.values()).items()):
            logger.info("\t{} {}-specific features".format(n, sg))

    def run(self):
        self._background = []
        try:
            self._run()
        except BaseException:
            try:
                self._finish_background()      # do not leave writers behind; the stage's error is the one to report
            except Exception:
                pass
            raise
        self._finish_background()
        if self.cleanup:
            logger.info("Cleaning {}".format(self._lay.tmpdir))
            shutil.rmtree(self._lay.tmpdir, ignore_errors=True)
        logger.info("Pipeline completed" + (" early" if self.just_core else ""))

    def _run(self):
        lay = self._lay = Layout(self.outdir, self.tmpdir, self.prefix, self.k, self.min_freq, self.min_fold)
        chromfiles, labels, d_size, assigned = self.stage_ingest(lay)
        self.chromfiles, self.labels, self.d_size = chromfiles, labels, d_size
        d_mat2 = self.stage_count_filter(lay, chromfiles, labels)
        cl, kmer_labels = self.stage_cluster(lay, d_mat2, assigned)
        self.d_sg, self.sg_names = cl.d_sg, cl.sg_names
        if not self.just_core:
            self.stage_windows(lay, chromfiles, labels, d_size, cl, kmer_labels)
            if self.custom_features is not None:
                self._finish_background()      # `.kmer.mat` / sig-k-mer writers done: their threads would compete with the feature writers' 
                self.stage_features(lay, cl, kmer_labels)
            if not (self.disable_ltr and self.disable_circos):
                logger.info("Modules 3-4 (LTR, circos) are not part of this build; run the reference on the "
                            "outputs above, or pass -disable_ltr -disable_circos to silence this note")


def main(argv=None):
    args = makeArgparse(argv)
    if os.environ.get("SP_STACKS_AFTER"):      # debugging aid: dump every thread's Python stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SP_STACKS_AFTER"]), exit=True)
    logger.info("Command: {}".format(" ".join(sys.argv)))
    logger.info("Version: subphaser_amd {}".format(__version__))
    logger.info("Arguments: {}".format(args.__dict__))
    Pipeline(**args.__dict__).run()


def cli(argv=None):
    """Process entry point (`python -m subphaser_amd`, the `subphaser` script): main(), then leave without the interpreter's
    teardown.  When run() returns every output file is closed and every background writer joined; what is left is returning
    ~40 GB of device and page-locked memory buffer by buffer and unwinding numpy / matplotlib -- 1.1 of the 6.1 s of a
    wheat-sized run (profiles/r06_e2e_cli_wheat.log) that the operating system does in one piece when the process ends.
    `SP_SLOW_EXIT=1` keeps the ordinary exit (and, with `SP_EXIT_TRACE=1`, stamps its steps on stderr)."""
    main(argv)
    trace = bool(os.environ.get("SP_EXIT_TRACE"))
    if trace:
        sys.stderr.write("[exit] run() returned\n")
        sys.stderr.flush()
    if os.environ.get("SP_SLOW_EXIT") == "1":
        if trace:
            from .runtime import close_context
            close_context()
            sys.stderr.write("[exit] context closed\n")
            sys.stderr.flush()
        return
    import logging
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


if __name__ == "__main__":
    cli()
