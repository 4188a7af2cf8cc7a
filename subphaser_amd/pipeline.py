"""`subphaser` command line + pipeline for modules 1-2 (count -> matrix/filter ->
cluster -> bin map -> window enrichment -> custom features).

Mirrors the flag surface and step order of the reference's Pipeline.run()
(subphaser/__main__.py:29-248 argparse, :250-544 run) for the hot path only.
Flags of the LTR / Circos groups (modules 3-4) are accepted so existing command
lines keep working, and are reported as skipped: those modules stay with the
reference and consume the files written here unchanged (`.kmer.mat`,
`.subgenome.bin.count`, `.bin.enrich`, `.bin.group`, checkpoints `*.ok`).
"""
import argparse
import os
import pickle
import shutil
import sys
from collections import Counter, OrderedDict

from . import REFERENCE_VERSION, __version__
from . import circos as Circos
from . import seqs as Seqs
from . import stats as Stats
from .cluster import Cluster
from .config import SGConfig, check_duplicates, parse_idmap
from .jellyfish import JellyfishDumps, plot_histogram, run_jellyfish_dumps
from .runtime import logger

NCPU = len(os.sched_getaffinity(0))


def makeArgparse(argv=None):
    p = argparse.ArgumentParser(
        prog="subphaser",
        formatter_class=argparse.RawDescriptionHelpFormatter,
        description="Phase subgenomes of an allopolyploid or hybrid based on repetitive kmers "
                    "(MI355X-native k-mer counting / enrichment; modules 1-2 of SubPhaser).")
    g = p.add_argument_group("Input", "Input genome and config files")
    g.add_argument("-i", "-genomes", dest="genomes", nargs="+", metavar="GENOME", required=True)
    g.add_argument("-c", "-sg_cfgs", dest="sg_cfgs", nargs="+", required=True, metavar="CFGFILE")
    g.add_argument("-labels", nargs="+", type=str, metavar="LABEL")
    g.add_argument("-no_label", action="store_true", default=False)
    g.add_argument("-target", default=None, type=str, metavar="FILE")
    g.add_argument("-sg_assigned", default=None, type=str, metavar="FILE")
    g.add_argument("-sep", default="|", type=str, metavar="STR")
    g.add_argument("-custom_features", nargs="+", metavar="FASTA", default=None)
    g = p.add_argument_group("Output")
    g.add_argument("-pre", "-prefix", default=None, dest="prefix", metavar="STR")
    g.add_argument("-o", "-outdir", default="phase-results", dest="outdir", metavar="DIR")
    g.add_argument("-tmpdir", default="tmp", type=str, metavar="DIR")
    g.add_argument("-colors", default=None, dest="colors", metavar="HEX,HEX[,...]")
    g = p.add_argument_group("Kmer", "Options to count and filter kmers")
    g.add_argument("-k", type=int, default=15, metavar="INT")
    g.add_argument("-f", "-min_fold", type=float, default=2, metavar="FLOAT", dest="min_fold")
    g.add_argument("-q", "-min_freq", type=int, default=200, metavar="INT", dest="min_freq")
    g.add_argument("-baseline", type=int, default=1)
    g.add_argument("-ratio", type=float, default=1)
    g.add_argument("-lower_count", type=int, default=3, metavar="INT")
    g.add_argument("-min_prop", type=float, default=None, metavar="FLOAT")
    g.add_argument("-max_freq", type=int, default=1e9, metavar="INT")
    g.add_argument("-max_prop", type=float, default=None, metavar="FLOAT")
    g.add_argument("-low_mem", action="store_true", default=None)
    g.add_argument("-by_count", action="store_true", default=False,
                   help="parsed but never forwarded to the filter, as in the reference (__main__.py:98 vs :422-426)")
    g.add_argument("-re_filter", action="store_true", default=False)
    g = p.add_argument_group("Cluster", "Options for clustering to phase")
    g.add_argument("-nsg", type=int, default=None, metavar="INT")
    g.add_argument("-replicates", type=int, default=1000, metavar="INT")
    g.add_argument("-jackknife", type=float, default=50, metavar="FLOAT")
    g.add_argument("-max_pval", type=float, default=0.05, metavar="FLOAT")
    g.add_argument("-test_method", default="ttest_ind", choices=["ttest_ind", "kruskal", "wilcoxon", "mannwhitneyu"])
    g.add_argument("-figfmt", default="pdf", type=str, choices=["pdf", "png"])
    g.add_argument("-heatmap_colors", nargs="+", default=("green", "black", "red"), metavar="COLOR")
    g.add_argument("-heatmap_options", metavar="STR", default="")
    g.add_argument("-just_core", action="store_true", default=False)
    g = p.add_argument_group("LTR / Circos", "accepted for command-line compatibility; modules 3-4 stay with the reference")
    g.add_argument("-disable_ltr", action="store_true", default=False)
    for opt in ("-ltr_finder_options", "-ltr_harvest_options", "-tesorter_options", "-trimal_options",
                "-tree_method", "-tree_options", "-ggtree_options", "-aligner", "-aligner_options", "-chr_ordered"):
        g.add_argument(opt, default=None, metavar="STR")
    for opt in ("-ltr_detectors", "-ltr_domains", "-alt_cfgs"):
        g.add_argument(opt, nargs="+", default=None)
    for opt in ("-all_ltr", "-intact_ltr", "-exclude_exchanges", "-non_specific", "-disable_ltrtree",
                "-disable_circos", "-disable_blocks"):
        g.add_argument(opt, action="store_true", default=False)
    g.add_argument("-mu", type=float, default=13e-9)
    g.add_argument("-subsample", type=int, default=1000)
    g.add_argument("-window_size", type=int, default=1000000, metavar="INT")
    g.add_argument("-min_block", type=int, default=100000)
    g = p.add_argument_group("Other options")
    g.add_argument("-p", "-ncpu", type=int, default=NCPU, metavar="INT", dest="ncpu")
    g.add_argument("-max_memory", type=str, default=None, metavar="MEM")
    g.add_argument("-cleanup", action="store_true", default=False)
    g.add_argument("-overwrite", action="store_true", default=False)
    g.add_argument("-engine", type=int, default=0, help="k-mer counting engine (0 auto, 1 atomic table, 2 LDS radix)")
    g.add_argument("-write_dumps", action="store_true", default=False,
                   help="also write jellyfish-style text dumps {chrom}_{k}.fa")
    g.add_argument("-v", "-version", action="version",
                   version="subphaser_amd {} (interface of SubPhaser {})".format(__version__, REFERENCE_VERSION))
    args = p.parse_args(argv)
    if args.prefix is not None:          # __main__.py:242-245
        args.prefix = args.prefix.replace("/", "_")
        args.outdir = args.prefix + args.outdir
        args.tmpdir = args.prefix + args.tmpdir
    return args


def mk_ckp(ckpfile, *data):
    """Checkpoint = sequentially pickled objects (small_tools.py:40-46)."""
    with open(ckpfile, "wb") as f:
        for d in data:
            pickle.dump(d, f)
    logger.info("New check point file: `{}`".format(ckpfile))


def check_ckp(ckpfile):
    """False if missing, True if empty, else the list of pickled objects (small_tools.py:49-70)."""
    if not os.path.exists(ckpfile):
        return False
    logger.info("Check point file: `{}` exists; skip this step".format(ckpfile))
    if os.path.getsize(ckpfile) == 0:
        return True
    data = []
    with open(ckpfile, "rb") as f:
        while True:
            try:
                data.append(pickle.load(f))
            except EOFError:
                break
    return data


class Pipeline:
    def __init__(self, genomes, sg_cfgs, labels=None, **kargs):
        self.genomes, self.sg_cfgs = genomes, sg_cfgs
        self.__dict__.update(**kargs)
        check_duplicates(genomes)
        check_duplicates(labels)
        if labels is None:
            if len(genomes) == 1 or self.no_label:
                self.labels = [""] * len(genomes)
            else:
                self.labels = ["{}-".format(i + 1) for i in range(len(genomes))]
        else:
            self.labels = labels
        cfg_labels = self.labels if len(self.labels) == len(self.sg_cfgs) else [None] * len(self.sg_cfgs)
        self.sgs, self.chrs, _nsg = [], [], 0
        for cfgfile, label in zip(self.sg_cfgs, cfg_labels):
            cfg = SGConfig(cfgfile, prefix=label, sep=self.sep)
            self.sgs += cfg.sgs
            self.chrs += cfg.chrs
            _nsg += cfg.nsg
        if not self.nsg or self.nsg < 2:
            self.nsg = _nsg
        if self.no_label:
            self.labels = [""] * len(genomes)

    # ------------------------------------------------------------------ helpers
    def mk_ckpfile(self, file):
        return "{}{}.ok".format(self.tmpdir, os.path.basename(file))

    def update_sgs(self, sgs, d_targets):
        return [[[d_targets.get(c, c) for c in chrs] for chrs in sg] for sg in sgs]

    def parse_assigned(self, d_targets):
        d = {}
        if not self.sg_assigned:
            return d
        for line in open(self.sg_assigned):
            if line.startswith("#") or not line.strip():
                continue
            c, sg = line.strip().split()[:2]
            d[d_targets.get(c, c)] = sg
        return d

    @staticmethod
    def sort_labels(order, labels, chromfiles):
        d = dict(zip(labels, chromfiles))
        out = [(lab, d[lab]) for lab in order if lab in d]
        return [x[0] for x in out], [x[1] for x in out]

    # ------------------------------------------------------------------ run
    def run(self):
        self.outdir = os.path.realpath(self.outdir)
        self.tmpdir = os.path.realpath(self.tmpdir)
        os.makedirs(self.outdir, exist_ok=True)
        os.makedirs(self.tmpdir, exist_ok=True)
        self.outdir += "/"
        self.tmpdir += "/"
        if self.prefix is not None:
            self.outdir += self.prefix
            self.tmpdir += self.prefix

        logger.info("Target chromosomes: {}".format(self.chrs))
        logger.info("Splitting genomes by chromosome into `{}`".format(self.tmpdir))
        ckp_file = self.mk_ckpfile("split")
        ckp = check_ckp(ckp_file)
        split = True
        if isinstance(ckp, list) and len(ckp) == 4 and not self.overwrite:
            chromfiles, labels, d_targets, d_size = ckp
            split = set(d_targets) != set(self.chrs) or not all(os.access(f, os.R_OK) for f in chromfiles)
            if set(d_targets) != set(self.chrs):
                self.re_filter = True
        if split:
            d_targets = parse_idmap(self.target)
            outdir = "{}chromosomes/".format(self.tmpdir)
            os.makedirs(outdir, exist_ok=True)
            data = chromfiles, labels, d_targets, d_size = Seqs.split_genomes(
                self.genomes, self.labels, self.chrs, outdir, d_targets=d_targets, sep=self.sep)
            mk_ckp(ckp_file, *data)
        labels, chromfiles = self.sort_labels(d_targets.values(), labels, chromfiles)
        logger.info("Chromosomes: {}".format(labels))
        logger.info("Chromosome Number: {}".format(len(labels)))
        self.chromfiles, self.labels = chromfiles, labels
        self.sgs = self.update_sgs(self.sgs, d_targets)
        self.sg_assigned = self.parse_assigned(d_targets)
        logger.info("CONFIG: {}".format(self.sgs))
        self.d_size = d_size
        if len(chromfiles) == 0:
            raise ValueError("0 chromosome remained after filtering. Please check the inputs.")
        logger.info("Genome size: {:,} bp".format(sum(d_size.values())))

        logger.info("###Step: Kmer Count")
        logger.info("Counting kmer on the GPU (replaces jellyfish)")
        dumpfiles = run_jellyfish_dumps(chromfiles, k=self.k, ncpu=self.ncpu, lower_count=self.lower_count,
                                        overwrite=self.overwrite, engine=self.engine,
                                        write_dumps=self.write_dumps)

        logger.info("Loading kmer matrix")
        dumps = JellyfishDumps(dumpfiles, labels, ncpu=self.ncpu)
        self.basename = "k{}_q{}_f{}".format(self.k, self.min_freq, self.min_fold)
        self.para_prefix = "{}{}".format(self.outdir, self.basename)
        matfile = self.para_prefix + ".kmer.mat"
        ckp_file = self.mk_ckpfile(matfile)
        d_mat = dumps.to_matrix()
        logger.info("Filtering differential kmers")
        histfig = self.para_prefix + ".kmer_freq." + self.figfmt
        d_mat2 = dumps.filter(d_mat, dumps.lengths, self.sgs, outfig=histfig,
                              min_fold=self.min_fold, baseline=self.baseline, min_freq=self.min_freq,
                              max_freq=self.max_freq, min_prop=self.min_prop, max_prop=self.max_prop,
                              ratio=self.ratio)
        logger.info("{} kmers in total".format(len(d_mat)))
        if len(d_mat2) == 0:
            raise ValueError("0 kmer remained after filtering. Please reset the filter options.")
        if self.overwrite or self.re_filter or not check_ckp(ckp_file) or not os.path.getsize(matfile):
            with open(matfile, "w") as fout:
                dumps.write_matrix(d_mat2, fout)
            try:
                plot_histogram(dumps.hist_tot(), histfig)
            except Exception as e:     # plotting is optional
                logger.warning("histogram not plotted: {}".format(e))
            mk_ckp(ckp_file)

        logger.info("###Step: Cluster")
        cluster = Cluster(d_mat2, n_clusters=self.nsg, sg_prefix="SG", sg_assigned=self.sg_assigned)
        self.d_sg = d_sg = cluster.d_sg
        logger.info("Subgenome assignments: {}".format(dict(d_sg)))
        self.sg_names = cluster.sg_names
        sg_chrs = self.para_prefix + ".chrom-subgenome.tsv"
        with open(sg_chrs, "w") as fout:
            cluster.output_subgenomes(fout)
        sg_kmers = self.para_prefix + ".sig.kmer-subgenome.tsv"
        logger.info("Outputing significant differiential `kmer` - `subgenome` maps to `{}`".format(sg_kmers))
        with open(sg_kmers, "w") as fout:
            d_kmers = cluster.output_kmers(fout, max_pval=self.max_pval, test_method=self.test_method)
        logger.info("{} significant subgenome-specific kmers".format(len(d_kmers) // 2))
        for sg, count in sorted(Counter(d_kmers.values()).items()):
            logger.info("\t{} {}-specific kmers".format(count // 2, sg))
        if self.just_core:
            self.step_final()
            logger.info("Pipeline completed early")
            return

        sg_map = self.para_prefix + ".subgenome.bin.count"
        ckp_file = self.mk_ckpfile(sg_map)
        logger.info("Outputing `coordinate` - `subgenome` maps to `{}`".format(sg_map))
        with open(sg_map, "w") as fout:
            Seqs.map_kmer3(chromfiles, d_kmers, fout=fout, k=self.k, bin_size=10000, sg_names=self.sg_names)
        mk_ckp(ckp_file)
        logger.info("Enriching subgenome by chromosome window (size: {})".format(self.window_size))
        bins, counts = Circos.stack_matrix(sg_map, window_size=self.window_size)
        bin_enrich = self.para_prefix + ".bin.enrich"
        bin_exchange = self.para_prefix + ".bin.group"
        with open(bin_enrich, "w") as fout, open(bin_exchange, "w") as fout2:
            self.sg_lines = Stats.enrich_bin(fout, fout2, self.d_sg, counts, colnames=self.sg_names,
                                             rownames=bins, max_pval=self.max_pval)
        logger.info("Output: {}".format(bin_enrich))

        if self.custom_features is not None:
            feat_map = self.para_prefix + ".custom.bin.count"
            logger.info("Mapping subgenome-specific kmers to custom features: {}".format(self.custom_features))
            with open(feat_map, "w") as fout:
                Seqs.map_kmer3(self.custom_features, d_kmers, fout=fout, k=self.k, bin_size=10000000,
                               sg_names=self.sg_names, chunk=False, log=False)
            logger.info("Enriching subgenome-specific features")
            bins, counts = Circos.stack_matrix(feat_map, window_size=100000000)
            feat_enrich = self.para_prefix + ".custom.enrich"
            with open(feat_enrich, "w") as fout:
                d_enriched, _ = Stats.enrich_ltr(fout, self.d_sg, counts, colnames=self.sg_names,
                                                 rownames=bins, max_pval=self.max_pval)
            logger.info("Output: {}".format(feat_enrich))
            logger.info("{} significant subgenome-specific features".format(len(d_enriched)))
            for sg, count in sorted(Counter(d_enriched.values()).items()):
                logger.info("\t{} {}-specific features".format(count, sg))

        if not self.disable_ltr or not self.disable_circos:
            logger.info("Modules 3-4 (LTR, circos) are not part of this build; run the reference on the "
                        "outputs above, or pass -disable_ltr -disable_circos to silence this note")
        self.step_final()
        logger.info("Pipeline completed")

    def step_final(self):
        if self.cleanup:
            logger.info("Cleaning {}".format(self.tmpdir))
            shutil.rmtree(self.tmpdir, ignore_errors=True)


def main(argv=None):
    args = makeArgparse(argv)
    logger.info("Command: {}".format(" ".join(sys.argv)))
    logger.info("Version: subphaser_amd {}".format(__version__))
    logger.info("Arguments: {}".format(args.__dict__))
    Pipeline(**args.__dict__).run()


if __name__ == "__main__":
    main()
