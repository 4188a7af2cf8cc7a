"""Synthetic allopolyploid genomes for bench.py (no network => no real genomes).

Shapes follow BASELINE.json `configs` / SURVEY.md section 8(d).  Bases are
generated directly in HBM by sp_synth_chrom (csrc/sp_synth.hip); this module
only fixes chromosome names, lengths, the sg.config structure and the truth
subgenome assignment (used in place of k-means so clustering randomness stays
out of the measurement, like `-sg_assigned`).
"""
import random

CONFIGS = {
    # name: (subgenomes S, homoeologous sets H, mean chromosome length, +- spread, seed)
    "wheat": dict(S=3, H=7, mean=6.67e8, spread=0.15, seed=3, letters="ABD"),
    "peanut": dict(S=2, H=10, mean=1.275e8, spread=0.10, seed=2, letters="AB"),
    "ara": dict(S=2, H=None, mean=2.077e7, spread=0.10, seed=1, letters="AB"),
    "small": dict(S=3, H=3, mean=3.0e7, spread=0.15, seed=4, letters="ABD"),
    "tiny": dict(S=2, H=2, mean=2.0e6, spread=0.10, seed=5, letters="AB"),
}


class SynthGenome:
    def __init__(self, name, scale=1.0):
        cfg = CONFIGS[name]
        self.name, self.S, self.seed = name, cfg["S"], cfg["seed"]
        rng = random.Random(cfg["seed"])
        self.chroms = []      # dicts: label, set_id, sg_id, length, exchange
        self.sgs = []         # sg.config structure (list of sets -> units -> chromosome labels)
        if name == "ara":
            # same shape as example_data/Arabidopsis_suecica_sg.config: 13 chromosomes, 5 + 8,
            # three config lines with comma-joined units
            layout = [([1], [6, 7]), ([2, 3], [9, 8, 10]), ([4, 5], [13, 11, 12])]
            for set_id, (a, b) in enumerate(layout):
                units = []
                for sg_id, ids in enumerate((a, b)):
                    unit = []
                    for cid in ids:
                        lab = "chr%d" % cid
                        ln = int(cfg["mean"] * scale * (1 - cfg["spread"] + 2 * cfg["spread"] * rng.random()))
                        self.chroms.append(dict(label=lab, set_id=set_id, sg_id=sg_id, length=ln // 64 * 64))
                        unit.append(lab)
                    units.append(unit)
                self.sgs.append(units)
            self.chroms.sort(key=lambda c: int(c["label"][3:]))
        else:
            for h in range(cfg["H"]):
                units = []
                for g in range(cfg["S"]):
                    lab = "Chr%d%s" % (h + 1, cfg["letters"][g])
                    ln = int(cfg["mean"] * scale * (1 - cfg["spread"] + 2 * cfg["spread"] * rng.random()))
                    self.chroms.append(dict(label=lab, set_id=h, sg_id=g, length=ln // 64 * 64))
                    units.append([lab])
                self.sgs.append(units)
        for i, c in enumerate(self.chroms):
            c["chrom_id"] = i
            c["exchange"] = 1 if (c["set_id"] == 0 and c["sg_id"] == 0) else 0
        self.labels = [c["label"] for c in self.chroms]
        self.sg_assigned = {c["label"]: "SG%d" % (c["sg_id"] + 1) for c in self.chroms}
        self.total_bases = sum(c["length"] for c in self.chroms)
