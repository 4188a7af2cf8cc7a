"""Deterministic toy allopolyploid genome shared by the tests and by
tests/golden/gen_golden.py (the fixture generator).  Pure numpy, seed-stable."""
import numpy as np

_B = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rand_seq(rng, n):
    return _B[rng.randint(0, 4, size=n)]


def _mutate(rng, seq, rate):
    seq = seq.copy()
    m = rng.random_sample(seq.size) < rate
    n = int(m.sum())
    if n:
        # substitute with a different base
        cur = np.searchsorted(_B, seq[m])
        seq[m] = _B[(cur + rng.randint(1, 4, size=n)) % 4]
    return seq


def make_toy_genome(seed=7, n_sets=3, chrom_len=40000, copies=25):
    rng = np.random.RandomState(seed)
    sgs_names = ["A", "B"]
    lib = {sg: [_rand_seq(rng, rng.randint(300, 800)) for _ in range(6)] for sg in sgs_names}
    shared = [_rand_seq(rng, rng.randint(300, 800)) for _ in range(4)]
    seqs, labels = {}, []
    for h in range(1, n_sets + 1):
        backbone = _rand_seq(rng, chrom_len + rng.randint(-3000, 3000))
        for sg in sgs_names:
            lab = "%s%d" % (sg, h)
            s = _mutate(rng, backbone, 0.08)
            n = s.size
            # planted exchange: the last quarter of A1 carries B's repeats
            for fam_i in range(6):
                for _ in range(copies):
                    pos = rng.randint(200, n - 1000)
                    src_sg = sg
                    if lab == "A1" and pos > 3 * n // 4:
                        src_sg = "B"
                    elif lab == "A1" and pos > 3 * n // 4 - 800:
                        continue
                    el = _mutate(rng, lib[src_sg][fam_i], rng.uniform(0, 0.04))
                    s[pos:pos + el.size] = el[: n - pos]
            for fam in shared:
                for _ in range(copies // 2):
                    pos = rng.randint(200, n - 1000)
                    el = _mutate(rng, fam, rng.uniform(0, 0.04))
                    s[pos:pos + el.size] = el[: n - pos]
            # hot keys: telomere-like repeat at both ends, poly-A stretch
            tel = np.frombuffer(b"TTTAGGG" * 40, dtype=np.uint8)
            s[: tel.size] = tel
            s[n - tel.size:] = tel
            s[5000:5060] = ord("A")
            # soft-masked stretches, N runs, a few IUPAC codes
            txt = bytearray(s.tobytes())
            for _ in range(12):
                p = rng.randint(0, n - 400)
                txt[p:p + 300] = bytes(txt[p:p + 300]).lower()
            for p in (1999, 2000 + 17, 9990, n // 2):
                txt[p:p + rng.randint(1, 30)] = b"N" * len(txt[p:p + rng.randint(1, 30)])
            for p, ch in ((1234, b"R"), (4321, b"y"), (7777, b"n"), (15000, b"K")):
                txt[p:p + 1] = ch
            seqs[lab] = txt.decode()
            labels.append(lab)
    sgs = [[["A%d" % h], ["B%d" % h]] for h in range(1, n_sets + 1)]
    sgs3 = [[["A1"], ["B1"], ["A3"]], [["A2"], ["B2"], ["B3"]]]
    sgs_grouped = [[["A1", "A2"], ["B1", "B2"]], [["A3"], ["B3"]]]
    sg_assigned = {lab: ("SG1" if lab.startswith("A") else "SG2") for lab in labels}
    feats = []
    frng = np.random.RandomState(seed + 1)
    for i in range(14):
        lab = labels[i % len(labels)]
        n = len(seqs[lab])
        st = int(frng.randint(0, n - 3000))
        ln = int(frng.randint(10, 2500))
        feats.append(("%s:%d-%d" % (lab, st, st + ln), seqs[lab][st:st + ln]))
    feats.append(("tiny:1-5", "ACGTA"))
    feats.append(("weird_id_without_coords", seqs["B2"][100:900]))
    return {"labels": labels, "seqs": seqs, "sgs": sgs, "sgs3": sgs3, "sgs_grouped": sgs_grouped,
            "sg_assigned": sg_assigned, "features": feats}


# ---------------------------------------------------------------------------------------------
# Toys with the chromosome / set STRUCTURE of the three BASELINE genomes (the filter's set loop,
# its 8-wide table batching and the comma-grouped units depend on the shape, not on the size):
#   wheat   21 chromosomes, 7 config lines x 3 single-chromosome units   (example_data/wheat_sg.config)
#   peanut  20 chromosomes, 10 lines x 2 units                            (example_data/peanut_sg.config)
#   ara     13 chromosomes, 3 lines with comma-joined units, 5 + 8        (example_data/Arabidopsis_suecica_sg.config)
SHAPES = {
    "wheat": dict(letters="ABD", layout=[[["Chr%d%s" % (h, g)] for g in "ABD"] for h in range(1, 8)]),
    "peanut": dict(letters="AB", layout=[[["Arahy.%02d" % h], ["Arahy.%02d" % (h + 10)]] for h in range(1, 11)]),
    "ara": dict(letters="AB", layout=[[["c1"], ["c6", "c7"]], [["c2", "c3"], ["c9", "c8", "c10"]],
                                      [["c4", "c5"], ["c13", "c11", "c12"]]]),
}


def make_shape_genome(shape, seed=11, chrom_len=9000, copies=9):
    cfg = SHAPES[shape]
    rng = np.random.RandomState(seed)
    S = len(cfg["letters"])
    lib = [[_rand_seq(rng, rng.randint(120, 360)) for _ in range(5)] for _ in range(S)]
    shared = [_rand_seq(rng, rng.randint(120, 360)) for _ in range(3)]
    seqs, sg_of = {}, {}
    for set_id, units in enumerate(cfg["layout"]):
        backbone = _rand_seq(rng, chrom_len + rng.randint(-1500, 1500))
        for sg_id, unit in enumerate(units):
            for lab in unit:
                s = _mutate(rng, backbone, 0.08)[: chrom_len + rng.randint(-1200, 1200)]
                n = s.size
                for fam in lib[sg_id]:
                    for _ in range(copies):
                        pos = rng.randint(50, n - 400)
                        el = _mutate(rng, fam, rng.uniform(0, 0.05))
                        s[pos:pos + el.size] = el[: n - pos]
                for fam in shared:
                    for _ in range(copies // 2):
                        pos = rng.randint(50, n - 400)
                        el = _mutate(rng, fam, rng.uniform(0, 0.05))
                        s[pos:pos + el.size] = el[: n - pos]
                tel = np.frombuffer(b"TTTAGGG" * 12, dtype=np.uint8)
                s[: tel.size] = tel
                txt = bytearray(s.tobytes())
                p = rng.randint(0, n - 200)
                txt[p:p + 150] = bytes(txt[p:p + 150]).lower()
                p = rng.randint(100, n - 100)
                txt[p:p + 7] = b"NNNNNNN"
                seqs[lab] = txt.decode()
                sg_of[lab] = "SG%d" % (sg_id + 1)
    if shape == "ara":
        labels = sorted(seqs, key=lambda x: int(x[1:]))
    else:
        labels = [lab for units in cfg["layout"] for unit in units for lab in unit]
    return {"labels": labels, "seqs": seqs, "sgs": cfg["layout"], "sg_assigned": {l: sg_of[l] for l in labels},
            "n_sg": S}
