#!/usr/bin/env python3
"""Randomised campaign for the searches that run straight from an expansion plan (not part of the test suite):
  tools/stress_views.py [seconds] [seed] [big]
Each round draws a region, a variant mix (SNV / deletion / insertion shares, indel lengths up to 40, densities from sparse to one
variant every ~12 nt, 2-200 samples - many samples share clusters, few do not), a PAM / guide shape, and holds
  * the view searched per dirty word (hawk_vsearch.hip) to the search of the materialised planes, column for column,
  * the view searched per distinct cluster (hawk_csearch.hip) to the same table in the reference's emission order,
  * the collapse of the cluster table to the collapse of the plane table (groups, members, G/C counts),
with the cluster path's first template reservation sometimes too small (the rerun).  The plane search itself is held to the
oracle by tools/stress_parity.py.  Prints one line per round; exits non-zero at the first difference."""
import os
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd", "/root/repo/tests"]
import numpy as np

from crisprhawk_hip import synth
from crisprhawk_hip.expand import HaplotypeBuildError
from crisprhawk_hip.workload import expand_on_device
from oracle import oracle as ora

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
big = len(sys.argv) > 3 and sys.argv[3] == "big"
rng = np.random.default_rng(seed)
PAMS = [("NGG", 20, False), ("NGG", 23, False), ("TTTV", 23, True), ("NNGRRT", 21, False), ("NAG", 18, False), ("TTN", 25, True),
        ("NGN", 20, False), ("NG", 8, False), ("NGG", 40, False)]
COLS = ("hap", "pos", "strand", "start", "stop", "flags")
os.environ["HAWK_CLUSTER_MIN_SHARE"] = "0"


def canonical(t):
    o = np.lexsort((t.pos, t.strand, t.hap))
    d = {c: getattr(t, c)[o] for c in COLS}
    d["win"], d["cfdon"] = t.win[:, o], t.cfdon[o]
    return d


def groups_of(tab):
    """report groups as a set-like structure independent of row order: key -> sorted member haplotypes"""
    g = tab.export_groups()
    out = {}
    for i in range(g.n_groups):
        mem = np.sort(g.member_hap[g.member_off[i]:g.member_off[i + 1]])
        out[(int(g.start[i]), int(g.strand[i]), tuple(int(x) for x in g.win[:, i]))] = (tuple(int(x) for x in mem), int(g.gc_num[i]), int(g.gc_den[i]))
    return out


t0 = time.time()
rounds = paths = 0
while time.time() - t0 < budget:
    rounds += 1
    rlen = int(rng.integers(300_000, 1_500_000)) if big else int(rng.integers(5_000, 200_000))
    dens = float(np.exp(rng.uniform(np.log(40 if big else 12), np.log(3000))))
    sites = max(3, int(rlen / dens))
    samples = int(rng.integers(50, 300)) if big else int(rng.choice([2, 3, 5, 12, 40, 120, 200]))
    fs = float(rng.uniform(0.2, 0.97)); fd = float(rng.uniform(0, 1 - fs))
    mi = int(rng.choice([2, 5, 12, 40]))
    pam, gl, right = PAMS[int(rng.integers(len(PAMS)))]
    reg = synth.make_region(int(rng.integers(1 << 30)), "chrS", rlen + 4000, int(rng.choice([0, 700, 1500])), 1500 + rlen,
                            iupac_frac=0.001 if rng.random() < 0.3 else 0.0)
    try:
        synth.add_phased_variants(reg, int(rng.integers(1 << 30)), sites, samples, frac_snv=fs, frac_del=fd, max_indel=mi,
                                  af_min=float(rng.choice([0.002, 0.05])), af_max=float(rng.choice([0.3, 0.9])), edge_margin=int(rng.choice([1, 30])))
    except ValueError:
        continue
    tag = f"round {rounds}: {rlen} nt, {len(reg.variants)} sites, {samples} samples, snv {fs:.2f} del {fd:.2f} maxindel {mi}, {pam}/{gl}{' right' if right else ''}"
    score = (not right) and len(pam) >= 2
    mm, pt = synth.cfd_tables() if score else (None, None)
    bits, bitsrc = ora.pam_encode(pam)[:2]
    try:
        ds, _info, _ms, _kept = expand_on_device(reg, len(pam), keep_plan=True)
    except (KeyError, HaplotypeBuildError) as e:
        print(tag, f"refused: {type(e).__name__}", flush=True)
        continue
    a = ds.search(bits, bitsrc, len(pam), gl, right, mm, pt, cfd_na_on_ambiguous=True)
    view = ds.plan.view()
    os.environ["HAWK_VIEW_SEARCH"] = "words"
    b = view.search(bits, bitsrc, len(pam), gl, right, mm, pt, cfd_na_on_ambiguous=True)
    del os.environ["HAWK_VIEW_SEARCH"]
    assert (b.n_rows, b.n_candidates, b.n_hits) == (a.n_rows, a.n_candidates, a.n_hits), (tag, "words totals")
    for c in COLS:
        assert np.array_equal(getattr(a, c), getattr(b, c)), (tag, "words", c)
    assert np.array_equal(a.win, b.win) and np.array_equal(a.cfdon, b.cfdon, equal_nan=True), (tag, "words win / cfdon")
    st = ds.plan.cluster_stats()
    note = f"clusters {st['instances']}/{st['distinct']}"
    if st["usable"]:
        paths += 1
        small = rng.random() < 0.3
        if small:
            os.environ["HAWK_CLUSTER_ROWS0"] = str(int(rng.integers(1, 200)))
        c = view.search(bits, bitsrc, len(pam), gl, right, mm, pt, cfd_na_on_ambiguous=True)
        os.environ.pop("HAWK_CLUSTER_ROWS0", None)
        assert c.timing["v_path"] == 2, tag
        assert (c.n_rows, c.n_candidates, c.n_hits) == (a.n_rows, a.n_candidates, a.n_hits), (tag, "cluster totals", (c.n_rows, c.n_candidates, c.n_hits), (a.n_rows, a.n_candidates, a.n_hits))
        ca, cc = canonical(a), canonical(c)
        for k in ca:
            assert np.array_equal(ca[k], cc[k], equal_nan=(k == "cfdon")), (tag, "cluster", k)
        assert (np.diff(c.hap.astype(np.int64)) >= 0).all(), (tag, "haplotype-major")
        if a.n_rows and rng.random() < 0.5:
            a2 = ds.search(bits, bitsrc, len(pam), gl, right, mm, pt, download=False, collapse=True, cfd_na_on_ambiguous=True)
            ga = groups_of(a2)
            c2 = view.search(bits, bitsrc, len(pam), gl, right, mm, pt, download=False, collapse=True, cfd_na_on_ambiguous=True)
            assert groups_of(c2) == ga, (tag, "collapse")
            note += f" groups {len(ga)}"
    else:
        note += f" (dictionary not usable: status {st['status']})"
    ds.plan.close()
    ds.close()
    print(tag, "rows", a.n_rows, note, "ok", flush=True)
print(f"{rounds} rounds ({paths} through the cluster path) in {time.time() - t0:.0f} s: all equal")
