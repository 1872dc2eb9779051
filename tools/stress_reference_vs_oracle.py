#!/usr/bin/env python3
"""Build container only (the reference cannot travel): random regions through the REFERENCE's own Python path and through
the oracle (oracle/hawk_oracle.c), which the GPU path is compared with everywhere else.  Haplotype strings and position
maps, scan bounds, PAM hit lists, the guide list of search() in the reference's order, CFDon.

    python tools/stress_reference_vs_oracle.py [seconds] [seed]
"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isdir("/root/reference/src"):
    sys.exit("runs in the build container only (/root/reference is absent)")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from crisprhawk_hip import synth  # noqa: E402
from oracle import oracle as ora  # noqa: E402
from util import oracle_haplotypes  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
PAMS = [("NGG", 20, False), ("NGG", 23, False), ("TTTV", 23, True), ("NNGRRT", 21, False), ("NAG", 18, False), ("TTN", 25, True), ("NGN", 20, False)]
t0 = time.time()
n_ok = n_ref_err = 0
mm, pt = synth.cfd_tables()
mmd, pamd = synth.cfd_tables_as_dicts(mm, pt)
while time.time() - t0 < budget:
    rlen = int(rng.integers(1_000, 12_000))
    b0 = int(rng.integers(150, 2_000))
    reg = synth.make_region(int(rng.integers(1 << 30)), "chrZ", b0 + rlen + int(rng.integers(150, 2_000)), b0, b0 + rlen,
                            iupac_frac=float(rng.choice([0.0, 0.0, 0.004])))
    sites = max(2, int(rlen / float(np.exp(rng.uniform(np.log(12), np.log(400))))))
    try:
        synth.add_phased_variants(reg, int(rng.integers(1 << 30)), sites, int(rng.integers(1, 7)), frac_snv=float(rng.uniform(0.2, 0.95)),
                                  frac_del=float(rng.uniform(0.0, 0.4)), max_indel=int(rng.choice([2, 5, 12])), af_min=0.1, af_max=0.9,
                                  edge_margin=int(rng.choice([1, 20])))
    except ValueError:
        continue
    pam_s, gl, right = PAMS[int(rng.integers(len(PAMS)))]
    tag = f"{rlen} nt, {len(reg.variants)} sites, {len(reg.samples)} samples, {pam_s}/{gl}"
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    # ---- reference ----
    try:
        region = mg._ref_region(reg)
        haps, vp, phased = mg._ref_haplotypes(reg, region)
        for i, h in enumerate(haps):
            h.id = f"hap_{i:08d}"
        pam = mg.R_pam.PAM(pam_s, right, True)
        pam.encode(0)
        bits = [mg.R_encoder.encode(h.sequence.sequence, 0, True) for h in haps]
        r_scan = [list(mg.R_search.compute_scan_start_stop(h, region.start, region.stop, len(pam))) for h in haps]
        r_hits = mg.R_search.pam_search(pam, region, haps, bits, 0, True)
        guides = mg.R_search.search(pam, region, haps, bits, gl, right, vp, phased, 0, True)
    except (KeyError, ValueError, SystemExit, IndexError) as e:
        # the reference refuses / crashes: the oracle must refuse too
        try:
            oh = oracle_haplotypes(fx)
            for h in oh:
                ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, len(pam_s))
        except ora.OracleError:
            n_ref_err += 1
            continue
        raise AssertionError((tag, f"reference raised {type(e).__name__}: {e}; the oracle did not"))
    # ---- oracle ----
    oh = oracle_haplotypes(fx)
    assert len(oh) == len(haps), (tag, "haplotype count")
    for a, b in zip(haps, oh):
        assert a.sequence.sequence == b["seq"], (tag, "haplotype sequence")
        assert [a.posmap[i] for i in range(len(a.posmap))] == b["posmap"].tolist(), (tag, "posmap")
        assert sorted(a.samples.split(",")) == b["samples"], (tag, "samples")
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, len(pam_s)) for h in oh]
    assert [list(s) for s in scan] == r_scan, (tag, "scan bounds")
    hs = ora.HapSet([h["seq"] for h in oh], [h["posmap"] for h in oh], [h["samples"] == ["REF"] for h in oh], scan)
    want = ora.search(hs, pam_s, gl, right)
    hapidx = {h.id: i for i, h in enumerate(haps)}
    got = [(g.start, g.stop, g.strand, hapidx[g.hapid], g.sequence) for g in guides]
    mine = [(int(r["start"]), int(r["stop"]), int(r["strand"]), int(r["hap"]), w) for r, w in zip(want.guides, want.windows)]
    assert got == mine, (tag, "guides", len(got), len(mine))
    assert want.n_hits == sum(len(f) + len(r) for f, r in r_hits), (tag, "hits")
    if (not right) and len(pam_s) >= 2:
        for g in guides:
            if g.strand == 1:
                g.reverse_complement()
        groups = mg.R_search.group_guides_position(guides, True)
        gid = {id(g): i for i, g in enumerate(guides)}
        ref_cfd = {}
        bad = False
        for _, grp in groups.items():
            gref, members = grp[0], grp[1]
            for sg in members:
                if gref is None:
                    ref_cfd[gid[id(sg)]] = None
                else:
                    try:
                        ref_cfd[gid[id(sg)]] = mg.R_cfd.compute_cfd(gref.guide, sg.guide, sg.pam[-2:], mmd, pamd, True)
                    except (KeyError, SystemExit, Exception):
                        bad = True
        if not bad:
            _, _, _, cfd, order = ora.reverse_and_cfdon(want, hs.is_ref, gl, len(pam_s), mm, pt)
            for i in range(len(guides)):
                r, o = ref_cfd[i], cfd[i]
                assert (r is None and o != o) or (r is not None and r == o), (tag, "cfdon", i, r, o)
    n_ok += 1
    if n_ok % 50 == 0:
        print(n_ok, "cases equal,", n_ref_err, "refused by both;", tag, flush=True)
print(f"{n_ok} cases equal, {n_ref_err} refused by both reference and oracle, in {time.time() - t0:.0f} s")
