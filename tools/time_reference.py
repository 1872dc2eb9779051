#!/usr/bin/env python3
"""Times the REFERENCE's own Python search path in the build container ("How the CPU side is timed", item 1;
/root/reference does not exist on the GPU box, so this never runs there): per-stage wall time of haplotype construction,
encode, pam_search, search, reverse_guides and CFDon (synthetic tables) on C1 exactly and on C3 restricted to REF + the
first H in {1, 2, 4, 8} samples at 100 kb and 1 Mb, single thread (the search ignores -t), and the linear extrapolation to
C3's 5009 haplotypes.  The reference is imported exactly as tests/golden/make_golden.py imports it.

    python tools/time_reference.py [--quick] > profiles/r02_reference_python_timing.json
"""
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isdir("/root/reference/src"):
    sys.exit("tools/time_reference.py runs in the build container only (/root/reference is absent)")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)  # installs the import stand-ins and imports the reference's modules; generates nothing
import numpy as np  # noqa: E402
from crisprhawk_hip import synth  # noqa: E402


def time_case(name, reg, pam_s="NGG", guidelen=20, right=False):
    t = {}
    t0 = time.perf_counter()
    region = mg._ref_region(reg)
    haps, variants_present, phased = mg._ref_haplotypes(reg, region)
    for i, h in enumerate(haps):
        h.id = f"hap_{i:08d}"  # the reference draws random ids here (haplotypes.py:807-814)
    t["haplotypes (region + VariantRecord + solve_haplotypes_phased)"] = time.perf_counter() - t0
    pam = mg.R_pam.PAM(pam_s, right, True)
    pam.encode(0)
    t0 = time.perf_counter()
    bits = [mg.R_encoder.encode(h.sequence.sequence, 0, True) for h in haps]
    t["encode"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    hits = mg.R_search.pam_search(pam, region, haps, bits, 0, True)
    t["pam_search"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    guides = mg.R_search.search(pam, region, haps, bits, guidelen, right, variants_present, phased, 0, True)
    t["search (pam_search again + retrieve_guides + remove_redundant_guides)"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for g in guides:  # annotation.reverse_guides (annotation.py:27-51)
        if g.strand == 1:
            g.reverse_complement()
    t["reverse_guides"] = time.perf_counter() - t0
    mm, pt = synth.cfd_tables()
    mmd, pamd = synth.cfd_tables_as_dicts(mm, pt)
    t0 = time.perf_counter()
    groups = mg.R_search.group_guides_position(guides, True)
    for _, grp in groups.items():
        gref, members = grp[0], grp[1]
        if gref is not None:
            for sg in members:
                mg.R_cfd.compute_cfd(gref.guide, sg.guide, sg.pam[-2:], mmd, pamd, True)
    t["cfdon_score (compute_cfd over the position groups, synthetic tables)"] = time.perf_counter() - t0
    # the metric's numerator: PAM hits, both strands, all haplotypes, passing is_pamhit_in_range
    cand = 0
    for h, (fwd, rev) in zip(haps, hits):
        for strand, lst in ((0, fwd), (1, rev)):
            rg = (right and strand == 0) or (not right and strand == 1)  # search_guides.py:538
            cand += sum(1 for p in lst if mg.R_search.is_pamhit_in_range(p, guidelen, len(pam), len(h), rg))
    path = sum(v for k, v in t.items() if not k.startswith("haplotypes"))
    return {"case": name, "haplotypes": len(haps), "region_nt": len(reg.sequence), "variant_sites": len(reg.variants),
            "candidates": cand, "guides": len(guides), "seconds": {k: round(v, 4) for k, v in t.items()},
            "path_seconds (encode + pam_search + search + reverse_guides + cfdon)": round(path, 4),
            "candidates_per_s": cand / path if path else None}


def main():
    quick = "--quick" in sys.argv
    out = {"what": "the reference's Python path timed in the build container (8 vCPU, single thread), imported as tests/golden/make_golden.py "
                   "imports it; C3 restricted to REF + the first H samples",
           "host": {"cpus": os.cpu_count()}, "cases": []}
    out["cases"].append(time_case("C1 (10 kb, no VCF)", synth.config_c1()))
    # 1 Mb stops at 4 samples (9 haplotypes): the reference's haplotype construction rewrites two 10^6-entry dicts per
    # carried indel and takes ~3 min per haplotype there - it is not part of the metric, but it has to run first
    for region_len in ((100_000,) if quick else (100_000, 1_000_000)):
        for H in ((1, 2) if quick else ((1, 2, 4, 8) if region_len < 1_000_000 else (1, 2, 4))):
            reg = synth.config_c3(n_samples=2504, n_sites=int(31_000 * region_len / 1_000_000), region_len=region_len)
            reg.samples = reg.samples[:H]
            for v in reg.variants:
                v.gt = v.gt[:H]
            reg.variants = [v for v in reg.variants if v.gt.any()]
            reg.gt_matrix = None
            r = time_case(f"C3 restricted: {region_len} nt, first {H} samples", reg)
            out["cases"].append(r)
            print(r["case"], r["haplotypes"], "haplotypes", round(r["path_seconds (encode + pam_search + search + reverse_guides + cfdon)"], 2), "s",
                  file=sys.stderr, flush=True)
    # extrapolation: seconds per haplotype-Mb of the largest case -> C3's 5009 haplotypes x 1 Mb
    big = out["cases"][-1]
    per_hap = big["path_seconds (encode + pam_search + search + reverse_guides + cfdon)"] / big["haplotypes"] / (big["region_nt"] / 1_000_201)
    out["extrapolation_to_c3"] = {"seconds_per_haplotype_Mb": per_hap, "c3_seconds_5009_haplotypes": per_hap * 5009,
                                  "c3_candidates_per_s": big["candidates_per_s"],
                                  "note": "linear in haplotypes; the full-scale reference run is infeasible in memory (two 1 M-entry dicts per haplotype)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
