#!/usr/bin/env python3
"""Randomised campaign, part 3 (GPU box; minutes): tools/stress_reports.py [seconds] [seed]
The guide report of a random region assembled three ways must be one TSV: row by row from the collapsed device table
(reports.report_frame, the restatement of the reference's per-guide assembly that the g7 fixtures pin byte for byte),
in columns from the exported groups (reports.report_from_groups, the fast path), and tile by tile
(tiling.TiledRegionSearch -> report_from_groups; haplotype ids differ per tile and are left out)."""
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd", "/root/repo/tests"]
import numpy as np

from crisprhawk_hip import reports, synth
from crisprhawk_hip.expand import HaplotypeBuildError
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.tiling import TiledRegionSearch, VariantPanel
from crisprhawk_hip.workload import expand_on_device, hap_labels, row_labels

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
PAMS = [("NGG", 20, False), ("TTTV", 23, True), ("NNGRRT", 21, False)]
t0 = time.time()
n_ok = n_skip = 0
while time.time() - t0 < budget:
    pam_s, gl, right = PAMS[int(rng.integers(len(PAMS)))]
    rlen = int(rng.integers(15_000, 50_000))
    reg = synth.make_region(int(rng.integers(1 << 30)), "chrR", rlen + 3000, 1000, 1000 + rlen)
    sites = max(5, int(rlen / float(np.exp(rng.uniform(np.log(30), np.log(600))))))
    try:
        synth.add_phased_variants(reg, int(rng.integers(1 << 30)), sites, int(rng.integers(2, 10)), frac_snv=float(rng.uniform(0.3, 0.9)),
                                  frac_del=float(rng.uniform(0.03, 0.3)), max_indel=int(rng.choice([3, 8, 20])), af_min=0.1, af_max=0.8)
    except ValueError:
        continue
    pam = PAM(pam_s, right, True)
    pam.encode(0)
    score = pam_s == "NGG"
    mm, pt = synth.cfd_tables() if score else (None, None)
    tag = f"report: {rlen} nt, {len(reg.variants)} sites, {len(reg.samples)} samples, {pam_s}/{gl}"
    try:
        ds, info, _ms, kept = expand_on_device(reg, len(pam_s))
    except (KeyError, HaplotypeBuildError):
        n_skip += 1
        continue
    tab = ds.search(pam.bits, pam.bitsrc, len(pam_s), gl, right, mm, pt, download=False, collapse=True)
    target = f"{reg.contig}:{reg.bed_start}-{reg.bed_stop}"
    g = tab.export_groups()  # while the table is device-resident
    df1 = reports.report_frame(reports.ReportInput.from_table(tab), row_labels(reg, ds, info, kept), pam, reg.contig, target, None, score)
    df2 = reports.report_from_groups(g, hap_labels(reg.contig, reg.variants, ds, info, kept), pam, reg.contig, target,
                                     is_ref_hap=np.asarray(ds.is_ref, dtype=bool), with_cfdon=score)
    t1, t2 = df1.to_csv(sep="\t", index=False), reports.to_tsv(df2)
    assert t1 == t2, (tag, "row-level vs columnar")
    try:
        trs = TiledRegionSearch(lambda lo, hi: reg.contig_seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, VariantPanel.from_region(reg), pam, gl,
                                right, tile_nt=int(rng.integers(2000, 9000)), flank=600)
        mg = trs.run(cfd=(mm, pt) if score else None)
    except ValueError as e:
        if "flank too small" not in str(e):
            raise
        mg = None
    if mg is not None:
        df3 = reports.report_from_groups(mg.groups(), mg.labels, pam, reg.contig, target, with_cfdon=score)
        cols = [c for c in df1.columns if c != "haplotype_id"]
        assert df1[cols].to_csv(sep="\t", index=False) == df3[cols].to_csv(sep="\t", index=False), (tag, "tiled")
    tab.close(); ds.close()
    n_ok += 1
    print(tag, "rows", len(df1), "ok", flush=True)
print(f"{n_ok} reports, {n_skip} inputs refused, in {time.time() - t0:.0f} s: all equal")
