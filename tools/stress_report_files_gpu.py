#!/usr/bin/env python3
"""GPU box: FASTA + BED + VCF files written from the campaign's inputs -> pipeline.search_files -> the TSV must be the text
the REFERENCE wrote for the same inputs (tools/campaign_report_fixtures.py).  Unphased cases: haplotype ids are matched by
count (the reference numbers window haplotypes in set order).

    python tools/stress_report_files_gpu.py tools/_campaign/reports.json.gz [--force-host-builder]
"""
import gzip
import io
import json
import os
import sys
import tempfile

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd", "/root/repo/tests"]
import pandas as pd
from crisprhawk_hip import pipeline, readers, synth

cases = json.load(gzip.open(sys.argv[1], "rt"))
if "--force-host-builder" in sys.argv:  # phased records through the fallback the pipeline takes when the device expansion declines them
    from crisprhawk_hip.expand import HaplotypeBuildError

    def _refuse(*a, **k):
        raise HaplotypeBuildError("forced: host haplotype builder")
    pipeline.expand_from_vcf = _refuse
n_ok = 0
with tempfile.TemporaryDirectory() as tmp:
    for k, fx in enumerate(cases):
        d = os.path.join(tmp, f"c{k}")
        os.makedirs(d)
        contig_seq = "N" * (fx["startp"] - 1) + fx["region_seq"] + "ACGT" * 10
        fa, bed, vcf = os.path.join(d, "g.fa"), os.path.join(d, "r.bed"), os.path.join(d, "v.vcf")
        readers.write_fasta(fa, fx["contig"], contig_seq, 80)
        with open(bed, "w") as f:
            f.write(f"{fx['contig']}\t{fx['bed_start']}\t{fx['bed_stop']}\n")
        sep = "/" if fx["unphased"] else "|"
        rows = [[fx["contig"], str(p), ".", r, a, ".", "PASS", f"AF={af:.6g}", "GT"] + [f"{g[0]}{sep}{g[1]}" for g in gts]
                for p, r, a, af, gts in fx["variants"]]
        readers.write_vcf(vcf, fx["contig"], fx["samples"], rows, False)
        (path,) = pipeline.search_files(fa, bed, [vcf], fx["pam"], fx["guidelen"], fx["right"], os.path.join(d, "out"),
                                        cfd_tables=synth.cfd_tables() if fx["cfd"] else None).values()
        text = open(path).read()
        if not fx["unphased"]:
            assert text == fx["report_tsv"], (k, fx["pam"], "phased report differs")
        else:
            got = pd.read_csv(io.StringIO(text), sep="\t", dtype=str, keep_default_na=False)
            want = pd.read_csv(io.StringIO(fx["report_tsv"]), sep="\t", dtype=str, keep_default_na=False)
            assert list(got.columns) == list(want.columns) and len(got) == len(want), (k, "unphased shape", len(got), len(want))
            for c in got.columns:
                if c != "haplotype_id":
                    assert (got[c] == want[c]).all(), (k, c, got[c][got[c] != want[c]].head().tolist(), want[c][got[c] != want[c]].head().tolist())
            assert (got["haplotype_id"].str.count(",") == want["haplotype_id"].str.count(",")).all(), (k, "haplotype id counts")
        n_ok += 1
print(f"{n_ok} reports ({sum(c['unphased'] for c in cases)} unphased): equal to the reference's TSV")
