#!/usr/bin/env python3
"""Build container only: N random regions (phased and unphased VCF records, four PAM / guide shapes) through the REFERENCE's
search -> annotation -> CFDon -> report chain (as tests/golden/make_golden.py:g7_report runs it), written as one gzip'd JSON
of inputs and the TSV text the reference produced, for tools/stress_report_files_gpu.py on the GPU box.  Campaign data,
not a committed fixture.

    python tools/campaign_report_fixtures.py N seed out.json.gz [--iupac]
"""
import gzip
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isdir("/root/reference/src"):
    sys.exit("runs in the build container only (/root/reference is absent)")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
import numpy as np  # noqa: E402
from crisprhawk_hip import synth  # noqa: E402

n, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(seed)
PAMS = [("NGG", 20, False, True), ("TTTV", 23, True, False), ("NNGRRT", 21, False, False), ("NGG", 19, False, True)]
cases = []
tries = 0
while len(cases) < n and tries < 20 * n:
    tries += 1
    rlen = int(rng.integers(800, 8000))
    b0 = int(rng.integers(200, 1500))
    reg = synth.make_region(int(rng.integers(1 << 30)), "chrR", b0 + rlen + int(rng.integers(200, 1500)), b0, b0 + rlen,
                            iupac_frac=float(rng.choice([0.0, 0.0, 0.003])) if "--iupac" in sys.argv else 0.0)
    sites = max(2, int(rlen / float(np.exp(rng.uniform(np.log(20), np.log(400))))))
    unphased = bool(rng.random() < 0.35)
    try:
        synth.add_phased_variants(reg, int(rng.integers(1 << 30)), sites, int(rng.integers(1, 7)), frac_snv=float(rng.uniform(0.4, 0.95)),
                                  frac_del=float(rng.uniform(0.0, 0.3)), max_indel=int(rng.choice([2, 5, 10])), af_min=0.1, af_max=0.9)
        pam_s, gl, right, cfd = PAMS[int(rng.integers(len(PAMS)))]
        obj = mg.g7_report(None, reg, pam_s, gl, right, cfd=cfd, unphased=unphased)
    except (KeyError, ValueError, IndexError, SystemExit):
        continue  # inputs the reference itself refuses
    del obj["haplotypes"]
    cases.append(obj)
with gzip.open(out, "wt") as f:
    json.dump(cases, f, separators=(",", ":"))
print(len(cases), "cases ->", out, os.path.getsize(out), "bytes;", sum(c["unphased"] for c in cases), "unphased")
