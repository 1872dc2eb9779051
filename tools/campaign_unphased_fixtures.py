#!/usr/bin/env python3
"""Build container only: N random unphased cases run through the REFERENCE (haplotype construction + search with
resolve_guide), written as one gzip'd JSON of inputs (generator parameters) and outputs for tools/stress_unphased_gpu.py
to check the GPU path against on the GPU box.  The file is campaign data, not a committed fixture.

    python tools/campaign_unphased_fixtures.py N seed out.json.gz
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

n, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(seed)
PAMS = [("NGG", 20, False), ("TTTV", 23, True), ("NNGRRT", 21, False), ("NAG", 19, False)]
cases = []
tries = 0
while len(cases) < n and tries < 20 * n:
    tries += 1
    rlen = int(rng.integers(800, 5000))
    b0 = int(rng.integers(200, 1500))
    sp = dict(region=[int(rng.integers(1 << 30)), "chrU", b0 + rlen + int(rng.integers(200, 1500)), b0, b0 + rlen],
              variants=[int(rng.integers(1 << 30)), max(2, int(rlen / float(np.exp(rng.uniform(np.log(20), np.log(300)))))), int(rng.integers(1, 7))],
              kw=dict(frac_snv=float(rng.uniform(0.4, 0.95)), frac_del=float(rng.uniform(0.0, 0.3)), max_indel=int(rng.choice([2, 5, 10])),
                      af_min=0.1, af_max=0.9))
    pam_s, gl, right = PAMS[int(rng.integers(len(PAMS)))]
    try:
        obj = mg.g4_unphased_case(sp, pam_s, gl, right)
    except (KeyError, ValueError, IndexError, SystemExit):
        continue  # inputs the reference itself refuses
    del obj["haplotypes"], obj["hits"], obj["scan"]  # the GPU check goes records -> guides end to end
    cases.append(obj)
with gzip.open(out, "wt") as f:
    json.dump(cases, f, separators=(",", ":"))
print(len(cases), "cases ->", out, os.path.getsize(out), "bytes")
