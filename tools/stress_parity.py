#!/usr/bin/env python3
"""Randomised parity campaign on the GPU box (not part of the test suite: minutes, not seconds):
  tools/stress_parity.py [seconds] [seed]
Each round draws a region (40-300 kb), a variant mix (SNV / deletion / insertion shares, indel lengths up to 40, densities
from sparse to one variant every ~15 nt, 2-12 samples), a PAM / guide shape, and checks
  * the device expansion against the host-built haplotypes (planes, segments, scan bounds),
  * the search of the expanded set against the oracle (rows, candidates, hits, windows, CFDon),
  * the collapse through both grouping paths against the oracle's grouping.
Prints one line per round; exits non-zero at the first difference."""
import os
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd", "/root/repo/tests"]
import numpy as np

from crisprhawk_hip import synth
from crisprhawk_hip.hapset import DeviceHapSet
from crisprhawk_hip.expand import HaplotypeBuildError
from crisprhawk_hip.workload import build_phased_haplotypes, expand_on_device
from oracle import oracle as ora
import test_gpu_parity as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
big = len(sys.argv) > 3 and sys.argv[3] == "big"  # 0.5-2 Mb x 20-100 samples: many tiles per row, tables of millions of rows
rng = np.random.default_rng(seed)
PAMS = [("NGG", 20, False), ("NGG", 23, False), ("TTTV", 23, True), ("NNGRRT", 21, False), ("NAG", 18, False), ("TTN", 25, True)]
t0 = time.time()
rounds = 0
while time.time() - t0 < budget:
    rounds += 1
    rlen = int(rng.integers(500_000, 2_000_000)) if big else int(rng.integers(40_000, 300_000))
    dens = float(np.exp(rng.uniform(np.log(60 if big else 15), np.log(3000))))       # nt per variant site
    sites = max(5, int(rlen / dens))
    samples = int(rng.integers(20, 101)) if big else int(rng.integers(2, 13))
    fs = float(rng.uniform(0.2, 0.95)); fd = float(rng.uniform(0, 1 - fs))
    mi = int(rng.choice([2, 5, 12, 40]))
    pam, gl, right = PAMS[int(rng.integers(len(PAMS)))]
    reg = synth.make_region(int(rng.integers(1 << 30)), "chrS", rlen + 4000, 1500, 1500 + rlen,
                            iupac_frac=0.001 if rng.random() < 0.3 else 0.0)
    try:
        synth.add_phased_variants(reg, int(rng.integers(1 << 30)), sites, samples, frac_snv=fs, frac_del=fd, max_indel=mi,
                                  af_min=0.05, af_max=0.9)
    except ValueError:
        continue
    tag = f"round {rounds}: {rlen} nt, {len(reg.variants)} sites, {samples} samples, snv {fs:.2f} del {fd:.2f} maxindel {mi}, {pam}/{gl}"
    # 1. expansion
    try:
        haps, info_h = build_phased_haplotypes(reg, len(pam))
    except (KeyError, HaplotypeBuildError) as e:
        # the scan start's position is deleted on some copy (the reference's own KeyError, search_guides.py:49-84), or a
        # variant pushed past the region's original length (its clamp, haplotype.py:199-201): both builders must refuse
        try:
            expand_on_device(reg, len(pam))
        except type(e):
            print(tag, f"refused by both builders: {type(e).__name__} {e}", flush=True)
            continue
        raise AssertionError((tag, f"host builder raised {type(e).__name__}, device path did not"))
    ds, info_d, ms, kept = expand_on_device(reg, len(pam))
    assert len(kept) == len(haps), tag
    want = DeviceHapSet(haps).planes()
    got = ds.planes()
    for j, r in enumerate(kept):
        n = (len(haps[j].seq) + 31) // 32
        assert np.array_equal(got[:, r, :n], want[:, j, :n]), (tag, "planes", j)
        assert not got[:, r, n:].any(), (tag, "pad", j)
        assert np.array_equal(ds.host_meta[r].seg.rel, haps[j].seg.rel) and np.array_equal(ds.host_meta[r].seg.gen, haps[j].seg.gen), (tag, "seg", j)
        assert tuple(ds.host_meta[r].scan) == tuple(haps[j].scan), (tag, "scan", j)
    ds.close()
    # 2. search + 3. collapse (oracle builds its own haplotypes from the same records)
    score = (not right) and len(pam) >= 2 and "N" not in reg.sequence.upper() and not any(c not in "ACGT" for c in reg.sequence.upper())
    try:
        hs, tab = T._oracle_vs_device(reg, pam, gl, right, score)
    except ora.OracleError as e:  # inputs the reference itself crashes on (divergence a'): only the expansion was checked
        print(tag, "oracle refuses:", e, flush=True)
        continue
    for mode in ("sort", "hash"):
        os.environ["HAWK_COLLAPSE_MODE"] = mode
        tab2 = T.device_set(hs).search(*ora.pam_encode(pam)[:2], len(pam), gl, right, collapse=True)
        T._check_collapse(hs, tab2, gl, len(pam), right)
    os.environ.pop("HAWK_COLLAPSE_MODE", None)
    print(tag, "rows", tab.n_rows, "groups", tab2.n_groups, "ok", flush=True)
print(f"{rounds} rounds in {time.time() - t0:.0f} s: all equal")
