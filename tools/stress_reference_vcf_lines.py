#!/usr/bin/env python3
"""Build container only: random VCF records (1-12 ALT alleles, SNVs / MNVs / insertions / deletions, missing alleles,
extra FORMAT fields, with and without AF, phased and unphased) through the REFERENCE's VariantRecord.read_vcf_line / split
(variant.py:286-331) and through crisprhawk_hip.variant.VariantRecord: every field the path reads.

    python tools/stress_reference_vcf_lines.py [seconds] [seed]
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
import numpy as np  # noqa: E402
from crisprhawk_hip.variant import VariantRecord as MRecord  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
n_ok = n_err = 0


def allele(n):
    return "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))


def view(vr):
    return dict(position=vr.position, ref=vr.ref, alt=list(vr.alt), vtype=list(vr.vtype), afs=[None if a != a else a for a in vr.afs], ids=list(vr.id),
                filter=vr.filter, contig=vr.contig, samples=[[sorted(s0), sorted(s1)] for s0, s1 in vr.samples])


while time.time() - t0 < budget:
    ns = int(rng.integers(1, 12))
    samples = [f"S{i:03d}" for i in range(ns)]
    ref = allele(int(rng.choice([1, 1, 1, 2, 3, 6])))
    n_alt = int(rng.choice([1, 1, 1, 2, 3, 12]))
    alts = []
    while len(alts) < n_alt:
        a = allele(int(rng.choice([1, 1, 2, 4, 7])))
        if a != ref and a not in alts:
            alts.append(a)
    phased = bool(rng.random() < 0.6)
    sep = "|" if phased else "/"
    info = str(rng.choice(["DP=10", "AF=" + ",".join(f"{rng.random():.4g}" for _ in alts), "AC=1;AF=" + ",".join(f"{rng.random():.3f}" for _ in alts) + ";DB"]))
    extra = bool(rng.random() < 0.3)
    gts = []
    for _ in samples:
        a = [("." if rng.random() < 0.08 else str(int(rng.integers(0, n_alt + 1)))) for _ in range(2)]
        g = sep.join(a)
        if extra:
            g += f":{int(rng.integers(1, 99))}:{rng.random():.2f}"
        gts.append(g)
    fields = ["chrL", str(int(rng.integers(1, 10**8))), str(rng.choice([".", "rs12"])), ref, ",".join(alts), "50", str(rng.choice(["PASS", "q10", "."])), info,
              "GT:DP:GQ" if extra else "GT"] + gts
    got = want = None
    r_err = m_err = None
    try:
        vr = mg.VariantRecord(True)
        vr.read_vcf_line(list(fields), samples, phased)
        want = (view(vr), [view(v) for v in vr.split()])
    except (ValueError, KeyError, IndexError, SystemExit, AssertionError) as e:
        r_err = e
    try:
        mr = MRecord(True)
        mr.read_vcf_line(list(fields), samples, phased)
        got = (view(mr), [view(v) for v in mr.split()])
    except (ValueError, KeyError, IndexError, SystemExit, AssertionError) as e:
        m_err = e
    if (r_err is None) != (m_err is None):
        raise AssertionError((fields[:9], f"reference: {r_err!r}; restatement: {m_err!r}"))
    if r_err is not None:
        n_err += 1
        continue
    assert got == want, (fields, got, want)
    n_ok += 1
print(f"{n_ok} records equal, {n_err} refused by both, in {time.time() - t0:.0f} s")
