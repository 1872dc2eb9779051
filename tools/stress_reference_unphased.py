#!/usr/bin/env python3
"""Build container only: random unphased VCF records through the REFERENCE's haplotype construction (IUPAC-encoded SNV
haplotypes per sample + one set of window haplotypes per indel, haplotypes.py:370-712) and through this package's
restatement (crisprhawk_hip/haplotypes.py); sequences, coordinates, samples, variants, position maps, variant_alleles.

    python tools/stress_reference_unphased.py [seconds] [seed]
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
from crisprhawk_hip import haplotypes as H  # noqa: E402
from crisprhawk_hip import synth  # noqa: E402
from crisprhawk_hip.coordinate import Coordinate as MCoordinate  # noqa: E402
from crisprhawk_hip.haplotype import Haplotype as MHaplotype  # noqa: E402
from crisprhawk_hip.hapset import segments_from_posmap  # noqa: E402
from crisprhawk_hip.region import Region as MRegion  # noqa: E402
from crisprhawk_hip.sequence import Sequence as MSequence  # noqa: E402
from crisprhawk_hip.variant import VariantRecord as MVariantRecord  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
n_ok = n_err = 0


def sig(seq, coord, samples, variants, breaks, n, va):
    return (seq, tuple(coord), tuple(sorted(samples.split(","))), tuple(sorted(variants.split(","))),
            tuple(map(tuple, breaks)), n, tuple(sorted((int(k), tuple(map(tuple, v))) for k, v in va.items())))


while time.time() - t0 < budget:
    rlen = int(rng.integers(600, 6000))
    b0 = int(rng.integers(150, 1500))
    reg = synth.make_region(int(rng.integers(1 << 30)), "chrU", b0 + rlen + int(rng.integers(150, 1500)), b0, b0 + rlen)
    sites = max(2, int(rlen / float(np.exp(rng.uniform(np.log(15), np.log(300))))))
    try:
        synth.add_phased_variants(reg, int(rng.integers(1 << 30)), sites, int(rng.integers(1, 7)), frac_snv=float(rng.uniform(0.3, 0.95)),
                                  frac_del=float(rng.uniform(0.0, 0.4)), max_indel=int(rng.choice([2, 5, 12])), af_min=0.1, af_max=0.9)
    except ValueError:
        continue
    tag = f"{rlen} nt, {len(reg.variants)} sites, {len(reg.samples)} samples"
    rows = []
    for v in reg.variants:
        f = reg.vcf_fields(v)
        f[9:] = [g.replace("|", "/") for g in f[9:]]
        rows.append(f)
    # ---- reference ----
    ref_err = None
    try:
        region = mg._ref_region(reg)
        haps = [mg.Haplotype(mg.Sequence(region.sequence.sequence, True), region.coordinates, False, 0, True)]
        recs = []
        for f in rows:
            vr = mg.VariantRecord(True)
            vr.read_vcf_line(list(f), reg.samples, False)
            recs.append(vr)
        variants = mg.flatten_list([r.split() for r in recs])
        snvs, indels = mg.R_haps.classify_variants(variants)
        if snvs:
            haps.extend(mg.R_haps.compute_snvs_haplotype_unphased(snvs, reg.samples, region.sequence.sequence, region.coordinates, False, True))
        for indel in indels:
            if region.coordinates.startp <= indel.position < region.coordinates.stopp:
                haps.extend(mg.R_haps.create_indels_haplotype_unphased(indel, snvs, region, False, True))
        want = [sig(h.sequence.sequence, [h.coordinates.startp, h.coordinates.stopp, h.coordinates.start, h.coordinates.stop], h.samples, h.variants,
                    mg._posmap_breaks(h.posmap), len(h.posmap), h.variant_alleles) for h in haps]
    except (KeyError, ValueError, IndexError, SystemExit) as e:
        ref_err = e
    # ---- this package ----
    mregion = MRegion(MSequence(reg.sequence, True), MCoordinate(reg.contig, reg.bed_start, reg.bed_stop, synth.PADDING))
    mrecs = []
    for f in rows:
        r = MVariantRecord(True)
        r.read_vcf_line(list(f), reg.samples, False)
        mrecs.append(r)
    try:
        mh = [MHaplotype(MSequence(mregion.sequence.sequence, True), mregion.coordinates, False, 0, True)]
        mh = H.add_variants_unphased(mh, mregion, reg.samples, mrecs, False, True)
    except (KeyError, ValueError, IndexError, SystemExit) as e:
        if ref_err is None:
            raise AssertionError((tag, f"the restatement raised {type(e).__name__}: {e}; the reference did not"))
        n_err += 1
        continue
    if ref_err is not None:
        raise AssertionError((tag, f"the reference raised {type(ref_err).__name__}: {ref_err}; the restatement did not"))
    got = []
    for h in mh:
        pm = h.segments.full()
        c = h.coordinates
        got.append(sig(h.sequence.sequence, [c.startp, c.stopp, c.start, c.stop], h.samples, h.variants,
                       [[int(a), int(b)] for a, b in zip(*segments_from_posmap(pm))], len(pm), h.variant_alleles))
    n_whole = sum(1 for w in want if w[1] == want[0][1])
    assert len(got) == len(want), (tag, len(got), len(want))
    assert got[:n_whole] == want[:n_whole], (tag, "whole-region haplotypes")
    assert [g[1] for g in got] == [w[1] for w in want], (tag, "window order")
    assert sorted(got) == sorted(want), (tag, "window haplotypes")
    n_ok += 1
    if n_ok % 100 == 0:
        print(n_ok, "cases equal,", n_err, "refused by both;", tag, flush=True)
print(f"{n_ok} cases equal, {n_err} refused by both, in {time.time() - t0:.0f} s")
