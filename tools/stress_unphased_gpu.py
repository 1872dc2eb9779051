#!/usr/bin/env python3
"""GPU box: the unphased path - VCF records -> haplotypes.add_variants_unphased -> device search -> resolve_guide - against
the guide lists the REFERENCE produced for the same records (tools/campaign_unphased_fixtures.py, build container).

    python tools/stress_unphased_gpu.py tools/_campaign/unphased.json.gz
"""
import gzip
import json
import sys

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd", "/root/repo/tests"]
from crisprhawk_hip import haplotypes as H
from crisprhawk_hip.haplotype import Haplotype
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.search_guides import search
from crisprhawk_hip.sequence import Sequence
from test_host_objects import _unphased_inputs

cases = json.load(gzip.open(sys.argv[1], "rt"))
n_ok = 0
for k, fx in enumerate(cases):
    reg, region, recs = _unphased_inputs(fx)
    assert reg.sequence == fx["region_seq"], k
    haps = [Haplotype(Sequence(region.sequence.sequence, True), region.coordinates, False, 0, True)]
    haps = H.add_variants_unphased(haps, region, reg.samples, recs, False, True)
    for i, h in enumerate(haps):
        h.id = f"hap_{i:08d}"
    pam = PAM(fx["pam"], fx["right"], True)
    pam.encode(0)
    guides = search(pam, region, haps, None, fx["guidelen"], fx["right"], True, False, 0, True)
    got = sorted([g.start, g.stop, g.strand, g.sequence, ",".join(sorted(g.samples.split(","))), g.right] for g in guides)
    want = sorted([g[0], g[1], g[2], g[3], ",".join(sorted(g[6].split(","))), g[5]] for g in fx["guides"])
    assert got == want, (k, fx["synth"], fx["pam"], len(got), len(want))
    n_ok += 1
print(f"{n_ok} unphased cases: guide lists equal to the reference's")
