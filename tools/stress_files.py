#!/usr/bin/env python3
"""Randomised campaign, part 4 (GPU box; minutes): tools/stress_files.py [seconds] [seed]
FASTA + BED + VCF files (plain, gzip, BGZF; random line widths; several regions per contig, so that the VCF index's
random access is exercised) -> readers -> device genotype parser -> expansion must give the planes, labels and guide
table of the in-memory path; pipeline.search_files must write the report the in-memory path writes."""
import os
import sys
import tempfile
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd", "/root/repo/tests"]
import numpy as np

from crisprhawk_hip import readers, synth
from crisprhawk_hip.expand import HaplotypeBuildError
from crisprhawk_hip.workload import expand_from_vcf, expand_on_device
from oracle import oracle as ora

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
n_ok = n_skip = 0
bits, bitsrc, _, _ = ora.pam_encode("NGG")
mm, pt = synth.cfd_tables()
with tempfile.TemporaryDirectory() as tmp:
    while time.time() - t0 < budget:
        rlen = int(rng.integers(10_000, 80_000))
        b0 = int(rng.integers(2000, 20_000))
        reg = synth.make_region(int(rng.integers(1 << 30)), "chrF", rlen + 30_000, b0, b0 + rlen)
        sites = max(5, int(rlen / float(np.exp(rng.uniform(np.log(30), np.log(800))))))
        try:
            synth.add_phased_variants(reg, int(rng.integers(1 << 30)), sites, int(rng.integers(2, 14)), frac_snv=float(rng.uniform(0.4, 0.9)),
                                      frac_del=float(rng.uniform(0.03, 0.3)), max_indel=int(rng.choice([3, 8])), af_min=0.1, af_max=0.8)
        except ValueError:
            continue
        mode = str(rng.choice(["plain", "gzip", "bgzf"]))
        fa, bed = os.path.join(tmp, "r.fa"), os.path.join(tmp, "r.bed")
        vcf = os.path.join(tmp, "v.vcf" if mode == "plain" else "v.vcf.gz")
        for f in (fa, fa + ".fai", vcf):
            if os.path.exists(f):
                os.remove(f)
        readers.write_fasta(fa, reg.contig, reg.contig_seq, int(rng.choice([60, 61, 80, 1000])))
        with open(bed, "w") as f:
            f.write(f"{reg.contig}\t{reg.bed_start}\t{reg.bed_stop}\n")
        rows = [reg.vcf_fields(v) for v in reg.variants]
        if mode == "bgzf":
            readers.write_vcf(vcf, reg.contig, reg.samples, rows, compress="bgzf")
        else:
            readers.write_vcf(vcf, reg.contig, reg.samples, rows, mode == "gzip")
        coord = readers.Bed(bed, synth.PADDING)[0]
        seq = readers.Fasta(fa).fetch(coord).sequence
        assert seq == reg.sequence
        v = readers.VCF(vcf)
        blk = v.fetch_block(coord)
        tag = f"files: {rlen} nt, {len(reg.variants)} sites, {len(reg.samples)} samples, {mode}"
        try:
            ds0, info0, _, kept0 = expand_on_device(reg, 3)
        except (KeyError, HaplotypeBuildError) as e:
            try:
                expand_from_vcf(seq, coord.start, coord.stop, blk, v.samples, 3, v.phased)
            except type(e):
                n_skip += 1
                continue
            raise AssertionError((tag, "in-memory path refused, file path did not"))
        ds1, info1, ms, kept1, vt = expand_from_vcf(seq, coord.start, coord.stop, blk, v.samples, 3, v.phased)
        assert kept0 == kept1 and [i.samples for i in info0] == [i.samples for i in info1], tag
        assert all(np.array_equal(a.variant_idx, b.variant_idx) for a, b in zip(info0, info1)), tag
        assert np.array_equal(ds0.planes(), ds1.planes()), tag
        t0_ = ds0.search(bits, bitsrc, 3, 20, False, mm, pt)
        t1_ = ds1.search(bits, bitsrc, 3, 20, False, mm, pt)
        for col in ("hap", "pos", "strand", "start", "stop", "flags", "win"):
            assert np.array_equal(getattr(t0_, col), getattr(t1_, col)), (tag, col)
        assert np.array_equal(np.nan_to_num(t0_.cfdon, nan=-1), np.nan_to_num(t1_.cfdon, nan=-1)), tag
        ds0.close(); ds1.close()
        n_ok += 1
        print(tag, "rows", t0_.n_rows, "ok", flush=True)
print(f"{n_ok} file sets, {n_skip} inputs refused, in {time.time() - t0:.0f} s: all equal")
