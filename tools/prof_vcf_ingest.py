#!/usr/bin/env python3
"""cProfile of the VCF-text ingest of the C3 workload (bench.time_vcf_ingest's expand_from_vcf half)."""
import cProfile
import pstats
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
import numpy as np
from crisprhawk_hip import synth
from crisprhawk_hip.readers import VcfBlock
from crisprhawk_hip.workload import expand_from_vcf

reg = synth.config_c3()
ns = len(reg.samples)
G = reg.gt_matrix
lut = np.array([ord("0"), ord("1")], np.uint8)
body = np.empty((len(reg.variants), 4 * ns), np.uint8)
body[:, 0::4] = lut[G[:, 0::2]]; body[:, 1::4] = ord("|"); body[:, 2::4] = lut[G[:, 1::2]]; body[:, 3::4] = ord("\t")
body[:, -1] = ord("\n")
fixed = [[reg.contig, str(v.pos), ".", v.ref, v.alt, ".", "PASS", f"AF={v.af:.6g}", "GT"] for v in reg.variants]
heads = [("\t".join(f) + "\t").encode() for f in fixed]
text = b"".join(h + body[i].tobytes() for i, h in enumerate(heads))
line_len = np.array([len(h) + 4 * ns for h in heads], dtype=np.int64)
line_off = np.concatenate(([0], np.cumsum(line_len)))
gt_off = line_off[:-1] + np.array([len(h) for h in heads])
blk = VcfBlock(np.frombuffer(text, dtype=np.uint8), line_off.astype(np.uint64), gt_off.astype(np.uint64), fixed)
for rep in range(2):
    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    ds, info, ms, kept, vt = expand_from_vcf(reg.sequence, reg.startp, reg.stopp, blk, reg.samples, 3, True, None)
    pr.disable()
    print("expand_from_vcf wall", round(time.time() - t, 3), ms)
    ds.close()
pstats.Stats(pr).sort_stats("cumtime").print_stats(16)
