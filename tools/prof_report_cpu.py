#!/usr/bin/env python3
"""Times reports.report_from_groups on a synthetic group table of C3's size (223 k report rows, ~30 M members) without a
GPU: performance only (correctness is pinned by the g7 fixtures and the GPU tests)."""
import cProfile
import pstats
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
import numpy as np

from crisprhawk_hip import reports
from crisprhawk_hip.hapset import PosSegments
from crisprhawk_hip.pam import PAM

rng = np.random.default_rng(1)
NV, NS, L, W = 31000, 2504, 23, 43
H = 1 + 2 * NS
pos = np.sort(rng.choice(np.arange(2000, 1_000_000), size=NV, replace=False)).astype(np.int64)
af = np.exp(rng.uniform(np.log(1 / (2 * NS)), np.log(0.5), size=NV))
ref_b = rng.integers(0, 4, size=NV)
alt_b = (ref_b + rng.integers(1, 4, size=NV)) % 4
vid = [f"chr22-{p}-{'ACGT'[r]}/{'ACGT'[a]}" for p, r, a in zip(pos.tolist(), ref_b.tolist(), alt_b.tolist())]
t0 = time.time()
Gm = rng.random((NV, 2 * NS), dtype=np.float32) < af[:, None].astype(np.float32)
cols, sites = np.nonzero(Gm.T)
cnt = np.bincount(cols, minlength=2 * NS)
var_off = np.concatenate(([0, 0], np.cumsum(cnt)))
samples = ["REF"] + [f"S{(h - 1) // 2:04d}:{'1|0' if (h - 1) % 2 == 0 else '0|1'}" for h in range(1, H)]
ids = [f"hap_{h:08d}" for h in range(H)]
is_ref = np.zeros(H, bool); is_ref[0] = True
segs = [PosSegments.identity(1, 1_000_201)] * H
lab = reports.HapLabels(samples, ids, is_ref, var_off, sites.astype(np.int64), vid, af, segs)
# groups
v_sites, v_cols = np.nonzero(Gm)
c_off = np.concatenate(([0], np.cumsum(np.bincount(v_sites, minlength=NV))))
n_ref = 125_000
reps = 3
ng = n_ref + reps * NV
start = np.empty(ng, np.int64); strand = rng.integers(0, 2, size=ng).astype(np.uint8)
start[:n_ref] = np.sort(rng.integers(1000, 1_000_000, size=n_ref))
code = rng.integers(0, 4, size=(ng, W)).astype(np.uint8)
member_off = [0]; member_parts = []
sizes = np.ones(ng, np.int64)
k = n_ref
for r in range(reps):
    off_in = 2 + 7 * r
    start[k:k + NV] = pos - off_in
    code[k:k + NV, 10 + off_in] = alt_b
    sizes[k:k + NV] = np.diff(c_off)
    k += NV
one_hot = (np.uint64(1) << np.arange(W, dtype=np.uint64))
win = np.zeros((5, ng), np.uint64)
for b in range(4):
    win[b] = ((code == b).astype(np.uint64) * one_hot).sum(axis=1, dtype=np.uint64)
k = n_ref
for r in range(reps):
    win[4, k:k + NV] = np.uint64(1) << np.uint64(10 + 2 + 7 * r)
    k += NV
members = np.concatenate([np.zeros(n_ref, np.int64)] + [v_cols.astype(np.int64) + 1] * reps)
member_off = np.concatenate(([0], np.cumsum(sizes)))
G = reports.ReportGroups(20, 3, False, (start - 1 + 20).astype(np.uint32), strand, start, start + L, rng.random(ng), win,
                         rng.integers(5, 15, size=ng).astype(np.uint8), np.full(ng, 20, np.uint8), member_off, members)
pam = PAM("NGG", False, True); pam.encode(0)
print(f"synthetic groups: {ng} groups, {len(members)} members, built in {time.time() - t0:.1f}s")
pr = cProfile.Profile(); pr.enable()
t = time.time()
df = reports.report_from_groups(G, lab, pam, "chr22", "chr22:100000-1100000")
pr.disable()
print("rows", len(df), "assemble s", round(time.time() - t, 2))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
t = time.time()
txt = df.to_csv(sep="\t", index=False)
print("to_csv s", round(time.time() - t, 2), "MB", len(txt) / 1e6)
print(df.iloc[n_ref // 2 + 7].to_dict())
