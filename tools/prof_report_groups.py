#!/usr/bin/env python3
"""Where the report assembly of C3's 2.2 x 10^5 groups spends its time (GPU box): tools/prof_report_groups.py [--cprofile]
labels -> reports.group_columns -> reports.write_report_tsv, each of group_columns' parts timed on its own."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
import numpy as np

from crisprhawk_hip import reports, synth
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device, hap_labels

reg = synth.config_c3()
ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
pam = PAM("NGG", False, True)
pam.encode(0)
mm, pt = synth.cfd_tables()
tab = ds.plan.view().search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False)
tab.collapse(download_perm=False)
g = tab.export_groups()
spent = {}


def timed(name):
    fn = getattr(reports, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        spent[name] = spent.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(reports, name, wrap)


for name in ("_samples_raw", "_hapids_raw", "_variant_columns_raw", "_polish_rows_native", "_report_order", "_plain"):
    timed(name)
out = os.path.join(tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None), "r.tsv")
for rep in range(3):
    spent.clear()
    pr = cProfile.Profile() if "--cprofile" in sys.argv else None
    if pr:
        pr.enable()
    t = time.perf_counter()
    labels = hap_labels(reg.contig, reg.variants, ds, info, kept)
    t1 = time.perf_counter()
    cols, order, plain = reports.group_columns(g, labels, pam, reg.contig, "x", is_ref_hap=np.asarray(ds.is_ref, dtype=bool))
    t2 = time.perf_counter()
    n = reports.write_report_tsv(out, cols, order, plain)
    t3 = time.perf_counter()
    if pr:
        pr.disable()
    print("labels %.3f  group_columns %.3f  write %.3f  rows %d  bytes %d" % (t1 - t, t2 - t1, t3 - t2, len(order), n))
    print("   " + "  ".join("%s %.3f" % kv for kv in spent.items()) + "   (the carriers' joins run beside the variant columns)")
os.unlink(out)
if pr:
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
    pstats.Stats(pr).sort_stats("cumtime").print_stats("reports.py|workload.py", 30)
