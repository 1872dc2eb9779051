import sys, time, cProfile, pstats
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
import numpy as np
from crisprhawk_hip import synth, reports
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device, hap_labels
reg = synth.config_c3()
ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
pam = PAM("NGG", False, True); pam.encode(0)
mm, pt = synth.cfd_tables()
tab = ds.plan.view().search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False)
tab.collapse(download_perm=False)
g = tab.export_groups()
for rep in range(2):
    pr = cProfile.Profile(); pr.enable()
    t = time.time()
    labels = hap_labels(reg.contig, reg.variants, ds, info, kept)
    t1 = time.time()
    df = reports.report_from_groups(g, labels, pam, reg.contig, "x", is_ref_hap=np.asarray(ds.is_ref, dtype=bool))
    t2 = time.time()
    txt = reports.to_tsv(df)
    t3 = time.time()
    pr.disable()
    print("labels", round(t1 - t, 3), "assemble", round(t2 - t1, 3), "tsv", round(t3 - t2, 3), "rows", len(df), "bytes", len(txt))
    del txt
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
