import sys, time, cProfile, pstats
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
from crisprhawk_hip import synth, reports
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device, row_labels
reg = synth.config_c3()
ds, info, ms, kept = expand_on_device(reg, 3)
pam = PAM("NGG", False, True); pam.encode(0)
mm, pt = synth.cfd_tables()
tab = ds.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False, collapse=True)
inp = reports.ReportInput.from_table(tab)
labels = row_labels(reg, ds, info, kept)
pr = cProfile.Profile(); pr.enable()
t = time.time()
df = reports.report_frame(inp, labels, pam, reg.contig, "x")
pr.disable()
print("rows", len(df), "s", time.time() - t)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
