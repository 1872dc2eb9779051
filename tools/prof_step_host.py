#!/usr/bin/env python3
"""Host side of the bench step (GPU box): cProfile over 200 steps of rebuild_dictionary + search on the C3 plan."""
import cProfile
import pstats
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
from crisprhawk_hip import _lib, synth
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device

reg = synth.config_c3()
ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
pam = PAM("NGG", False, True)
pam.encode(0)
mm, pt = synth.cfd_tables()
plan = ds.plan
view = plan.view()


def step():
    plan.rebuild_dictionary()
    t = view.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False)
    c = (t.n_candidates, t.n_rows)
    t.close()
    return c


for _ in range(5):
    step()
t0 = time.perf_counter()
for _ in range(100):
    step()
print("ms/step", (time.perf_counter() - t0) * 10)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
