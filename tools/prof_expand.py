#!/usr/bin/env python3
"""cProfile of the host side of the C3 expansion (workload.expand_on_device): where end_to_end's first second goes."""
import cProfile
import pstats
import sys
import time

sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
from crisprhawk_hip import synth
from crisprhawk_hip.workload import expand_on_device

reg = synth.config_c3()
ds, *_ = expand_on_device(reg, 3)  # warm-up: library load, allocator
ds.close()
pr = cProfile.Profile()
pr.enable()
t = time.time()
ds, info, ms, kept = expand_on_device(reg, 3)
dt = time.time() - t
pr.disable()
print("expand_on_device wall s", round(dt, 3), "kernel ms", round(ms, 2))
pstats.Stats(pr).sort_stats("cumtime").print_stats(25)
