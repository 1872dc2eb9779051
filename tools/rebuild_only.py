#!/usr/bin/env python3
"""The cluster dictionary of the C3 plan built twelve times and nothing else (GPU box; for rocprofv3 / PMC runs of the k_cl_* kernels)."""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
from crisprhawk_hip import synth
from crisprhawk_hip.workload import expand_on_device
reg = synth.config_c3()
ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
for _ in range(12):
    ds.plan.rebuild_dictionary()
print(ds.plan.cluster_stats())
