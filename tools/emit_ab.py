#!/usr/bin/env python3
"""k_cs_emit variants on the SAME column buffers (the emit pass's time depends on where the allocator put them): alternate an
environment switch between searches of one view.   tools/emit_ab.py VAR val0 val1 [placements]"""
import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
import numpy as np
from crisprhawk_hip import synth
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device
var, v0, v1 = sys.argv[1], sys.argv[2], sys.argv[3]
reg = synth.make_region(1003, "chr22", 1_200_000, 100_000, 1_100_000)
synth.add_phased_variants(reg, 1003_1, 31000, 2504)
pam = PAM("NGG", False, True); pam.encode(0)
mm, pt = synth.cfd_tables()
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 3):
    os.environ["HAWK_COLS_PAD"] = str(rep * 5000)
    ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
    v = ds.plan.view()
    v.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False)
    out = {}
    for it in range(6):
        for val in (v0, v1):
            os.environ[var] = val
            out.setdefault(val, []).append(v.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False).timing["v_emit_ms"])
    print("placement", rep, {k: " ".join(f"{x:.3f}" for x in t) for k, t in out.items()}, flush=True)
    ds.plan.close(); ds.close()
