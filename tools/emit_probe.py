#!/usr/bin/env python3
"""The emit pass of the C3 cluster search (k_rows_pack + k_cs_emit_rows: packed 64-byte rows) over fresh reservations of the
table: every round releases the library's cached device memory, builds plan, view and table again and prints the emit times of
eight searches - the spread the twelve column arrays of round 3 showed (0.50 .. 0.66 ms) should be gone."""
import ctypes as C
import os
import sys
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
from crisprhawk_hip import _lib, synth
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device
reg = synth.config_c3()
pam = PAM("NGG", False, True)
pam.encode(0)
mm, pt = synth.cfd_tables()
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
    v = ds.plan.view()
    t = [v.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False).timing for _ in range(8)]
    print(rnd, "emit", " ".join(f"{x['v_emit_ms']:.3f}" for x in t), "| rows kernel", " ".join(f"{x['v_emit_rows_ms']:.3f}" for x in t[-3:]), flush=True)
    ds.plan.close()
    ds.close()
    _lib.lib().hawk_release_cached_memory(_lib.context())
