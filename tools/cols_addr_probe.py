#!/usr/bin/env python3
"""The emit pass's time next to the device addresses of the twelve column arrays, for several reservations in one process."""
import ctypes as C, os, sys
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
import numpy as np
from crisprhawk_hip import _lib, synth
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device
reg = synth.make_region(1003, "chr22", 1_200_000, 100_000, 1_100_000)
synth.add_phased_variants(reg, 1003_1, 31000, 2504)
pam = PAM("NGG", False, True); pam.encode(0)
mm, pt = synth.cfd_tables()
L = _lib.lib()
for pad in [int(x) for x in (sys.argv[1:] or "0 1000 4096 12345 65536 100000 262144 1000003 7 0 31 64".split())]:
    os.environ["HAWK_COLS_PAD"] = str(pad)
    ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
    v = ds.plan.view()
    tabs = [v.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False) for _ in range(5)]
    t = tabs[-1]
    ptrs = [C.c_void_p() for _ in range(8)]
    stride = C.c_uint64()
    L.hawk_table_device_columns(t._t, *[C.byref(p) for p in ptrs], C.byref(stride))
    em = np.mean([x.timing["v_emit_ms"] for x in tabs[1:]])
    print(f"pad {pad:8d} emit {em:.3f} cap {stride.value} " + " ".join(f"{(p.value or 0):x}" for p in ptrs), flush=True)
    ds.plan.close(); ds.close()
