#!/usr/bin/env python3
"""How much of the emit pass's run-to-run spread is the placement of the twelve column arrays: the same C3 search with the
columns reserved for n_rows + pad rows (different pads -> different blocks from the allocator and different distances between
the planes of `win`)."""
import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
import numpy as np
from crisprhawk_hip import synth
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import expand_on_device
reg = synth.make_region(1003, "chr22", 1_200_000, 100_000, 1_100_000)
synth.add_phased_variants(reg, 1003_1, 31000, 2504)
pam = PAM("NGG", False, True); pam.encode(0)
mm, pt = synth.cfd_tables()
for pad in [int(x) for x in (sys.argv[1:] or "0 0 1000 4096 12345 65536 100000 262144 1000003 0".split())]:
    os.environ[os.environ.get("PROBE_VAR", "HAWK_COLS_PAD")] = str(pad)
    ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
    v = ds.plan.view()
    t = [v.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False).timing for _ in range(8)]
    print(pad, "emit", " ".join(f"{x['v_emit_ms']:.3f}" for x in t), flush=True)
    ds.plan.close(); ds.close()
