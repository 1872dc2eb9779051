#!/usr/bin/env python3
"""How fast a plain fill of C4-tile-sized planes runs on this box (five 2.5 GB hipMemsetAsync calls inside
hawk_hapset_create): the write-rate yardstick for k_hx_build.  Run under rocprofv3 --kernel-trace --stats."""
import ctypes as C
import sys
import time

import numpy as np

sys.path[:0] = ["/root/repo/crispr-hawk_amd"]
from crisprhawk_hip import _lib

L = _lib.lib()
ctx = _lib.context(0)
n_hap = int(sys.argv[1]) if len(sys.argv) > 1 else 5009
length = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_100
lens = np.full(n_hap, length, dtype=np.uint32)
for rep in range(3):
    h = C.c_void_p()
    t = time.perf_counter()
    _lib.check(L.hawk_hapset_create(ctx, n_hap, lens.ctypes.data_as(C.c_void_p), C.byref(h)), "create")
    L.hawk_sync(ctx)
    dt = time.perf_counter() - t
    print(f"rep {rep}: {dt*1e3:.2f} ms wall for {5 * n_hap * ((length + 31) // 32 + 5) * 4 / 1e9:.2f} GB", flush=True)
    L.hawk_hapset_destroy(h)
