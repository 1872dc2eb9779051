import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
import numpy as np
from crisprhawk_hip import synth, workload
from crisprhawk_hip.workload import expand_on_device
reg = synth.config_c3()
ds, *_ = expand_on_device(reg, 3); ds.close()
# monkeypatch timers
import ctypes as C
from crisprhawk_hip import _lib
L = _lib.lib()
T = {}
def wrap(name):
    f = getattr(L, name)
    def g(*a):
        t = time.perf_counter(); r = f(*a); T[name] = T.get(name, 0) + time.perf_counter() - t; return r
    return g
class Proxy:
    def __init__(self, L): self.L = L; self.c = {}
    def __getattr__(self, n):
        if n not in self.c: self.c[n] = wrap(n)
        return self.c[n]
_lib._lib = Proxy(L)
t = time.perf_counter()
ds, info, ms, kept = expand_on_device(reg, 3)
print("total", round(time.perf_counter() - t, 3))
for k, v in sorted(T.items(), key=lambda x: -x[1]): print(f"{k:34s} {v*1e3:8.2f} ms")
