"""Where the wall time of workload.expand_on_device goes on C3: library calls (wrapped) and the host code between them
(cProfile, cumulative top)."""
import cProfile
import pstats
import sys
import time
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
from crisprhawk_hip import _lib, synth  # noqa: E402
from crisprhawk_hip.workload import expand_on_device  # noqa: E402
reg = synth.config_c3()
ds, *_ = expand_on_device(reg, 3, keep_plan=True)
ds.plan.close(); ds.close()
L = _lib.lib()
T = {}


def wrap(name):
    f = getattr(L, name)

    def g(*a):
        t = time.perf_counter(); r = f(*a); T[name] = T.get(name, 0) + time.perf_counter() - t; return r
    return g


class Proxy:
    def __init__(self, L):
        self.L, self.c = L, {}

    def __getattr__(self, n):
        if n not in self.c:
            self.c[n] = wrap(n)
        return self.c[n]


_lib._lib = Proxy(L)
for rep in range(2):
    T.clear()
    t = time.perf_counter()
    ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
    print("total", round((time.perf_counter() - t) * 1e3, 2), "ms; library calls", round(sum(T.values()) * 1e3, 2), "ms")
    ds.plan.close(); ds.close()
for k, v in sorted(T.items(), key=lambda x: -x[1]):
    print(f"{k:34s} {v*1e3:8.2f} ms")
pr = cProfile.Profile()
pr.enable()
ds, info, ms, kept = expand_on_device(reg, 3, keep_plan=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
