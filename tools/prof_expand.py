import sys, time, cProfile, pstats
sys.path[:0] = ["/root/repo", "/root/repo/crispr-hawk_amd"]
from crisprhawk_hip import synth
from crisprhawk_hip.workload import expand_on_device
reg = synth.config_c3(2504, 31000, 1_000_000)
t = time.time()
pr = cProfile.Profile(); pr.enable()
ds, info, ms, kept = expand_on_device(reg, 3)
pr.disable()
print("total", time.time() - t, "kernel ms", ms)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
