#!/usr/bin/env python3
"""Yardsticks for the mixed read + write rate k_emit_list runs at: a device-to-device copy (as many bytes read as written)
and a 3 : 2 read : write kernel (torch.add of two tensors into a third is 2 : 1) on this box."""
import torch

n = 2_700_000_000 // 4
a = torch.empty(n, dtype=torch.int32, device="cuda").zero_()
b = torch.empty_like(a)
c = torch.empty_like(a)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]


def timed(f, reps=10):
    f()
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        f()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


ms = timed(lambda: b.copy_(a))
print(f"copy      : {ms:.3f} ms, {2 * n * 4 / ms / 1e9:.2f} TB/s (read + written)")
ms = timed(lambda: torch.add(a, b, out=c))
print(f"add 2r:1w : {ms:.3f} ms, {3 * n * 4 / ms / 1e9:.2f} TB/s")
ms = timed(lambda: a.zero_())
print(f"fill      : {ms:.3f} ms, {n * 4 / ms / 1e9:.2f} TB/s")
ms = timed(lambda: a.sum())
print(f"read (sum): {ms:.3f} ms, {n * 4 / ms / 1e9:.2f} TB/s")
