#!/usr/bin/env python3
"""Achievable HBM bandwidth on this box: device-to-device copy (read + write) and a read-only
reduction over buffers far larger than L2 + MALL, timed with events."""
import torch
assert torch.cuda.is_available()
dev = torch.device("cuda", 0)
n = 4 << 30  # 4 GiB per buffer
a = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
b = torch.empty_like(a)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: b.copy_(a))
print(f"copy 4 GiB: {t*1e3:.2f} ms  -> {2*n/t/1e9:.0f} GB/s (read + write)")
af = a.view(torch.float32)
t = timed(lambda: af.sum())
print(f"read-only sum over 4 GiB: {t*1e3:.2f} ms -> {n/t/1e9:.0f} GB/s")
t = timed(lambda: b.fill_(3))
print(f"fill 4 GiB: {t*1e3:.2f} ms -> {n/t/1e9:.0f} GB/s (write)")
