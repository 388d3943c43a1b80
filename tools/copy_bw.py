#!/usr/bin/env python3
"""Reference points for the HBM roofline on this GPU: device-to-device copy and fill rates through torch."""
import torch

dev = torch.device("cuda", 0)
n = 8 * 4000 * 6000 * 3
a = torch.rand(n, device=dev)
b = torch.empty_like(a)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t = timeit(lambda: b.copy_(a))
print(f"copy  f32 {n*4/1e9:.2f} GB -> {t:.3f} ms, {2*n*4/t/1e6:.0f} GB/s (read + write)")
t = timeit(lambda: b.fill_(1.0))
print(f"fill  {t:.3f} ms, {n*4/t/1e6:.0f} GB/s (write only)")
t = timeit(lambda: torch.add(a, 1.0, out=b))
print(f"add1  {t:.3f} ms, {2*n*4/t/1e6:.0f} GB/s (read + write)")
t = timeit(lambda: a.sum())
print(f"sum   {t:.3f} ms, {n*4/t/1e6:.0f} GB/s (read only)")
