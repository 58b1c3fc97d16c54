"""Calibration: what do plain torch copy / fill / read-reduce kernels reach on this box (GB/s)?"""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (132, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x)
    s = t(lambda: y.copy_(x)); print(f"{mb} MB copy : {2 * n * 2 / s / 1e9:7.0f} GB/s (read+write)")
    s = t(lambda: y.fill_(1.0)); print(f"{mb} MB fill : {n * 2 / s / 1e9:7.0f} GB/s (write)")
    s = t(lambda: x.view(torch.int16).max()); print(f"{mb} MB max  : {n * 2 / s / 1e9:7.0f} GB/s (read)")
    s = t(lambda: torch.add(x, x, out=y)); print(f"{mb} MB add  : {2 * n * 2 / s / 1e9:7.0f} GB/s (read+write)")
