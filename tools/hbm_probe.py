#!/usr/bin/env python3
"""Development aid: write-only / read-only / copy HBM rates through torch elementwise kernels (gpurun only)."""
import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (32, 154, 308, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
    w = t(lambda: x.fill_(1.0)); r = t(lambda: x.sum()); c = t(lambda: y.copy_(x)); rmw = t(lambda: x.add_(1.0))
    print(f"{mb:5d} MB  fill {mb/1024/w/1.024**0*1.073741824:7.2f} GB/s... write {n*4/w/1e12:5.2f} TB/s  read {n*4/r/1e12:5.2f} TB/s  copy(r+w) {2*n*4/c/1e12:5.2f} TB/s  rmw {2*n*4/rmw/1e12:5.2f} TB/s")
