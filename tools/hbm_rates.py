#!/usr/bin/env python3
"""practical HBM rates of this device through torch element-wise kernels (the yardstick for the HBM-bound passes)"""
import torch
n = 80*4096*128
a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda"); c = torch.empty(n, device="cuda")
def t(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: torch.add(a, b, out=c)); print(f"add 2R+1W {3*n*4/ms/1e9:.2f} TB/s ({ms:.4f} ms)")
ms = t(lambda: c.copy_(a)); print(f"copy 1R+1W {2*n*4/ms/1e9:.2f} TB/s ({ms:.4f} ms)")
ms = t(lambda: torch.sum(a)); print(f"sum 1R {n*4/ms/1e9:.2f} TB/s ({ms:.4f} ms)")
ms = t(lambda: torch.dot(a, b)); print(f"dot 2R {2*n*4/ms/1e9:.2f} TB/s ({ms:.4f} ms)")
ms = t(lambda: c.fill_(1.0)); print(f"fill 1W {n*4/ms/1e9:.2f} TB/s ({ms:.4f} ms)")
