#!/usr/bin/env python3
"""sgd_conv3_narrow_in back to back at the UNet-batch-160 stem shape, one or more builds interleaved:
    python tools/bench_narrow.py lib1.so [lib2.so ...]
(each library only has to export sgd_conv3_narrow_in / _parts: `hipcc -shared csrc/narrow.hip` is enough)"""
import ctypes as C
import sys

import torch

n, h, w, cin, cout = 160, 64, 64, 3, 128
x = torch.randn(n, h, w, cin, device="cuda")
wt = torch.randn(cout, cin, 3, 3, device="cuda")
b = torch.randn(cout, device="cuda")
y = torch.empty(n, h, w, cout, device="cuda")
libs = [C.CDLL(p) for p in sys.argv[1:]]
parts = libs[0].sgd_conv3_narrow_in_parts(h, w)
st = torch.empty(n, parts, 2, cout, device="cuda")
s = torch.cuda.current_stream().cuda_stream
P = lambda t: C.c_void_p(t.data_ptr())
for rep in range(3):
    for path, lib in zip(sys.argv[1:], libs):
        f = lambda: lib.sgd_conv3_narrow_in(P(x), P(wt), P(b), P(y), P(st), n, h, w, cin, cout, cout, 0, C.c_void_p(s))
        for _ in range(5):
            assert f() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print(f"{path.split('/')[-1]}: {ms * 1e3:.1f} us  ({(x.numel() + y.numel()) * 4 / ms / 1e9:.2f} TB/s of algorithmic bytes)")
