#!/usr/bin/env python3
"""Micro-benchmark of the two attention cores (exact fp32 MFMA vs split precision) through the C-ABI at the shapes the
shipped configs launch:  python tools/bench_attn.py [--reps 50]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=50); a = ap.parse_args()
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
P = lambda t: C.c_void_p(t.data_ptr())
# (label, batch, heads, tq, tk, d, multi-query)
SHAPES = [("C2 16x16 legacy", 80, 6, 256, 256, 64, 0), ("C2 8x8 legacy", 80, 8, 64, 64, 64, 0),
          ("C5 16x16 [ctx|null|self]", 80, 8, 256, 273, 64, 1), ("C5 8x8", 80, 8, 64, 81, 64, 1),
          ("s64 32x32 legacy", 16, 8, 1024, 1024, 32, 0)]
for label, b, h, tq, tk, d, mq in SHAPES:
    if mq:
        q = torch.randn(b, tq, h * d, device="cuda"); kv = torch.randn(b, tk, 2 * d, device="cuda")
        args = (P(q), h * d, d, P(kv), C.c_void_p(kv.data_ptr() + 4 * d), 2 * d, 0)
    else:
        ch = h * d; q = torch.randn(b, tq, 3 * ch, device="cuda")
        args = (P(q), 3 * ch, 3 * d, C.c_void_p(q.data_ptr() + 4 * d), C.c_void_p(q.data_ptr() + 8 * d), 3 * ch, 3 * d)
    out = torch.empty(b, tq, h * d, device="cuda"); outs = {}
    flop = 4.0 * b * h * tq * tk * d
    line = f"{label:28s}"
    for name in ("sgd_attention", "sgd_attention_split"):
        fn = getattr(lib, name)
        call = lambda: L.check(fn(*args, b, h, tq, tk, d, d ** -0.5, P(out), h * d, None, st), name)
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps): call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        outs[name] = out.clone()
        line += f"  {name[4:]:16s} {ms * 1e3:8.1f} us {flop / ms / 1e9:7.1f} TF"
    err = float((outs["sgd_attention"] - outs["sgd_attention_split"]).abs().max() / outs["sgd_attention"].abs().max())
    print(line, f"  max-rel diff {err:.2e}")
