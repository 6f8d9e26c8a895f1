#!/usr/bin/env python3
"""The fold of the weight-gradient slabs (sgd_wgrad_reduce: dw[co, ci, tap] = scale * sum_k slabs[k][tap][co][ci]) at the slab counts
and layer shapes of a C2 training step (UNet batch 80), two builds of the library interleaved in one process (--old PATH: e.g. the
library before the 16-byte form); checks that both give the same bits.
    python tools/bench_wgrad_reduce.py --old self-guided-diffusion-models_amd/sgdm_amd/lib/libsgdm_hip_old.so"""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L
from sgdm_amd.train import wgrad_ksplit
ap = argparse.ArgumentParser()
ap.add_argument("--old", default=""); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
new = L.load()
libs = [("new", new)]
if a.old:
    old = C.CDLL(a.old)
    old.sgd_wgrad_reduce.restype = C.c_int32
    old.sgd_wgrad_reduce.argtypes = new.sgd_wgrad_reduce.argtypes
    libs.insert(0, ("old", old))
st = torch.cuda.current_stream().cuda_stream
p = lambda t: C.c_void_p(t.data_ptr())
# (taps, cout, cin, rows): the ResBlock convs and skips of the C2 plan at UNet batch 80
n = 80
shapes = [(9, 128, 128, n * 4096), (9, 128, 256, n * 4096), (9, 128, 384, n * 4096), (9, 256, 256, n * 1024), (9, 256, 512, n * 1024),
          (9, 256, 768, n * 1024), (9, 512, 512, n * 256), (9, 512, 1024, n * 256), (9, 512, 512, n * 64),
          (1, 128, 256, n * 4096), (1, 256, 512, n * 1024), (1, 512, 1024, n * 256)]


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


tot = {nm: 0.0 for nm, _ in libs}
for taps, co, ci, rows in shapes:
    ks = wgrad_ksplit(taps, co, ci, rows)
    slabs = torch.randn(ks, taps, co, ci, device="cuda")
    outs, res = {}, {nm: [] for nm, _ in libs}
    for nm, lib in libs:
        outs[nm] = torch.full((co, ci, taps), float("nan"), device="cuda")
    for _ in range(a.rounds):
        for nm, lib in libs:
            fn = lambda lib=lib, nm=nm: L.check(lib.sgd_wgrad_reduce(p(slabs), ks, taps, co, ci, p(outs[nm]), 0, 0.5, st), "reduce")
            res[nm].append(timed(fn, a.reps))
    byt = slabs.numel() * 4 + co * ci * taps * 4
    line = f"taps={taps} cout={co} cin={ci} ksplit={ks} ({byt / 1e6:.0f} MB):"
    for nm, _ in libs:
        t = statistics.median(res[nm]); tot[nm] += t
        line += f"  {nm} {t * 1e3:.1f} us ({byt / t / 1e9:.2f} TB/s)"
    if a.old:
        line += f"  bits equal: {bool(torch.equal(outs['old'], outs['new']))}"
    print(line, flush=True)
print("sum over the listed shapes:", {k: round(v, 4) for k, v in tot.items()}, "ms")
