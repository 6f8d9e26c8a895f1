#!/usr/bin/env python3
"""GroupNorm-backward HBM passes: the round-1 kernels (sgd_gn_bwd_reduce / sgd_gn_bwd_apply) against the row-stream reduce of
round 6 (sgd_gn_bwd_reduce_rows), interleaved in one process; algorithmic bytes / time.
    python tools/bench_gn_bwd.py [--n 80] [--rounds 7] [--reps 20] [--drop 0.1]"""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=80); ap.add_argument("--rounds", type=int, default=7); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--drop", type=float, default=0.0)
ap.add_argument("--shapes", nargs="+", default=["64,128", "64,256", "32,256", "32,512", "16,512", "16,1024", "32,128"])
a = ap.parse_args()
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
p = lambda t: C.c_void_p(t.data_ptr())


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for shp in a.shapes:
    h, c = (int(v) for v in shp.split(","))
    n = a.n
    x, gu, gres = (torch.randn(n, h, h, c, device="cuda") for _ in range(3))
    dst = torch.empty(n, h, h, c, device="cuda")
    ab = [torch.randn(n, c, device="cuda") for _ in range(5)]
    k = int(lib.sgd_gn_bwd_rows_chunks(n, h, h, c))
    S, P = torch.empty(n, c, 2, device="cuda"), torch.empty(n, max(1, k), c, 2, device="cuda")
    r0 = lambda: L.check(lib.sgd_gn_bwd_reduce(p(x), n, h, h, c, c, 0, p(ab[0]), p(ab[1]), 1, p(gu), c, 0, a.drop, 7, p(S), st), "r0")
    r1 = lambda: L.check(lib.sgd_gn_bwd_reduce_rows(p(x), n, h, h, c, c, 0, p(ab[0]), p(ab[1]), 1, p(gu), c, a.drop, 7, k, p(P), st), "r1")
    a0 = lambda: L.check(lib.sgd_gn_bwd_apply(p(x), n, h, h, c, c, 0, p(ab[0]), p(ab[1]), 1, p(gu), c, 0, a.drop, 7, p(ab[2]), p(ab[3]), p(ab[4]),
                                              p(gres), c, 0, p(dst), c, 0, 0, st), "a0")
    res = {nm: [] for nm in ("reduce", "reduce_rows", "apply")}
    for _ in range(a.rounds):
        for nm, fn in (("reduce", r0), ("reduce_rows", r1), ("apply", a0)):
            if k or not nm.endswith("rows"):
                res[nm].append(timed(fn, a.reps))
    el = 4.0 * n * h * h * c
    line = f"n={n} {h}x{h} c={c} ({el / 1e6:.0f} MB per tensor, {k} chunks):"
    for nm, byt in (("reduce", 2 * el), ("reduce_rows", 2 * el), ("apply", 4 * el)):
        if res[nm]:
            t = statistics.median(res[nm])
            line += f"  {nm} {t:.4f} ms ({byt / t / 1e9:.2f} TB/s)"
    print(line, flush=True)
