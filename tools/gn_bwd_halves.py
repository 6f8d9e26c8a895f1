#!/usr/bin/env python3
"""Does the GroupNorm backward's second pass (apply) find the first pass's (reduce) operands in the 256 MB memory-side cache when
the batch is processed in image groups small enough to fit?  GroupNorm is per image, so reduce -> coefficients -> apply can run
group by group.  Times reduce + apply over the whole batch against the same work split into G image groups.
    python tools/gn_bwd_halves.py [--shapes 80,64,128 80,32,256 ...]      shape = n,hw,c"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="+", default=["80,64,128", "80,64,256", "80,32,256", "80,32,512", "80,16,512"])
ap.add_argument("--groups", nargs="+", type=int, default=[1, 2, 4, 8, 16])
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
lib = L.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t, off=0: C.c_void_p(t.data_ptr() + 4 * off)
for shp in a.shapes:
    n, hw, c = (int(v) for v in shp.split(","))
    g = torch.Generator(device="cuda").manual_seed(1)
    x, gu, gres = (torch.randn(n, hw, hw, c, device="cuda", generator=g) for _ in range(3))
    ca, cb = 1 + 0.1 * torch.randn(n, c, device="cuda", generator=g), 0.1 * torch.randn(n, c, device="cuda", generator=g)
    S = torch.zeros(n, c, 2, device="cuda")
    A, B, Cc = (0.1 * torch.randn(n, c, device="cuda", generator=g) for _ in range(3))
    dx = torch.empty_like(x)
    line = f"n={n} {hw}x{hw} c={c} ({x.numel() * 4 / 1e6:.0f} MB per tensor):"
    for G in a.groups:
        if n % G:
            continue
        m = n // G

        def run():
            for i in range(G):
                eo, co = i * m * hw * hw * c, i * m * c
                L.check(lib.sgd_gn_bwd_reduce(p(x, eo), m, hw, hw, c, c, 0, p(ca, co), p(cb, co), 1, p(gu, eo), c, 0, 0.0, 0, p(S, 2 * co), st), "reduce")
                L.check(lib.sgd_gn_bwd_apply(p(x, eo), m, hw, hw, c, c, 0, p(ca, co), p(cb, co), 1, p(gu, eo), c, 0, 0.0, 0, p(A, co), p(B, co), p(Cc, co),
                                             p(gres, eo), c, 0, p(dx, eo), c, 0, 0, st), "apply")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record(); torch.cuda.synchronize()
        line += f"  G={G}: {e0.elapsed_time(e1) / a.reps:.3f} ms"
    print(line, flush=True)
