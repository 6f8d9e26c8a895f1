#!/usr/bin/env python3
"""Per-launch table of one UNet evaluation (HIP events around every launch).
    python tools/profile_layers.py [--workload c2] [--prec f16x3] [--batch 40]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
import bench
from sgdm_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2"); ap.add_argument("--prec", default="f16x3"); ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
wl = bench.WORKLOADS[a.workload]; B = a.batch or wl["batch"]
dev = torch.device("cuda", 0)
m, sd, data = bench.build_model(wl, dev, a.prec, B)
cond = data.get("cond"); cond = None if cond is None else (cond.to(dev) if wl["kind"] == "unet_fast" else cond.float().to(dev))
layout = data["layout"].to(dev) if "layout" in data else None
x = torch.randn(B, 3, wl["image"], wl["image"], device=dev); t = torch.full((B,), 500, device=dev, dtype=torch.long)
with torch.no_grad():
    for _ in range(2): m.forward_with_cond_scale(x, t, cond_scale=2.0, cond=cond, layout=layout)
eng = m._engines[(2 * B, wl["image"], wl["image"], L.PREC_BY_NAME[a.prec])]
st = torch.cuda.current_stream().cuda_stream
rows = None
for _ in range(a.reps):
    r = eng.prog.run_profiled(st)
    rows = r if rows is None else [(x0[0], x0[1], x0[2] + y[2], x0[3], x0[4]) for x0, y in zip(rows, r)]
tot = 0
print(f"{'tag':48s} {'kernel':18s} {'ms':>8s} {'TFLOP/s':>8s} {'GB/s':>8s}")
for tag, sym, ms, fl, nb in rows:
    ms /= a.reps; tot += ms
    print(f"{tag:48s} {sym:18s} {ms:8.3f} {fl/ms/1e9 if fl else 0:8.1f} {nb/ms/1e6 if nb else 0:8.0f}")
print("total ms", tot)
