#!/usr/bin/env python3
"""Per-launch table of one training step's backward (or, --forward, forward) program (HIP events around every launch),
slowest first.
    python tools/profile_train_layers.py [--workload c2|c5|c4] [--batch 80] [--top 40] [--match wgrad] [--forward] [--dropout 0.0] [--lib libsgdm_hip_base.so]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
import bench
from sgdm_amd import _lib as L
from sgdm_amd.diffusion import LatentDiffusion

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=80); ap.add_argument("--top", type=int, default=40); ap.add_argument("--match", default="")
ap.add_argument("--prec", default="f16x3"); ap.add_argument("--forward", action="store_true")
ap.add_argument("--dropout", type=float, default=-1.0, help="override the model dropout")
ap.add_argument("--workload", default="c2")
ap.add_argument("--lib", default="", help="another build of the library (path relative to sgdm_amd/lib/): A/B on one box")
a = ap.parse_args()
if a.lib: L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), a.lib)
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[a.workload]
m, sd, data = bench.build_model(wl, dev, a.prec, a.batch)
if a.dropout >= 0: m.dropout = a.dropout
m.train()
d = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS).train()
d.set_denoise_fn(m.forward, m.forward_with_cond_scale)
x = data["image"].to(dev); cond = data["cond"].to(dev) if wl["kind"] == "unet_fast" else data["cond"].float().to(dev)
layout = data["layout"].to(dev) if "layout" in data else None
for _ in range(2):
    loss, _ = d.forward_tao(x, cond=cond, layout=layout, cond_drop_prob=0.1)
    for p in m.parameters(): p.grad = None
    loss.backward()
eng = m._engines[(a.batch, 64, 64, L.PREC_BY_NAME[a.prec])]
st = torch.cuda.current_stream().cuda_stream
rows = None
for _ in range(3):
    r = (eng.prog if a.forward else eng.backward.prog).run_profiled(st)
    rows = r if rows is None else [(x0[0], x0[1], x0[2] + y[2], x0[3], x0[4]) for x0, y in zip(rows, r)]
rows = [(t, s, ms / 3, fl) for t, s, ms, fl, nb in rows if a.match in t or a.match in s]
print("total ms", sum(r[2] for r in rows))
for t, s, ms, fl in sorted(rows, key=lambda r: -r[2])[:a.top]:
    print(f"{t:52s} {s:24s} {ms:8.3f} ms {fl / ms / 1e9 if fl else 0:8.1f} TF")
