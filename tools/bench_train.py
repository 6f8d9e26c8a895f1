#!/usr/bin/env python3
"""Train-step timing of the HIP path (forward + backward + AdamW + EMA) on synthetic data.
    python tools/bench_train.py [--batch 40] [--prec f16x3] [--steps 5] [--profile]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
import bench
from sgdm_amd.diffusion import LatentDiffusion
from sgdm_amd.ema import LitEma

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=40); ap.add_argument("--prec", default="f16x3")
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--profile", action="store_true"); ap.add_argument("--dropout", type=float, default=-1.0, help="override the model dropout")
a = ap.parse_args()
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["c2"]
m, sd, data = bench.build_model(wl, dev, a.prec, a.batch)
if a.dropout >= 0: m.dropout = a.dropout
m.train()
d = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS).train()
d.set_denoise_fn(m.forward, m.forward_with_cond_scale)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
ema = LitEma(m)
x = data["image"].to(dev); cond = data["cond"].to(dev)
def step():
    loss, _ = d.forward_tao(x, cond=cond, cond_drop_prob=0.1)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    ema(m)
    return loss
for _ in range(a.warmup): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(f"train step bs={a.batch} prec={a.prec}: {dt*1e3:.1f} ms  loss={l.item():.4f}")
if a.profile:
    from sgdm_amd import _lib as L
    eng = m._engines[(a.batch, 64, 64, L.PREC_BY_NAME[a.prec])]
    st = torch.cuda.current_stream().cuda_stream
    for nm, prog in (("forward", eng.prog), ("backward", eng.backward.prog)):
        agg = {}
        for tag, sym, ms, fl, nb in prog.run_profiled(st):
            agg[sym] = agg.get(sym, 0.0) + ms
        print(nm, {k: round(v, 2) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])}, "total", round(sum(agg.values()), 2))
