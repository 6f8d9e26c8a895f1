#!/usr/bin/env python3
"""BASELINE.json configs[0] (C1: cifar10 unet_fast ch64, 32x32, bs=8, 10-step DDIM) on the GPU, whole trajectory, eager and
hipGraph-captured; ~140 small launches per step, so per-launch overheads show here first.  (tools only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
import bench
from sgdm_amd.diffusion import LatentDiffusion
dev = torch.device("cuda", 0)
C1 = bench.C1
m1, _, d1 = bench.build_model(C1, dev, "f16x3", C1["batch"])
diff1 = LatentDiffusion(device=str(dev), **bench.MODEL_PARAMS)
diff1.set_denoise_fn(m1.forward, m1.forward_with_cond_scale)
k1 = dict(cond=d1["cond"].to(dev), layout=None, cond_scale=2.0)
skw = dict(sampling_method="ddim", num_timesteps=C1["ddim_steps"], ddim_eta=0.0, log_num_per_prog=10, clip_denoised=True, dtp=1,
           temperature=1.0, noise_dropout=0, random_sample_condition=False, return_inter_dict=True)
shape1 = (C1["batch"], 3, C1["image"], C1["image"])
with torch.no_grad():
    for name, g_on in (("eager", False), ("graph", True)):
        kw_ = dict(skw, hip_graph=g_on)
        run1 = lambda: diff1.p_sample_loop("ddim", shape1, kw_, denoise_sample_fn_kwargs=dict(k1), condition_kwargs={})
        run1()
        ts = []
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run1(); torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"{name}: {sorted(ts)[4]:.2f} ms per 10-step trajectory (min {min(ts):.2f})", flush=True)
