#!/usr/bin/env python3
"""Long form of test_training_step_is_bitwise_repeatable: the same training step (dropout on, fixed seeds) N times at the
benchmarked batch; loss and every gradient compared bit for bit with the first.   python tools/stress_repeat_train.py [--reps 20]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
import bench
from sgdm_amd.diffusion import LatentDiffusion
from sgdm_amd.synth import synth_batch

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--batch", type=int, default=80)
a = ap.parse_args()
for workload in ("c2", "c5"):
    wl = bench.WORKLOADS[workload]
    B = a.batch
    model, _, data = bench.build_model(wl, torch.device("cuda"), "f16x3", B)
    model.train()
    diff = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS).train()
    diff.set_denoise_fn(model.forward, model.forward_with_cond_scale)
    g = torch.Generator().manual_seed(13)
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    noise = torch.randn(B, 3, 64, 64, generator=g).cuda()
    mask = (torch.rand(B, generator=g) < 0.2).cuda()
    x0 = data["image"].cuda()
    cond = data["cond"].cuda() if wl["kind"] == "unet_fast" else data["cond"].float().cuda()
    kw = dict(cond=cond, cond_drop_prob=0.2, cond_drop_mask=mask)
    if "layout" in data: kw["layout"] = data["layout"].cuda()
    ref, bad, t0 = None, 0, time.time()
    for rep in range(a.reps):
        torch.manual_seed(4321)
        for p in model.parameters(): p.grad = None
        loss, _ = diff.p_losses(x0, t, noise, **kw)
        loss.backward()
        torch.cuda.synchronize()
        cur = [loss.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
        if ref is None: ref = cur
        elif not all(torch.equal(u, v) for u, v in zip(cur, ref)): bad += 1
    print(f"{workload}: {a.reps} training steps at batch {B} (dropout {model.dropout}), {len(ref) - 1} gradient tensors, {bad} steps differ from the first ({time.time() - t0:.1f} s)", flush=True)
    del model
    torch.cuda.empty_cache()
