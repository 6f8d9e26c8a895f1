#!/usr/bin/env python3
"""Long form of tests/test_hip_fullsize.py::test_cfg_evaluation_is_bitwise_repeatable: N evaluations of the same inputs at the
benchmarked batch per workload and arithmetic mode, every output bit compared with the first.
    python tools/stress_repeat.py [--reps 400]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=400)
a = ap.parse_args()
for workload in ("c2", "c5", "c4"):
    wl = bench.WORKLOADS[workload]
    m, sd, data = bench.build_model(wl, torch.device("cuda"), "f16x3", wl["batch"])
    B, S = wl["batch"], wl["image"]
    cond = data["cond"] if wl["kind"] == "unet_fast" else data["cond"].float()
    layout = data.get("layout")
    g = torch.Generator().manual_seed(97)
    x = torch.randn(B, 3, S, S, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    kw = dict(cond=cond.cuda(), layout=None if layout is None else layout.cuda(), cond_scale=2.0)
    for prec in ("f16x3", "bf16x3", "f32"):
        reps = a.reps if prec != "f32" else max(10, a.reps // 8)
        m.hip_precision = prec
        t0 = time.time()
        with torch.no_grad():
            first = m.forward_with_cond_scale(x, t, **kw).clone()
            bad = 0
            for i in range(reps - 1):
                again = m.forward_with_cond_scale(x, t, **kw)
                if not torch.equal(again, first):
                    bad += 1
        torch.cuda.synchronize()
        print(f"{workload} {prec}: {reps} evaluations at UNet batch {2 * B}, {bad} differ from the first ({time.time() - t0:.1f} s)", flush=True)
    del m
    torch.cuda.empty_cache()
