#!/usr/bin/env python3
"""Small-batch form of tools/stress_repeat.py (round 6: launches of at most 64 tiles of 128 columns run on the 32-column instance,
many of them with the balanced tail splitting their few tiles along K): N CFG evaluations of the same inputs per workload at batches
1 .. 6, every output bit compared with the first and with the evaluation of the same samples inside a batch of 8 (a few 1e-6: other tiles,
other K splits).   python tools/stress_small.py [--reps 200]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
for workload in ("c2", "c5"):
    wl = bench.WORKLOADS[workload]
    S = wl["image"]
    m, sd, data = bench.build_model(wl, torch.device("cuda"), "f16x3", 8)
    cond8 = data["cond"] if wl["kind"] == "unet_fast" else data["cond"].float()
    lay8 = data.get("layout")
    g = torch.Generator().manual_seed(97)
    x8 = torch.randn(8, 3, S, S, generator=g).cuda()
    t8 = torch.randint(0, 1000, (8,), generator=g).cuda()
    with torch.no_grad():
        ref8 = m.forward_with_cond_scale(x8, t8, cond=cond8.cuda(), layout=None if lay8 is None else lay8.cuda(), cond_scale=2.0).clone()
    for B in (1, 2, 3, 4, 6):
        kw = dict(cond=cond8[:B].cuda(), layout=None if lay8 is None else lay8[:B].cuda(), cond_scale=2.0)
        t0 = time.time()
        with torch.no_grad():
            first = m.forward_with_cond_scale(x8[:B], t8[:B], **kw).clone()
            bad = 0
            for i in range(a.reps - 1):
                y = m.forward_with_cond_scale(x8[:B], t8[:B], **kw)
                if not torch.equal(y, first):
                    bad += 1
        torch.cuda.synchronize()
        rel = float((first - ref8[:B]).abs().max() / ref8[:B].abs().max())
        print(f"{workload} batch {B} (UNet batch {2 * B}): {a.reps} evaluations, {bad} differ from the first, NaN {bool(torch.isnan(first).any())}, "
              f"max-rel vs the same samples in a batch of 8: {rel:.1e} ({time.time() - t0:.1f} s)", flush=True)
        for e in m._engines.values():
            e.check_health()
