#!/usr/bin/env python3
"""Per-tensor errors of the attention_ldm classes in grad mode against the gradients the reference's autograd recorded
(tests/golden/attention_ldm_train.npz): one line per case and arithmetic mode of the forward (the backward is exact fp32 in both).
    python tools/attention_ldm_train_errors.py  > profiles/r6_attention_ldm_train_errors.txt   (MI355X)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
from conftest import load_npz, max_rel  # noqa: E402
from test_hip_attention_ldm import CASES, _module  # noqa: E402

v = load_npz("attention_ldm_train.npz")
print("# max|a - b| / max|b| per tensor, HIP grad-mode call vs the reference's autograd (loss = sum(y * gy))")
for name in sorted(CASES):
    for prec in ("f32", "f16x3"):
        m = _module(name, v, prec)
        x = torch.from_numpy(v[name + ".x"]).cuda().requires_grad_(True)
        ctx = torch.from_numpy(v[name + ".context"]).cuda().requires_grad_(True)
        mask = torch.from_numpy(v[name + ".mask"]).cuda() if name + ".mask" in v else None
        gy = torch.from_numpy(v[name + ".gy"]).cuda()
        y = m(x, ctx, mask=mask)
        (y * gy).sum().backward()
        errs = {"y": max_rel(y.detach().cpu(), v[name + ".y"]), "g.x": max_rel(x.grad.cpu(), v[name + ".g.x"]),
                "g.context": max_rel(ctx.grad.cpu(), v[name + ".g.context"])}
        for k, prm in m.named_parameters():
            errs["g." + k] = max_rel(prm.grad.cpu(), v[name + ".g." + k])
        print(f"{name:22s} forward {prec:6s} worst {max(errs.values()):.2e}  " + "  ".join(f"{k} {e:.1e}" for k, e in errs.items()))
