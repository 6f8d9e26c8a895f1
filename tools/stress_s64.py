#!/usr/bin/env python3
"""Repeat one CFG evaluation of the shipped unetca_fast_s64 width (ch=224: BN=32 tiles, padded heads, balanced tail on small
launches) and report evaluations that differ from the first / contain NaN.  (tools only; chasing an intermittent NaN)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sgdm_amd.synth import synth_batch, weights_from_seed
from sgdm_amd.unet import UNetModelCA


class AttrDict(dict):
    __getattr__ = dict.__getitem__


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
kw = dict(image_size=64, in_channels=3, out_channels=3, model_channels=224, num_res_blocks=2, channel_mult=[1, 2, 3, 4],
          attention_resolutions=[4, 8], num_heads=32, num_head_channels=-1, use_scale_shift_norm=True, use_ca_block=True, legacy=False,
          dropout=0.0, cond_token_num=1, cond_dim=27, context_dim=32, use_cls_token_as_pooled=True, condition_method="stegoclusterlayout")
m = UNetModelCA(condition=AttrDict(scale_type="imagen", stegoclusterlayout=AttrDict(layout_dim=27)), **kw)
manifest = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
m.load_state_dict(weights_from_seed(manifest, 23))
m = m.cuda().eval(); m.hip_precision = prec
batch = synth_batch("stegoclusterlayout", 1, 64, 27, 27, seed=3)
cond, layout = batch["cond"].float().cuda(), batch["layout"].cuda()
g = torch.Generator().manual_seed(4)
x, t = torch.randn(1, 3, 64, 64, generator=g).cuda(), torch.tensor([321]).cuda()
bad = 0
with torch.no_grad():
    first = m.forward_with_cond_scale(x, t, cond_scale=2.0, cond=cond, layout=layout).clone()
    print("first has nan:", bool(torch.isnan(first).any()))
    for i in range(n):
        y = m.forward_with_cond_scale(x, t, cond_scale=2.0, cond=cond, layout=layout)
        if i % 25 == 0:
            torch.cuda.synchronize(); print("iter", i, flush=True)
        if not torch.equal(y, first):
            bad += 1
            d = (y - first).abs()
            print(i, "differs: nan", int(torch.isnan(y).sum()), "max", float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else None, flush=True)
print("bad", bad, "of", n)
