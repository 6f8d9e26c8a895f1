#!/usr/bin/env python3
"""One forward evaluation and one training step of every shipped config/dynamic/*.yaml plan at its full width on the HIP path,
each against the oracle (forward eps; loss and every parameter gradient): which plans run, and how close.
    python tools/audit_plans.py [--prec f16x3] [--plans unet,unetca,...]  > profiles/r6_audit_plans.txt      (MI355X)
The plans (reference config/dynamic/): unet_fast, unet_fast_s64, unetca_fast, unetca_fast_s64 come from tests/golden/unet_index.json
(ctor kwargs recorded from the yaml files); unet (attention at ds 2 and 4 with 32 heads: 8 and 16 channels per head) and unetca
(attention at ds 4 and 2) are those entries with the yaml's differences applied."""
import argparse
import copy
import json
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))


def plans():
    idx = json.load(open(os.path.join(ROOT, "tests", "golden", "unet_index.json")))
    out = {}
    out["unet_fast"] = copy.deepcopy(idx["uf_cluster5000_c128_s64"])
    out["unet_fast_s64"] = copy.deepcopy(idx["uf_s64_c256"])
    out["unetca_fast"] = copy.deepcopy(idx["ca_stego_c128_s64"])
    out["unetca_fast_s64"] = copy.deepcopy(idx["ca_s64_c224"])
    u = copy.deepcopy(idx["uf_cluster5000_c128_s64"])
    u["ctor"].update(attention_resolutions=[2, 4], num_heads=32)                 # config/dynamic/unet.yaml:8,11
    out["unet"] = u
    c = copy.deepcopy(idx["ca_stego_c128_s64"])
    c["ctor"].update(attention_resolutions=[4, 2])                               # config/dynamic/unetca.yaml:10
    out["unetca"] = c
    # the data configs' other image sizes (config/data/: 32 x 32 -- cifar, in32 --, 128 x 128 -- ffhq128) on the two benchmark plans
    for base in ("unet_fast", "unetca_fast"):
        for size in (32, 128):
            e = copy.deepcopy(out[base])
            e["ctor"]["image_size"] = size
            out[f"{base}@{size}"] = e
    for e in out.values():
        e.pop("manifest", None)
    return out


def sample_audit(m, kw, e, batch, B, S):
    """a short trajectory of every sampler of LatentDiffusion.sampler_list on this plan: finite uint8 images, healthy engines"""
    import bench
    from sgdm_amd.diffusion import LatentDiffusion
    m.eval()
    d = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS).eval()
    d.set_denoise_fn(m.forward, m.forward_with_cond_scale)
    cnd = batch["cond"].float().cuda() if batch.get("cond") is not None else None
    lay = batch["layout"].cuda() if batch.get("layout") is not None else None
    done = []
    for method, steps in (("ddim", 10), ("plms", 10), ("native", 1000)):
        sk = dict(sampling_method=method, num_timesteps=steps, ddim_eta=0.0, log_num_per_prog=10, clip_denoised=True, dtp=1,
                  temperature=1.0, noise_dropout=0, random_sample_condition=False, return_inter_dict=True)
        img, inter = d.p_sample_loop(method, (B, 3, S, S), sk,
                                     denoise_sample_fn_kwargs=dict(cond=cnd if e["kind"] != "unet_fast" else batch.get("cond").cuda(),
                                                                   layout=lay, cond_scale=2.0))
        assert img.dtype == torch.uint8 and tuple(img.shape) == (B, 3, S, S), (method, img.dtype, img.shape)
        sd = float(img.float().std())
        assert sd > 1.0, (method, sd)
        done.append(f"{method}{steps}")
    return " ".join(done)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="f16x3")
    ap.add_argument("--plans", default="")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--sample", action="store_true", help="also run a short trajectory of every sampler on each plan")
    a = ap.parse_args()
    import bench
    from conftest import cfg_from_index, max_rel
    from oracle import diffusion_ref as D
    from oracle import unet_ref as U
    from sgdm_amd.diffusion import LatentDiffusion
    from sgdm_amd.synth import synth_batch, weights_from_seed
    from sgdm_amd.unet import UNetModel, UNetModelCA
    from test_hip_unet import AttrDict
    P = plans()
    names = [n for n in a.plans.split(",") if n] or sorted(P)
    print(f"# plan: forward eps vs oracle | train step: loss, worst parameter gradient (max|a-b|/max|b|) vs the oracle's autograd; prec {a.prec}, B {a.batch}")
    for name in names:
        e = P[name]
        kw = dict(e["ctor"])
        t0 = time.time()
        try:
            cond = AttrDict(scale_type="imagen")
            if e["layout_dim"]:
                cond[kw["condition_method"]] = AttrDict(layout_dim=e["layout_dim"])
            cls = UNetModel if e["kind"] == "unet_fast" else UNetModelCA
            m = cls(condition=cond, **kw)
            params = dict(m.named_parameters())
            manifest = [[k, list(v.shape), ("param" if params[k].requires_grad else "frozen") if k in params else "buffer"]
                        for k, v in m.state_dict().items()]
            cfg = cfg_from_index(e)
            assert [k for k, _, _ in manifest] == [k for k, _, _ in U.param_manifest(cfg)], "manifest differs from the oracle's"
            w = weights_from_seed(manifest, 23)
            m.load_state_dict(w)
            m = m.cuda().eval()
            m.hip_precision = a.prec
            m.dropout = 0.0
            B, S = a.batch, int(kw["image_size"])
            batch = synth_batch(kw["condition_method"], B, S, kw["cond_dim"], e["layout_dim"], seed=5)
            g = torch.Generator().manual_seed(5)
            t = torch.randint(0, 1000, (B,), generator=g)
            noise = torch.randn(B, 3, S, S, generator=g)
            mask = torch.tensor([False, True, False, False][:B])
            cnd = batch["cond"].float() if batch.get("cond") is not None else None
            lay = batch.get("layout")
            with torch.no_grad():
                eps = m(batch["image"].cuda(), t.cuda(), cond=cnd.cuda() if cnd is not None else None,
                        layout=lay.cuda() if lay is not None else None, cond_drop_prob=0.0)[0]
                ref = U.unet_forward(cfg, w, batch["image"], t, cnd, lay, None)
            ferr = max_rel(eps.cpu(), ref)
            m.train()
            d = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS).train()
            d.set_denoise_fn(m.forward, m.forward_with_cond_scale)
            loss, _ = d.p_losses(batch["image"].cuda(), t.cuda(), noise.cuda(), cond=cnd.cuda() if cnd is not None else None,
                                 layout=lay.cuda() if lay is not None else None, cond_drop_prob=0.5, cond_drop_mask=mask.cuda())
            loss.backward()
            sd = {k: tt.clone().requires_grad_(kind == "param") for (k, _, kind), tt in zip(manifest, w.values())}
            l, _, _, _ = D.p_losses(D.make_schedule(), lambda xn, tt: U.unet_forward(cfg, sd, xn, tt, cnd, lay, mask),
                                    batch["image"], t, noise)
            l.backward()
            worst, n = ("", 0.0), 0
            for k, p in m.named_parameters():
                if p.requires_grad and sd[k].grad is not None and float(sd[k].grad.abs().max()) > 1e-6:
                    err = max_rel(p.grad.cpu(), sd[k].grad)
                    n += 1
                    if err > worst[1]:
                        worst = (k, err)
            smp = ("  samplers: " + sample_audit(m, kw, e, batch, B, S)) if a.sample else ""
            print(f"{name:16s} OK  forward {ferr:.2e} | loss rel {abs(loss.item() - l.item()) / abs(l.item()):.1e}  worst of {n} gradients "
                  f"{worst[1]:.2e} ({worst[0]}){smp}  [{time.time() - t0:.0f} s]", flush=True)
        except Exception as ex:                                           # the audit's purpose: report, go on
            tb = traceback.format_exc().strip().splitlines()
            print(f"{name:16s} FAILED  {type(ex).__name__}: {str(ex)[:300]}   at {tb[-3].strip() if len(tb) >= 3 else ''}", flush=True)


if __name__ == "__main__":
    main()
