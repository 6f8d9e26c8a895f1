#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (it needs /root/reference); the GPU box and the
test-suite only read the committed ``*.npz`` / ``*.json`` files.  The reference's
own tests pin nothing for this path, so these vectors --
inputs and the reference's outputs on them -- are what pins the oracle.

Absent third-party modules are replaced by throw-away stubs defined below
(logging, einops_exts' three helpers, MagicMock for trainer/vis packages).  No
reference source is copied: the reference modules are imported from where they lie.

Weights are NOT stored: every party regenerates them from (name, shape, seed)
with ``sgdm_amd.synth.tensor_from_seed`` (numpy legacy RandomState).  The
reference module's ordered state_dict manifest IS stored, so tests can check
that the oracle's and the product's parameter naming/shape/order match it.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))


# ----------------------------------------------------------------------------
# stubs for modules the container lacks (SURVEY.md 8(c) import recipe)
# ----------------------------------------------------------------------------
def install_stubs():
    import einops

    class _Logger:
        def __getattr__(self, _name):
            return lambda *a, **k: None

    loguru = types.ModuleType("loguru")
    loguru.logger = _Logger()
    sys.modules["loguru"] = loguru

    ee = types.ModuleType("einops_exts")
    ee.rearrange_many = lambda ts, pattern, **kw: tuple(einops.rearrange(t, pattern, **kw) for t in ts)
    ee.repeat_many = lambda ts, pattern, **kw: tuple(einops.repeat(t, pattern, **kw) for t in ts)
    ee.check_shape = lambda *a, **k: None
    sys.modules["einops_exts"] = ee

    for name in ("pytorch_lightning", "wandb", "torchvision", "torchvision.io", "torchvision.utils",
                 "torchvision.transforms", "seaborn", "distinctipy", "omegaconf", "omegaconf.listconfig",
                 "matplotlib", "matplotlib.pyplot", "cleanfid", "h5py"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock()


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def condition_obj(scale_type="imagen", layout_dims=None):
    d = AttrDict(scale_type=scale_type)
    for k, v in (layout_dims or {}).items():
        d[k] = AttrDict(layout_dim=v)
    return d


def manifest_of(module):
    params = dict(module.named_parameters())
    out = []
    for name, t in module.state_dict().items():
        if name in params:
            kind = "param" if params[name].requires_grad else "frozen"
        else:
            kind = "buffer"
        out.append([name, list(t.shape), kind])
    return out


def load_seeded(module, seed):
    from sgdm_amd.synth import tensor_from_seed
    sd = {k: tensor_from_seed(k, v.shape, seed) for k, v in module.state_dict().items()}
    module.load_state_dict(sd)
    return sd


def np_(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------
# configurations (small enough that the committed vectors stay tiny)
# ----------------------------------------------------------------------------
def unet_configs():
    """name -> (kind, ctor kwargs (reference names), synth spec)."""
    base = dict(in_channels=3, out_channels=3, num_res_blocks=2, channel_mult=[1, 2, 4],
                attention_resolutions=[4], num_heads=8, use_scale_shift_norm=True,
                use_checkpoint=False, use_fp16=False)
    cfgs = {}
    # C1-like: unet_fast / label K=10 (config/dynamic/unet_fast.yaml)
    cfgs["uf_label_c32_s16"] = ("unet_fast", dict(base, image_size=16, model_channels=32, dropout=0.1,
                                                  resblock_updown=True, cond_dim=10, condition_method="label"),
                                dict(batch=2, layout_dim=0))
    # C2-like: unet_fast / cluster K=5000 at reduced width
    cfgs["uf_cluster5000_c32_s16"] = ("unet_fast", dict(base, image_size=16, model_channels=32, dropout=0.1,
                                                        resblock_updown=True, cond_dim=5000,
                                                        condition_method="cluster"),
                                      dict(batch=2, layout_dim=0))
    # unet_fast / clusterlayout (layout concat on the self-attention UNet, openaimodel.py:933-939)
    cfgs["uf_clusterlayout_c32_s16"] = ("unet_fast", dict(base, image_size=16, model_channels=32, dropout=0.0,
                                                          resblock_updown=True, cond_dim=20,
                                                          condition_method="clusterlayout"),
                                        dict(batch=2, layout_dim=1))
    ca = dict(base, use_ca_block=True, transformer_depth=1, legacy=False, dropout=0.0,
              use_cls_token_as_pooled=True)
    # C4-like: unetca_fast / clusterlayout (LOST) cond_dim=100 ctx=32 L=1
    cfgs["ca_clusterlayout_c32_s16"] = ("unetca_fast", dict(ca, image_size=16, model_channels=32, cond_token_num=1,
                                                            cond_dim=100, context_dim=32,
                                                            condition_method="clusterlayout"),
                                        dict(batch=2, layout_dim=1))
    # C5-like: unetca_fast / stegoclusterlayout L=27
    cfgs["ca_stego_c32_s16"] = ("unetca_fast", dict(ca, image_size=16, model_channels=32, cond_token_num=1,
                                                    cond_dim=27, context_dim=32,
                                                    condition_method="stegoclusterlayout"),
                                dict(batch=2, layout_dim=27))
    # layout-only, cond_token_num=0 (openaimodel_ca.py:944-958)
    cfgs["ca_layout_c32_s16"] = ("unetca_fast", dict(ca, image_size=16, model_channels=32, cond_token_num=0,
                                                     cond_dim=0, context_dim=32, condition_method="layout"),
                                 dict(batch=2, layout_dim=21))
    # one full-width instance of each operator at the BASELINE shape (B=1)
    cfgs["uf_cluster5000_c128_s64"] = ("unet_fast", dict(base, image_size=64, model_channels=128, dropout=0.1,
                                                         resblock_updown=True, cond_dim=5000,
                                                         condition_method="cluster"),
                                       dict(batch=1, layout_dim=0))
    cfgs["ca_stego_c128_s64"] = ("unetca_fast", dict(ca, image_size=64, model_channels=128, cond_token_num=1,
                                                     cond_dim=27, context_dim=32,
                                                     condition_method="stegoclusterlayout"),
                                 dict(batch=1, layout_dim=27))
    return cfgs


def build_reference_unet(kind, kw, layout_dim, scale_type="imagen"):
    if kind == "unet_fast":
        from dynamic.diffusionmodules.openaimodel import UNetModel
    else:
        from dynamic.diffusionmodules.openaimodel_ca import UNetModel
    cm = kw["condition_method"]
    cond = condition_obj(scale_type, {cm: layout_dim} if layout_dim else {})
    return UNetModel(condition=cond, **kw).eval()


def gen_unet_vectors(out_dir, seed=23):
    from sgdm_amd.synth import synth_batch
    index = {}
    for name, (kind, kw, spec) in unet_configs().items():
        print("unet", name, flush=True)
        m = build_reference_unet(kind, kw, spec["layout_dim"])
        load_seeded(m, seed)
        B, S = spec["batch"], kw["image_size"]
        batch = synth_batch(kw["condition_method"], B, S, kw["cond_dim"], spec["layout_dim"], seed=seed)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, 3, S, S, generator=g)
        t = torch.tensor([500, 37][:B], dtype=torch.long) if B <= 2 else torch.randint(0, 1000, (B,), generator=g)
        cond, layout = batch.get("cond"), batch.get("layout")
        arrays = dict(x=np_(x), t=np_(t))
        if cond is not None:
            arrays["cond"] = np_(cond)
        if layout is not None:
            arrays["layout"] = np_(layout).astype(np.uint8)     # {0,1} masks: store compactly
        kwargs = dict(cond=cond, layout=layout)
        with torch.no_grad():
            # forward with an injected drop mask through tensor cond_drop_prob in {0,1}
            for tag, p in (("keep", torch.zeros(B)), ("drop", torch.ones(B)),
                           ("mixed", torch.tensor([0.0, 1.0][:B]))):
                eps, loss_in, logd = m(x, t, cond_drop_prob=p, **kwargs)
                assert loss_in == 0.0 and logd == {}
                arrays[f"eps_{tag}"] = np_(eps)
            if "s16" in name:
                for st in ("imagen", "cfg"):
                    m.condition["scale_type"] = st
                    for w in (0, 1, 2, 2.0, 1.5):
                        e = m.forward_with_cond_scale(x, t, cond_scale=w, **kwargs)
                        arrays[f"cfg_{st}_{w!r}"] = np_(e)
                m.condition["scale_type"] = "imagen"
            else:
                arrays["cfg_imagen_2.0"] = np_(m.forward_with_cond_scale(x, t, cond_scale=2.0, **kwargs))
        np.savez_compressed(os.path.join(out_dir, f"unet_{name}.npz"), **arrays)
        index[name] = dict(kind=kind, ctor=kw, layout_dim=spec["layout_dim"], batch=B, seed=seed,
                           manifest=manifest_of(m))
    with open(os.path.join(out_dir, "unet_index.json"), "w") as f:
        json.dump(index, f)


# ----------------------------------------------------------------------------
# per-block vectors (reference leaf modules, reduced width)
# ----------------------------------------------------------------------------
def gen_block_vectors(out_dir, seed=23):
    from dynamic.diffusionmodules import openaimodel as om, openaimodel_ca as omca
    from dynamic.crossattetion_lr import Attention_LR
    g = torch.Generator().manual_seed(seed + 2)
    arrays, meta = {}, {}

    def run(tag, mod, prefix, *inputs):
        mod.eval()
        load_seeded_prefixed(mod, prefix, seed)
        with torch.no_grad():
            y = mod(*inputs)
        for i, a in enumerate(inputs):
            if a is not None:
                arrays[f"{tag}.in{i}"] = np_(a)
        arrays[f"{tag}.out"] = np_(y)
        meta[tag] = dict(prefix=prefix, manifest=[[prefix + "." + k, list(v.shape)] for k, v in mod.state_dict().items()])

    def load_seeded_prefixed(mod, prefix, seed):
        from sgdm_amd.synth import tensor_from_seed
        mod.load_state_dict({k: tensor_from_seed(prefix + "." + k, v.shape, seed) for k, v in mod.state_dict().items()})

    emb = torch.randn(2, 96, generator=g)
    x64 = torch.randn(2, 64, 8, 8, generator=g)
    rb = dict(emb_channels=96, dropout=0.0, use_scale_shift_norm=True)
    run("res_plain", om.ResBlock(64, out_channels=64, **rb), "blk.res_plain", x64, emb)
    run("res_skip", om.ResBlock(64, out_channels=128, **rb), "blk.res_skip", x64, emb)
    run("res_down", om.ResBlock(64, out_channels=64, down=True, **rb), "blk.res_down", x64, emb)
    run("res_up", om.ResBlock(64, out_channels=64, up=True, **rb), "blk.res_up", x64, emb)
    run("res_noss", om.ResBlock(64, 96, 0.0, out_channels=64, use_scale_shift_norm=False), "blk.res_noss", x64, emb)
    x128 = torch.randn(2, 128, 8, 8, generator=g)
    run("attn_legacy", om.AttentionBlock(128, num_heads=4, num_head_channels=-1), "blk.attn_legacy", x128)
    ctx = torch.randn(2, 16, 32, generator=g)
    run("attn_lr", Attention_LR(query_dim=128, heads=4, dim_head=32, context_dim=32), "blk.attn_lr", x128, ctx)
    run("down_conv", omca.Downsample(64, True, dims=2, out_channels=64), "blk.down_conv", x64)
    run("up_conv", omca.Upsample(64, True, dims=2, out_channels=64), "blk.up_conv", x64)
    np.savez_compressed(os.path.join(out_dir, "blocks.npz"), **arrays)
    with open(os.path.join(out_dir, "blocks_index.json"), "w") as f:
        json.dump(meta, f)


# ----------------------------------------------------------------------------
# schedules, timestep embedding, samplers, train step, EMA, LR
# ----------------------------------------------------------------------------
MODEL_PARAMS = dict(  # config/model/ddpm.yaml:3-41 with device='cpu'
    given_betas=None, beta_schedule="linear", linear_start=0.0001, linear_end=0.02, cosine_s=8e-3,
    v_posterior=0.0, logvar_init=0.0, learn_logvar=False, clip_denoised=True, parameterization="eps",
    device="cpu", log_num_per_prog=10, loss_type="l2", tero_noise_sampling=False,
    tero_loss_weighting=False, sampling="native", num_timesteps=1000, sampling_imagelogger="ddim",
    num_timesteps_imagelogger=250, sampling_val="ddim", num_timesteps_val=50, sampling_test="ddim",
    num_timesteps_test=250, log_dir="/tmp", exp=None)


def sampling_kwargs(method, steps, eta=0.0):
    # dynamic_input/misc.py:128-141
    return dict(sampling_method=method, vis=None, num_timesteps=steps, ddim_eta=eta, log_num_per_prog=10,
                clip_denoised=True, dtp=1, temperature=1.0, noise_dropout=0, random_sample_condition=False,
                return_inter_dict=True, disable_tqdm=True)


def gen_diffusion_vectors(out_dir, seed=23):
    from diffusion.ddpm import LatentDiffusion
    from dynamic.diffusionmodules.util import timestep_embedding, make_ddim_timesteps, make_ddim_sampling_parameters
    from dynamic.ema import LitEma
    from diffusion_utils.lr_scheduler import LambdaLinearScheduler
    from sgdm_amd.synth import synth_batch

    arrays = {}
    diff = LatentDiffusion(**MODEL_PARAMS)
    for k, v in diff.sampler.state_dict().items():
        arrays["sched." + k] = np_(v)
    arrays["sched.lvlb_weights"] = np_(diff.sampler.lvlb_weights)
    for S in (10, 50, 250):
        steps = make_ddim_timesteps("uniform", S, 1000, verbose=False)
        for eta in (0.0, 1.0):
            sig, a, ap = make_ddim_sampling_parameters(diff.sampler.alphas_cumprod.cpu(), steps, eta, verbose=False)
            arrays[f"ddim{S}.eta{eta}.sigmas"] = np.asarray(sig, dtype=np.float64)
            arrays[f"ddim{S}.eta{eta}.alphas"] = np.asarray(a, dtype=np.float64)
            arrays[f"ddim{S}.eta{eta}.alphas_prev"] = np.asarray(ap, dtype=np.float64)
        arrays[f"ddim{S}.timesteps"] = steps.astype(np.int64)
    tt = torch.tensor([0, 1, 500, 999], dtype=torch.long)
    for dim in (32, 64, 128):
        arrays[f"temb.{dim}"] = np_(timestep_embedding(tt, dim))
    arrays["temb.t"] = np_(tt)
    # snapshot index lists (ddpm_sampler.py:219-220 ; ddim_plms_sampler.py:310-315)
    for total in (10, 50, 250, 1000):
        arrays[f"snap.{total}"] = torch.linspace(0, total, 10, dtype=torch.int).numpy().astype(np.int64)

    # LR multipliers (lr_scheduler.py:81-98 with config/optim/adamw.yaml)
    sch = LambdaLinearScheduler(warm_up_steps=[500], cycle_lengths=[10000000000000], f_start=[1.e-6],
                                f_max=[1.], f_min=[1.])
    ns = [0, 1, 250, 499, 500, 501, 10 ** 6]
    arrays["lr.n"] = np.asarray(ns, dtype=np.int64)
    arrays["lr.f"] = np.asarray([sch.schedule(n) for n in ns], dtype=np.float64)

    # ---------------- samplers on a tiny unet_fast (label, K=10) ----------------
    name = "uf_label_c32_s16"
    kind, kw, spec = unet_configs()[name]
    B, S = 2, kw["image_size"]
    batch = synth_batch("label", B, S, 10, seed=seed)
    for st in ("imagen",):
        m = build_reference_unet(kind, kw, 0, st)
        load_seeded(m, seed)
        diff.set_denoise_fn(m.forward, m.forward_with_cond_scale)
        dkw = dict(cond=batch["cond"], layout=None, cond_scale=2.0)
        # DDIM 10 steps (config C1 shape of run), eta = 0 and eta = 1
        for eta in (0.0, 1.0):
            torch.manual_seed(seed + 10)
            samples, inter = diff.p_sample_loop("ddim", (B, 3, S, S), sampling_kwargs("ddim", 10, eta),
                                                denoise_sample_fn_kwargs=dkw, condition_kwargs={})
            # replay the RNG stream: x_T, then per step uniform_(2B) inside the UNet, randn(shape)
            torch.manual_seed(seed + 10)
            xT = torch.randn(B, 3, S, S)
            zs = []
            for _ in range(10):
                torch.zeros(2 * B).float().uniform_(0, 1)
                zs.append(torch.randn(B, 3, S, S))
            tag = f"ddim10.eta{eta}"
            arrays[tag + ".x_T"] = np_(xT)
            arrays[tag + ".z"] = np_(torch.stack(zs))
            arrays[tag + ".samples_u8"] = np_(samples)
            arrays[tag + ".pred_x0_u8"] = np_(inter["pred_x0"])
            arrays[tag + ".x_inter"] = np_(inter["x_inter"])
        # native DDPM, 1000 steps (the BASELINE metric's sampler); noise by RNG replay
        print("native 1000-step trajectory ...", flush=True)
        torch.manual_seed(seed + 11)
        samples, inter = diff.p_sample_loop("native", (B, 3, S, S), sampling_kwargs("native", 1000),
                                            denoise_sample_fn_kwargs=dkw, condition_kwargs={})
        torch.manual_seed(seed + 11)
        xT = torch.randn(B, 3, S, S)
        keep = {}
        for i in reversed(range(1000)):
            torch.zeros(2 * B).float().uniform_(0, 1)
            z = torch.randn(B, 3, S, S)
            if i in (999, 500, 0):
                keep[i] = z
        arrays["native1000.rng_seed"] = np.asarray(seed + 11)
        arrays["native1000.x_T"] = np_(xT)
        for i, z in keep.items():
            arrays[f"native1000.z{i}"] = np_(z)
        arrays["native1000.samples_u8"] = np_(samples)
        arrays["native1000.pred_x0_u8"] = np_(inter["pred_x0"])
        arrays["native1000.x_inter"] = np_(inter["x_inter"])

    # ---------------- one training step (dropout=0 config so no mask RNG) ----------------
    for name in ("uf_clusterlayout_c32_s16", "ca_stego_c32_s16"):
        kind, kw, spec = unet_configs()[name]
        assert kw["dropout"] == 0.0
        m = build_reference_unet(kind, kw, spec["layout_dim"]).train()
        load_seeded(m, seed)
        diff.set_denoise_fn(m.forward, m.forward_with_cond_scale)
        diff.train()
        B, S = 4, kw["image_size"]
        batch = synth_batch(kw["condition_method"], B, S, kw["cond_dim"], spec["layout_dim"], seed=seed + 3)
        torch.manual_seed(seed + 12)
        loss, ld = diff.forward_tao(batch["image"], cond=batch["cond"].float(), layout=batch.get("layout"),
                                    cond_drop_prob=0.5)
        loss.backward()
        torch.manual_seed(seed + 12)
        t = torch.randint(0, 1000, (B,)).long()
        noise = torch.randn_like(batch["image"])
        mask = torch.zeros((B,)).float().uniform_(0, 1) < torch.full((B,), 0.5)
        tag = f"train.{name}"
        arrays[tag + ".t"] = np_(t)
        arrays[tag + ".noise"] = np_(noise)
        arrays[tag + ".drop_mask"] = np_(mask)
        arrays[tag + ".loss"] = np_(loss)
        arrays[tag + ".per_sample"] = np_(ld["train/epoch_stats_y"])
        assert torch.equal(ld["train/epoch_stats_x"], t)
        arrays[tag + ".loss_keys"] = np.asarray(sorted(ld.keys()))
        grads = {k: p.grad for k, p in m.named_parameters()}
        unused = sorted(k for k, p in m.named_parameters() if p.requires_grad and p.grad is None)
        arrays[tag + ".unused_params"] = np.asarray(unused)
        pick = ["input_blocks.0.0.weight", "out.2.weight", "out.0.weight", "time_embed.0.weight",
                "input_blocks.1.0.emb_layers.1.weight", "middle_block.0.in_layers.2.weight",
                "output_blocks.0.0.skip_connection.weight", "output_blocks.8.0.out_layers.3.bias"]
        if kind == "unet_fast":
            pick += ["mlp_cond.0.weight", "middle_block.1.qkv.weight", "middle_block.1.norm.weight",
                     "middle_block.1.proj_out.weight", "input_blocks.3.0.in_layers.2.weight"]
        else:
            pick += ["cond_mlp.0.weight", "to_cond_tokens.0.weight", "middle_block.1.to_q.weight",
                     "middle_block.1.to_kv.weight", "middle_block.1.null_kv", "middle_block.1.norm.gamma",
                     "middle_block.1.to_out.1.gamma", "middle_block.1.to_context.1.weight", "norm_cond.weight",
                     "input_blocks.3.0.op.weight", "output_blocks.2.2.conv.weight"]
        for k in pick:
            arrays[f"{tag}.grad.{k}"] = np_(grads[k])
        # global grad norm over all used params (cheap whole-model check)
        arrays[tag + ".grad_sqnorm"] = np.asarray(
            sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None))

    # ---------------- LitEma after 3 updates ----------------
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for p in lin.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    ema = LitEma(lin)
    arrays["ema.init"] = np.concatenate([np_(p).ravel() for p in lin.parameters()])
    deltas = []
    for _ in range(3):
        with torch.no_grad():
            for p in lin.parameters():
                d = torch.randn(p.shape, generator=g)
                deltas.append(np_(d).ravel())
                p.add_(d)
        ema(lin)
    arrays["ema.deltas"] = np.concatenate(deltas)
    arrays["ema.keys"] = np.asarray(list(ema.m_name2s_name.values()))
    arrays["ema.shadow"] = np.concatenate([np_(dict(ema.named_buffers())[s]).ravel()
                                           for s in ema.m_name2s_name.values()])
    arrays["ema.num_updates"] = np_(ema.num_updates)
    np.savez_compressed(os.path.join(out_dir, "diffusion.npz"), **arrays)


# ----------------------------------------------------------------------------
# condition plugin surface (dynamic_input/condition.py) -- key mapping vectors
# ----------------------------------------------------------------------------
def gen_condition_vectors(out_dir):
    from dynamic_input.condition import prepare_condition_kwargs, prepare_denoise_fn_kwargs_4sampling
    res = {}

    def mk(method, how=None, training=True):
        hp = AttrDict(cond_dim=5, condition_method=method, cond_drop_prob=0.1,
                      condition=AttrDict(clusterlayout=AttrDict(how=how), layout=AttrDict(how=how)))
        return AttrDict(hparams=hp, training=training, device="cpu")

    batch = {k: torch.ones(2, 3) for k in ("label", "cluster", "lostbboxmask", "segmask", "stegomask",
                                            "stego_attr", "id")}
    for method, how in ((None, None), ("label", None), ("cluster", None), ("clusterlayout", "lost"),
                        ("clusterlayout", "oracle"), ("clusterlayout", "stego"), ("layout", "lost"),
                        ("layout", "stego"), ("stegoclusterlayout", None)):
        for training in (True, False):
            b = {k: (v * (i + 1)) for i, (k, v) in enumerate(batch.items())}
            r = prepare_condition_kwargs(mk(method, how, training), b)
            src = {}
            for k, v in r.items():
                if torch.is_tensor(v):
                    src[k] = [name for name, bv in b.items() if torch.equal(bv.float(), v.float())][0]
                else:
                    src[k] = v
            res[f"{method}|{how}|{int(training)}"] = src
        b = {k: (v * (i + 1)) for i, (k, v) in enumerate(batch.items())}
        r = prepare_denoise_fn_kwargs_4sampling(mk(method, how, False), b,
                                                dict(random_sample_condition=False), cond_scale=2.0)
        res[f"sampling|{method}|{how}"] = sorted(r.keys())
    with open(os.path.join(out_dir, "condition_plugin.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def main():
    assert os.path.isdir(REF), "this script needs the reference checkout at /root/reference"
    install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    out_dir = HERE
    gen_condition_vectors(out_dir)
    gen_block_vectors(out_dir)
    gen_unet_vectors(out_dir)
    gen_diffusion_vectors(out_dir)
    print("done")


if __name__ == "__main__":
    main()
