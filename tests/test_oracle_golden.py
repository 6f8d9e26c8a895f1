"""Pin the CPU oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only (-m "not gpu")."""
import numpy as np
import pytest
import torch

from conftest import cfg_from_index, load_json, load_npz, max_rel, rel_l2
from oracle import diffusion_ref as D
from oracle import unet_ref as U
from sgdm_amd.synth import tensor_from_seed, weights_from_seed

INDEX = load_json("unet_index.json")
# fp32 summation-order noise floor of one UNet evaluation is ~2e-6
TOL_EVAL = 2e-5


def _weights(entry):
    return weights_from_seed(entry["manifest"], entry["seed"])


def _inputs(v):
    x = torch.from_numpy(v["x"])
    t = torch.from_numpy(v["t"])
    cond = torch.from_numpy(v["cond"]) if "cond" in v else None
    layout = torch.from_numpy(v["layout"]).float() if "layout" in v else None
    return x, t, cond, layout


@pytest.mark.parametrize("name", sorted(INDEX))
def test_manifest_matches_reference(name):
    entry = INDEX[name]
    mine = [[n, list(s), k] for n, s, k in U.param_manifest(cfg_from_index(entry))]
    assert mine == entry["manifest"]


@pytest.mark.parametrize("name", sorted(INDEX))
def test_unet_forward_matches_reference(name):
    entry = INDEX[name]
    cfg = cfg_from_index(entry)
    sd = _weights(entry)
    v = load_npz(f"unet_{name}.npz")
    x, t, cond, layout = _inputs(v)
    B = x.shape[0]
    if cond is not None and cfg["kind"] == "unetca_fast":
        cond = cond.float()                     # caller's .float() (dynamic_input/condition.py:46-47,80-81)
    masks = dict(keep=torch.zeros(B, dtype=torch.bool), drop=torch.ones(B, dtype=torch.bool),
                 mixed=torch.tensor([False, True][:B]))
    with torch.no_grad():
        for tag, m in masks.items():
            eps = U.unet_forward(cfg, sd, x, t, cond, layout, m)
            assert eps.shape == v[f"eps_{tag}"].shape
            assert max_rel(eps, v[f"eps_{tag}"]) < TOL_EVAL, tag
            assert rel_l2(eps, v[f"eps_{tag}"]) < TOL_EVAL, tag
    # the conditioning must matter, otherwise the comparison above is vacuous
    if cond is not None or layout is not None:
        assert rel_l2(v["eps_keep"], v["eps_drop"]) > 1e-3


@pytest.mark.parametrize("name", [n for n in sorted(INDEX) if "s16" in n])
def test_cfg_paths_match_reference(name):
    entry = INDEX[name]
    v = load_npz(f"unet_{name}.npz")
    x, t, cond, layout = _inputs(v)
    sd = _weights(entry)
    for st in ("imagen", "cfg"):
        cfg = dict(cfg_from_index(entry), scale_type=st)
        if cond is not None and cfg["kind"] == "unetca_fast":
            cond = cond.float()
        for w in (0, 1, 2, 2.0, 1.5):
            with torch.no_grad():
                e = U.forward_with_cond_scale(cfg, sd, x, t, w, cond, layout)
            assert max_rel(e, v[f"cfg_{st}_{w!r}"]) < TOL_EVAL, (st, w)


def test_cfg_full_width_instances():
    # (the *_s64_* entries are the shipped config/dynamic/*_s64.yaml plans: tests/golden/make_golden_s64.py)
    for name in ("uf_cluster5000_c128_s64", "ca_stego_c128_s64", "ca_clusterlayout_c128_s64", "ca_s64_c224", "ca_s64_c224_layout",
                 "uf_s64_c256"):
        entry = INDEX[name]
        cfg = cfg_from_index(entry)
        v = load_npz(f"unet_{name}.npz")
        x, t, cond, layout = _inputs(v)
        if cfg["kind"] == "unetca_fast" and cond is not None:
            cond = cond.float()
        with torch.no_grad():
            e = U.forward_with_cond_scale(cfg, _weights(entry), x, t, 2.0, cond, layout)
        assert max_rel(e, v["cfg_imagen_2.0"]) < TOL_EVAL


def test_blocks_match_reference():
    v = load_npz("blocks.npz")
    meta = load_json("blocks_index.json")
    cfg_ss = dict(use_scale_shift_norm=True)
    cfg_no = dict(use_scale_shift_norm=False)

    def sd_of(tag):
        return {n: tensor_from_seed(n, s, 23) for n, s in meta[tag]["manifest"]}

    def tin(tag, i):
        return torch.from_numpy(v[f"{tag}.in{i}"])

    with torch.no_grad():
        for tag, ud, cfg in (("res_plain", None, cfg_ss), ("res_skip", None, cfg_ss), ("res_down", "down", cfg_ss),
                             ("res_up", "up", cfg_ss), ("res_noss", None, cfg_no)):
            y = U.res_block(cfg, sd_of(tag), meta[tag]["prefix"], tin(tag, 0), tin(tag, 1), ud)
            assert max_rel(y, v[f"{tag}.out"]) < 5e-6, tag
        y = U.attention_block(sd_of("attn_legacy"), "blk.attn_legacy", tin("attn_legacy", 0), 4)
        assert max_rel(y, v["attn_legacy.out"]) < 5e-6
        y = U.attention_lr(sd_of("attn_lr"), "blk.attn_lr", tin("attn_lr", 0), tin("attn_lr", 1), 4)
        assert max_rel(y, v["attn_lr.out"]) < 5e-6
        ca = dict(kind="unetca_fast")
        y = U._run_block(ca, {k.replace("blk.down_conv", "b.0"): t for k, t in sd_of("down_conv").items()},
                         "b", [("down", 64, True)], tin("down_conv", 0), None, None, None)
        assert max_rel(y, v["down_conv.out"]) < 5e-6
        y = U._run_block(ca, {k.replace("blk.up_conv", "b.0"): t for k, t in sd_of("up_conv").items()},
                         "b", [("up", 64, True)], tin("up_conv", 0), None, None, None)
        assert max_rel(y, v["up_conv.out"]) < 5e-6


# ---------------------------------------------------------------- host-side bookkeeping: bit exact
def test_schedule_tables_bit_exact():
    v = load_npz("diffusion.npz")
    s = D.make_schedule()
    for k in D.SCHEDULE_KEYS:
        assert np.array_equal(s[k].numpy(), v["sched." + k]), k
    # probes quoted in SURVEY.md 8(a) A4
    assert s["betas"][0].item() == np.float32(9.9999997e-05)
    assert abs(s["alphas_cumprod"][999].item() - 7.3341245e-04) < 1e-10


def test_ddim_tables_bit_exact():
    v = load_npz("diffusion.npz")
    s = D.make_schedule()
    for S in (10, 50, 250):
        steps = D.make_ddim_timesteps(S)
        assert np.array_equal(steps, v[f"ddim{S}.timesteps"])
        for eta in (0.0, 1.0):
            tabs = D.make_ddim_tables(s["alphas_cumprod"], steps, eta)
            assert np.array_equal(np.asarray(tabs["ddim_sigmas"], dtype=np.float64), v[f"ddim{S}.eta{eta}.sigmas"])
            assert np.array_equal(np.asarray(tabs["ddim_alphas"], dtype=np.float64), v[f"ddim{S}.eta{eta}.alphas"])
            assert np.array_equal(np.asarray(tabs["ddim_alphas_prev"], dtype=np.float64),
                                  v[f"ddim{S}.eta{eta}.alphas_prev"])
    assert list(D.make_ddim_timesteps(10)) == [1, 101, 201, 301, 401, 501, 601, 701, 801, 901]


def test_timestep_embedding_and_snapshots():
    v = load_npz("diffusion.npz")
    t = torch.from_numpy(v["temb.t"])
    for dim in (32, 64, 128):
        assert np.array_equal(U.timestep_embedding(t, dim).numpy(), v[f"temb.{dim}"])
    for total in (10, 50, 250, 1000):
        assert D.snapshot_indices(total) == v[f"snap.{total}"].tolist()


def test_lr_lambda():
    v = load_npz("diffusion.npz")
    got = [D.lr_lambda_linear(int(n)) for n in v["lr.n"]]
    assert np.array_equal(np.asarray(got, dtype=np.float64), v["lr.f"])


def test_ema_three_updates():
    v = load_npz("diffusion.npz")
    shapes = [(3, 4), (3,), (2, 3), (2,)]
    sizes = [int(np.prod(s)) for s in shapes]
    init = np.split(v["ema.init"], np.cumsum(sizes)[:-1])
    params = {str(i): torch.from_numpy(a.copy()).reshape(s) for i, (a, s) in enumerate(zip(init, shapes))}
    shadow = {k: p.clone() for k, p in params.items()}
    deltas = np.split(v["ema.deltas"], np.cumsum(sizes * 3)[:-1])
    n = 0
    for step in range(3):
        for i, k in enumerate(params):
            params[k] = params[k] + torch.from_numpy(deltas[step * 4 + i]).reshape(shapes[i])
        n = D.ema_update(shadow, params, n)
    got = np.concatenate([shadow[k].numpy().ravel() for k in shadow])
    assert n == int(v["ema.num_updates"])
    assert np.allclose(got, v["ema.shadow"], rtol=0, atol=1e-6)
    assert list(v["ema.keys"]) == ["0weight", "0bias", "1weight", "1bias"]     # dot-stripped names (ema.py:18)


# ---------------------------------------------------------------- samplers
def _tiny_label_model():
    entry = INDEX["uf_label_c32_s16"]
    cfg = cfg_from_index(entry)
    sd = _weights(entry)
    from sgdm_amd.synth import synth_batch
    cond = synth_batch("label", 2, 16, 10, seed=23)["cond"]

    def eps_fn(x, t):
        with torch.no_grad():
            return U.forward_with_cond_scale(cfg, sd, x, t, 2.0, cond, None)
    return eps_fn


@pytest.mark.parametrize("eta", [0.0, 1.0])
def test_ddim10_trajectory(eta):
    v = load_npz("diffusion.npz")
    tag = f"ddim10.eta{eta}"
    z = torch.from_numpy(v[tag + ".z"])
    img, pred, inter, visited = D.ddim_sample(D.make_schedule(), _tiny_label_model(), torch.from_numpy(v[tag + ".x_T"]),
                                              lambda i: z[i], 10, eta=eta)
    assert [s for _, s in visited] == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
    assert [i for i, _ in visited] == list(range(9, -1, -1))
    assert pred.shape[0] == 9            # index 10 is never visited
    # short free-running trajectory: loose tolerance (plumbing check)
    assert rel_l2(inter, v[tag + ".x_inter"]) < 1e-3
    d = (D.to_uint8(img).int() - torch.from_numpy(v[tag + ".samples_u8"]).int()).abs()
    assert d.max() <= 1
    d = (D.to_uint8(pred).int() - torch.from_numpy(v[tag + ".pred_x0_u8"]).int()).abs()
    assert d.max() <= 1


def test_native_1000_step_trajectory():
    """The BASELINE metric's sampler, 1000 ancestral steps; noise by replaying the
    reference's RNG consumption order (x_T; per step uniform_(2B) then randn)."""
    v = load_npz("diffusion.npz")
    B, S = 2, 16
    torch.manual_seed(int(v["native1000.rng_seed"]))
    x_T = torch.randn(B, 3, S, S)
    assert torch.equal(x_T, torch.from_numpy(v["native1000.x_T"]))
    seen = {}

    def noises(i):
        torch.zeros(2 * B).float().uniform_(0, 1)
        z = torch.randn(B, 3, S, S)
        if i in (999, 500, 0):
            seen[i] = z
        return z

    eps_fn = _tiny_label_model()
    # the oracle draws z after the UNet call exactly like the reference: emulate by ordering
    sched = D.make_schedule()
    img = x_T
    snaps = D.snapshot_indices(1000)
    pred, inter = [], []
    for i in reversed(range(1000)):
        ts = torch.full((B,), i, dtype=torch.long)
        eps = eps_fn(img, ts)
        img, x0 = D.ddpm_step(sched, img, ts, eps, noises(i))
        if i in snaps:
            pred.append(x0)
            inter.append(img)
    for i in (999, 500, 0):
        assert torch.equal(seen[i], torch.from_numpy(v[f"native1000.z{i}"]))
    assert len(pred) == 9
    u8 = D.to_uint8(img).int()
    ref = torch.from_numpy(v["native1000.samples_u8"]).int()
    assert (u8 - ref).abs().max() <= 1
    assert ((u8 - ref) != 0).float().mean() < 1e-3
    assert rel_l2(torch.stack(inter), v["native1000.x_inter"]) < 1e-3


# ---------------------------------------------------------------- training step
@pytest.mark.parametrize("name", ["uf_clusterlayout_c32_s16", "ca_stego_c32_s16"])
def test_train_step_loss_and_grads(name):
    from sgdm_amd.synth import synth_batch
    v = load_npz("diffusion.npz")
    entry = INDEX[name]
    cfg = cfg_from_index(entry)
    sd = {k: t.clone().requires_grad_(kind == "param")
          for (k, _, kind), t in zip(entry["manifest"], _weights(entry).values())}
    tag = f"train.{name}"
    batch = synth_batch(cfg["condition_method"], 4, 16, cfg["cond_dim"], entry["layout_dim"], seed=23 + 3)
    t = torch.from_numpy(v[tag + ".t"])
    noise = torch.from_numpy(v[tag + ".noise"])
    mask = torch.from_numpy(v[tag + ".drop_mask"])
    fn = lambda xn, tt: U.unet_forward(cfg, sd, xn, tt, batch["cond"].float(), batch.get("layout"), mask)
    loss, per_sample, _, _ = D.p_losses(D.make_schedule(), fn, batch["image"], t, noise)
    assert abs(loss.item() - float(v[tag + ".loss"])) < 1e-5 * abs(float(v[tag + ".loss"]))
    assert max_rel(per_sample.detach(), v[tag + ".per_sample"]) < 1e-5
    loss.backward()
    unused = sorted(k for k, p in sd.items() if p.requires_grad and (p.grad is None or not p.grad.any()))
    assert unused == list(v[tag + ".unused_params"])
    sq = 0.0
    for k, p in sd.items():
        if p.grad is not None:
            sq += float((p.grad.double() ** 2).sum())
    assert abs(sq - float(v[tag + ".grad_sqnorm"])) < 1e-4 * float(v[tag + ".grad_sqnorm"])
    for key in v:
        if key.startswith(tag + ".grad."):
            pname = key[len(tag + ".grad."):]
            assert max_rel(sd[pname].grad, v[key]) < 5e-5, pname
    assert sorted(v[tag + ".loss_keys"].tolist()) == ["train/ddpm_loss", "train/epoch_stats_x",
                                                      "train/epoch_stats_y", "train/loss"]


def test_train_step_new_attention_order_matches_reference():
    """use_new_attention_order=True (QKVAttention, openaimodel.py:427-455): the oracle's loss and the gradients of every attention
    parameter against one training step of the reference module (tests/golden/make_golden_new_attention_order.py)"""
    v = load_npz("train_newattn.npz")
    u = load_npz("unet_uf_newattn_label_c32_s16.npz")
    entry = INDEX["uf_newattn_label_c32_s16"]
    cfg = cfg_from_index(entry)
    assert cfg["use_new_attention_order"] is True
    sd = {k: t.clone().requires_grad_(kind == "param")
          for (k, _, kind), t in zip(entry["manifest"], _weights(entry).values())}
    x, t, cond = torch.from_numpy(u["x"]), torch.from_numpy(u["t"]), torch.from_numpy(u["cond"])
    eps = U.unet_forward(cfg, sd, x, t, cond.float(), None, torch.tensor([False, True]))
    noise = torch.from_numpy(v["noise"])
    loss = ((noise - eps) ** 2).reshape(2, -1).mean(1).mean()
    assert abs(loss.item() - float(v["loss"])) < 1e-5 * abs(float(v["loss"]))
    loss.backward()
    keys = [k for k in v if k.startswith("g:")]
    assert len(keys) == 32
    for key in keys:
        ref = torch.from_numpy(v[key])
        if float(ref.abs().max()) > 1e-6:
            assert max_rel(sd[key[2:]].grad, ref) < 5e-5, key


def test_train_step_narrow_attention_heads_matches_reference():
    """config/dynamic/unet.yaml's attention layout (ds 2 and 4, 8- and 16-wide heads): the oracle's loss and gradients against one
    training step of the reference module (tests/golden/make_golden_narrow_heads.py)"""
    v = load_npz("train_heads8.npz")
    u = load_npz("unet_uf_heads8_label_c32_s16.npz")
    entry = INDEX["uf_heads8_label_c32_s16"]
    cfg = cfg_from_index(entry)
    sd = {k: t.clone().requires_grad_(kind == "param")
          for (k, _, kind), t in zip(entry["manifest"], _weights(entry).values())}
    x, t, cond = torch.from_numpy(u["x"]), torch.from_numpy(u["t"]), torch.from_numpy(u["cond"])
    eps = U.unet_forward(cfg, sd, x, t, cond.float(), None, torch.tensor([False, True]))
    noise = torch.from_numpy(v["noise"])
    loss = ((noise - eps) ** 2).reshape(2, -1).mean(1).mean()
    assert abs(loss.item() - float(v["loss"])) < 1e-5 * abs(float(v["loss"]))
    loss.backward()
    keys = [k for k in v if k.startswith("g:")]
    assert len(keys) == 59
    for key in keys:
        ref = torch.from_numpy(v[key])
        if float(ref.abs().max()) > 1e-6:
            assert max_rel(sd[key[2:]].grad, ref) < 5e-5, key


def test_train_step_without_scale_shift_norm_matches_reference():
    """use_scale_shift_norm=False (openaimodel.py:317-319): the oracle's loss and gradients against one training step of the
    reference module (tests/golden/make_golden_no_scale_shift.py)"""
    v = load_npz("train_noss.npz")
    u = load_npz("unet_uf_noss_label_c32_s16.npz")
    entry = INDEX["uf_noss_label_c32_s16"]
    cfg = cfg_from_index(entry)
    assert cfg["use_scale_shift_norm"] is False
    sd = {k: t.clone().requires_grad_(kind == "param")
          for (k, _, kind), t in zip(entry["manifest"], _weights(entry).values())}
    x, t, cond = torch.from_numpy(u["x"]), torch.from_numpy(u["t"]), torch.from_numpy(u["cond"])
    eps = U.unet_forward(cfg, sd, x, t, cond.float(), None, torch.tensor([False, True]))
    noise = torch.from_numpy(v["noise"])
    loss = ((noise - eps) ** 2).reshape(2, -1).mean(1).mean()
    assert abs(loss.item() - float(v["loss"])) < 1e-5 * abs(float(v["loss"]))
    loss.backward()
    keys = [k for k in v if k.startswith("g:")]
    assert len(keys) == 113
    for key in keys:
        ref = torch.from_numpy(v[key])
        if float(ref.abs().max()) > 1e-6:
            assert max_rel(sd[key[2:]].grad, ref) < 5e-5, key


def test_plms10_trajectory():
    """PLMS (ddim_plms_sampler.py:394-525): double UNet evaluation on the first step, Adams-Bashforth history"""
    v = load_npz("plms.npz")
    z = torch.from_numpy(v["plms10.z"])
    img, pred, inter, visited = D.plms_sample(D.make_schedule(), _tiny_label_model(), torch.from_numpy(v["plms10.x_T"]),
                                              lambda j: z[j], 10)
    assert [s for _, s in visited] == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
    assert pred.shape[0] == 9
    assert rel_l2(inter, v["plms10.x_inter"]) < 1e-3
    assert (D.to_uint8(img).int() - torch.from_numpy(v["plms10.samples_u8"]).int()).abs().max() <= 1
    assert (D.to_uint8(pred).int() - torch.from_numpy(v["plms10.pred_x0_u8"]).int()).abs().max() <= 1
