"""Sampling loops of the HIP drop-in (LatentDiffusion.p_sample_loop through the reference API) against
the trajectories recorded from the reference.  GPU only.

Per SURVEY Appendix C: deterministic DDIM with random weights is an expansive map, so short free-running
trajectories are a plumbing check at a loose tolerance; the gate proper is per evaluation with teacher
forcing.  Native DDPM trajectories are well conditioned and are compared on uint8 output."""
import pytest
import torch

from conftest import load_npz, max_rel, rel_l2
from test_hip_unet import build_model

pytestmark = pytest.mark.gpu


def _diffusion(model):
    import bench
    from sgdm_amd.diffusion import LatentDiffusion
    d = LatentDiffusion(device="cuda", **bench.MODEL_PARAMS)
    d.set_denoise_fn(model.forward, model.forward_with_cond_scale)
    return d


def _skw(method, steps, eta=0.0):
    # dynamic_input/misc.py:128-141
    return dict(sampling_method=method, vis=None, num_timesteps=steps, ddim_eta=eta, log_num_per_prog=10,
                clip_denoised=True, dtp=1, temperature=1.0, noise_dropout=0, random_sample_condition=False,
                return_inter_dict=True, disable_tqdm=True)


def _cond():
    from sgdm_amd.synth import synth_batch
    return synth_batch("label", 2, 16, 10, seed=23)["cond"].cuda()


@pytest.mark.parametrize("eta", [0.0, 1.0])
def test_ddim10_trajectory_vs_reference(eta):
    v = load_npz("diffusion.npz")
    tag = f"ddim10.eta{eta}"
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    z = torch.from_numpy(v[tag + ".z"])
    samples, inter = d.p_sample_loop("ddim", (2, 3, 16, 16), _skw("ddim", 10, eta),
                                     denoise_sample_fn_kwargs=dict(cond=_cond(), layout=None, cond_scale=2.0),
                                     condition_kwargs={}, x_T=torch.from_numpy(v[tag + ".x_T"]),
                                     noise_fn=lambda i: z[i])
    assert samples.dtype == torch.uint8 and tuple(samples.shape) == (2, 3, 16, 16)
    assert tuple(inter["pred_x0"].shape) == (9, 2, 3, 16, 16)          # index 10 never visited (Appendix B)
    assert rel_l2(inter["x_inter"].cpu(), v[tag + ".x_inter"]) < 1e-3
    assert (samples.cpu().int() - torch.from_numpy(v[tag + ".samples_u8"]).int()).abs().max() <= 1
    assert (inter["pred_x0"].cpu().int() - torch.from_numpy(v[tag + ".pred_x0_u8"]).int()).abs().max() <= 1


def test_ddim_teacher_forced_steps_vs_oracle():
    """the gate proper: every step fed the ORACLE's x_t; eps-combined update compared per step"""
    from oracle import diffusion_ref as D
    from oracle import unet_ref as U
    from conftest import cfg_from_index
    from sgdm_amd.synth import weights_from_seed
    v = load_npz("diffusion.npz")
    tag = "ddim10.eta1.0"
    m, entry = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    cfg, sd = cfg_from_index(entry), weights_from_seed(entry["manifest"], entry["seed"])
    cond = _cond()
    sched = D.make_schedule()
    steps = D.make_ddim_timesteps(10)
    tabs = D.make_ddim_tables(sched["alphas_cumprod"], steps, 1.0)
    z = torch.from_numpy(v[tag + ".z"])
    x = torch.from_numpy(v[tag + ".x_T"])
    ds = d.sampler_list["ddim"]
    skw = dict(_skw("ddim", 10, 1.0), alphas_cumprod=d.sampler.alphas_cumprod)
    ds.make_schedule(skw)
    import ctypes as C
    from sgdm_amd import _lib as L
    from sgdm_amd.diffusion import _StepRunner
    runner = _StepRunner(d.denoise_sample_fn, dict(cond=cond, layout=None, cond_scale=2.0))
    for i, step in enumerate(reversed(steps.tolist())):
        index = 10 - i - 1
        ts = torch.full((2,), int(step), dtype=torch.long)
        with torch.no_grad():
            eps_ref = U.forward_with_cond_scale(cfg, sd, x, ts, 2.0, cond.cpu(), None)
        x_ref, x0_ref = D.ddim_step(tabs, index, x, eps_ref, z[i])
        xd = x.cuda()
        eps, mode, w, bb, cc = runner.eps(xd, ts.cuda())
        coef = (C.c_float * 4)(float(ds.ddim_sqrt_one_minus_alphas[index]), float(ds.ddim_alphas[index]),
                               float(ds.ddim_alphas_prev[index]), float(ds.ddim_sigmas[index]))
        out, x0 = torch.empty_like(xd), torch.empty_like(xd)
        zd = z[i].cuda()
        L.check(L.load().sgd_ddim_step(C.c_void_p(xd.data_ptr()), C.c_void_p(eps.data_ptr()), C.c_void_p(zd.data_ptr()),
                                       mode, w, coef, 1.0, 1, bb, cc, 256, C.c_void_p(out.data_ptr()),
                                       C.c_void_p(x0.data_ptr()), torch.cuda.current_stream().cuda_stream), "ddim")
        assert rel_l2(out.cpu(), x_ref) < 1e-4, index           # north_star: 1e-4 per evaluation
        assert rel_l2(x0.cpu(), x0_ref) < 1e-4, index
        x = x_ref                                                # teacher forcing


def test_native_1000_step_trajectory_vs_reference():
    """the BASELINE metric's sampler end to end; z by replaying the reference's RNG consumption order"""
    v = load_npz("diffusion.npz")
    B, S = 2, 16
    m, _ = build_model("uf_label_c32_s16", "f32")        # (module init draws from the CPU generator: build first)
    d = _diffusion(m)
    torch.manual_seed(int(v["native1000.rng_seed"]))
    x_T = torch.randn(B, 3, S, S)
    assert torch.equal(x_T, torch.from_numpy(v["native1000.x_T"]))

    def noise_fn(i):
        torch.zeros(2 * B).float().uniform_(0, 1)        # the UNet's mask draw precedes z in the stream
        return torch.randn(B, 3, S, S)

    samples, inter = d.p_sample_loop("native", (B, 3, S, S), _skw("native", 1000),
                                     denoise_sample_fn_kwargs=dict(cond=_cond(), layout=None, cond_scale=2.0),
                                     condition_kwargs={}, x_T=x_T, noise_fn=noise_fn)
    ref = torch.from_numpy(v["native1000.samples_u8"]).int()
    diff = (samples.cpu().int() - ref).abs()
    assert diff.max() <= 1
    assert (diff != 0).float().mean() < 1e-2
    assert tuple(inter["pred_x0"].shape) == (9, B, 3, S, S)
    assert rel_l2(inter["x_inter"].cpu(), v["native1000.x_inter"]) < 1e-3


def test_generic_denoiser_path_matches_fused_path():
    """a denoise_sample_fn that is NOT the drop-in's bound method takes the generic (guided eps, NCHW) path"""
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    cond = _cond()
    g = torch.Generator().manual_seed(3)
    x_T = torch.randn(2, 3, 16, 16, generator=g)
    z = torch.randn(10, 2, 3, 16, 16, generator=g)
    kw = dict(denoise_sample_fn_kwargs=dict(cond=cond, layout=None, cond_scale=2.0), condition_kwargs={},
              x_T=x_T, noise_fn=lambda i: z[i])
    a, _ = d.p_sample_loop("ddim", (2, 3, 16, 16), _skw("ddim", 10, 1.0), **kw)
    d.set_denoise_fn(m.forward, lambda x, t, **k: m.forward_with_cond_scale(x, t, **k))
    b, _ = d.p_sample_loop("ddim", (2, 3, 16, 16), _skw("ddim", 10, 1.0), **kw)
    assert (a.int() - b.int()).abs().max() <= 1


def test_plms10_trajectory_vs_reference():
    """sampling_method='plms' through p_sample_loop vs the reference's own PLMS trajectory (tests/golden/plms.npz)"""
    v = load_npz("plms.npz")
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    z = torch.from_numpy(v["plms10.z"])
    samples, inter = d.p_sample_loop("plms", (2, 3, 16, 16), _skw("plms", 10, 0.0),
                                     denoise_sample_fn_kwargs=dict(cond=_cond(), layout=None, cond_scale=2.0),
                                     condition_kwargs={}, x_T=torch.from_numpy(v["plms10.x_T"]),
                                     noise_fn=lambda j: z[j])
    assert samples.dtype == torch.uint8 and tuple(samples.shape) == (2, 3, 16, 16)
    assert tuple(inter["pred_x0"].shape) == (9, 2, 3, 16, 16)
    # free-running plumbing check: the Adams-Bashforth weights (up to 59/24) amplify the fp32
    # summation-order noise of each eps more than DDIM does, so the trajectory bound is looser than DDIM's 1e-3 (measured 1.8e-3); uint8 stays within 1 LSB
    r = rel_l2(inter["x_inter"].cpu(), v["plms10.x_inter"])
    d1 = (samples.cpu().int() - torch.from_numpy(v["plms10.samples_u8"]).int()).abs()
    d2 = (inter["pred_x0"].cpu().int() - torch.from_numpy(v["plms10.pred_x0_u8"]).int()).abs()
    f2 = float((d2 > 1).float().mean())
    print("plms free-running: rel_l2", r, "u8 max", int(d1.max()), int(d2.max()), "u8 >1 frac", float((d1 > 1).float().mean()), f2)
    assert r < 5e-3
    # final samples within 1 LSB; the nine intermediate pred_x0 images may hold an isolated 2-LSB pixel (a value that sits on a
    # rounding boundary of the uint8 conversion: with the head conv on its own fp32 kernel -- another summation order -- one of
    # 13,824 does, with it on the tiled kernel none; rel_l2 1.69e-3 vs 1.77e-3)
    assert d1.max() <= 1 and d2.max() <= 2 and f2 < 1e-3


def test_tensor_cond_scale_matches_per_sample_numbers():
    """forward_with_cond_scale with a [B,1,1,1] tensor of guidance weights == one evaluation per weight"""
    m, _ = build_model("uf_label_c32_s16", "f32")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 16, 16, generator=g).cuda()
    t = torch.tensor([700, 20]).cuda()
    w = torch.tensor([0.375, 2.25]).reshape(2, 1, 1, 1).cuda()
    with torch.no_grad():
        got = m.forward_with_cond_scale(x, t, cond_scale=w, cond=_cond(), layout=None)
        for i in range(2):
            ref = m.forward_with_cond_scale(x, t, cond_scale=float(w[i]), cond=_cond(), layout=None)
            assert max_rel(got[i].cpu(), ref[i].cpu()) < 2e-6


def test_ddim_chainvis_pairs_share_their_start_noise():
    """vis.chainvis (ddim_plms_sampler.py:157-175): (conditional, unconditional) chains from the same x_T"""
    class V:
        chainvis = True

        class chainvis_c:
            samples = 2
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    skw = dict(_skw("ddim", 4, 0.0), vis=V())
    torch.manual_seed(5)
    samples, inter = d.p_sample_loop("ddim", (2, 3, 16, 16), skw,
                                     denoise_sample_fn_kwargs=dict(cond=_cond(), layout=None, cond_scale=2.0),
                                     condition_kwargs={})
    assert tuple(samples.shape) == (4, 3, 16, 16)
    # rows 1 and 3 are unconditional (p0 = 0 keeps... the mask convention: p = 1 drops) -> differ from their partners
    assert (samples[0].int() - samples[1].int()).abs().max() > 0


# ------------------------------------------------------------------------------------------------------------------
# hipGraph-captured sampling step (sgdm_amd/diffusion.py: _GraphedStep): same kernels, same buffers' contents, same RNG
# draws in the same order -> the trajectory must be IDENTICAL to the eager launch sequence, bit for bit
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,method,steps", [("uf_label_c32_s16", "native", 40), ("uf_label_c32_s16", "ddim", 20),
                                               ("ca_stego_c32_s16", "native", 30)])
def test_graph_captured_step_equals_eager(name, method, steps):
    from test_hip_unet import INDEX
    from sgdm_amd.synth import synth_batch
    m, entry = build_model(name, "f16x3")
    d = _diffusion(m)
    kw = entry["ctor"]
    batch = synth_batch(kw["condition_method"], 2, 16, kw["cond_dim"], entry["layout_dim"], seed=23)
    cond = batch["cond"].cuda() if entry["kind"] == "unet_fast" else batch["cond"].float().cuda()
    dkw = dict(cond=cond, layout=batch["layout"].cuda() if "layout" in batch else None, cond_scale=2.0)
    if method == "native":
        skw, extra = _skw("native", 1000), dict(step_indices=list(range(steps - 1, -1, -1)))
        # snapshots at i in linspace(0, 1000, 10): only i = 0 falls into the visited range
    else:
        skw, extra = _skw("ddim", steps, 1.0), {}
    out = {}
    for graph in (False, True):
        torch.manual_seed(1234)                                   # z and the mask uniform come from the global generator
        samples, inter = d.p_sample_loop(method, (2, 3, 16, 16), dict(skw, hip_graph=graph),
                                         denoise_sample_fn_kwargs=dict(dkw), condition_kwargs={}, **extra)
        out[graph] = (samples.cpu(), inter["pred_x0"].cpu(), inter["x_inter"].cpu())
    assert torch.equal(out[False][0], out[True][0])
    assert torch.equal(out[False][1], out[True][1])
    assert torch.equal(out[False][2], out[True][2])
    assert out[True][1].shape[0] >= 1                             # at least one snapshot was taken from a replayed step


def test_graph_step_does_not_touch_the_callers_x_T():
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    x_T = torch.randn(2, 3, 16, 16, device="cuda")
    keep = x_T.clone()
    d.p_sample_loop("native", (2, 3, 16, 16), dict(_skw("native", 1000), hip_graph=True),
                    denoise_sample_fn_kwargs=dict(cond=_cond(), layout=None, cond_scale=2.0), condition_kwargs={},
                    x_T=x_T, step_indices=[999, 998, 997])
    assert torch.equal(x_T, keep)


def test_dynamic_thresholding_native_step_vs_oracle():
    """dtp < 1 (clip_x0_minus_one_to_one, diffusion_utils/util.py:70-79) through the sampler: per-sample quantile of |x0|
    by the radix-select kernel + the *_dyn step, against the oracle's torch.quantile restatement fed the same guided eps"""
    from oracle import diffusion_ref as D
    from sgdm_amd.synth import synth_batch
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    g = torch.Generator().manual_seed(77)
    B, i = 3, 700
    x = torch.randn(B, 3, 16, 16, generator=g) * 1.7
    z = torch.randn(B, 3, 16, 16, generator=g)
    dkw = dict(cond=synth_batch("label", B, 16, 10, seed=5)["cond"].cuda(), layout=None, cond_scale=2.0)
    t = torch.full((B,), i, dtype=torch.long)
    with torch.no_grad():
        eps = m.forward_with_cond_scale(x.cuda(), t.cuda(), **dkw).cpu()
    got, _ = d.sampler.sample((B, 3, 16, 16), sampling_kwargs=dict(_skw("native", 1000), dtp=0.9), denoise_sample_fn=d.denoise_sample_fn,
                              denoise_sample_fn_kwargs=dkw, x_T=x, noise_fn=lambda _i: z, step_indices=[i])
    sched = D.make_schedule()
    want, x0 = D.ddpm_step(sched, x, t, eps, z, True, 0.9, 1.0)
    assert max_rel(got.cpu(), want) < 2e-5
    unclipped = D.ddpm_step(sched, x, t, eps, z, False, 1.0, 1.0)[1]
    assert float(torch.quantile(unclipped.reshape(B, -1).abs(), 0.9, dim=-1).min()) > 1.0      # the threshold was active


def test_plms_with_dynamic_thresholding_vs_oracle():
    """sampling_method='plms' with dtp < 1 (p_sample_plms hands dtp to the same clip helper as ddim,
    ddim_plms_sampler.py:505-512): an analytic denoiser evaluated identically on both sides isolates the update"""
    from oracle import diffusion_ref as D
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)

    def eps_fn(x, t, **_):
        return torch.tanh(1.3 * x) * (0.5 + t.view(-1, 1, 1, 1).float() / 1000.0)

    d.set_denoise_fn(m.forward, eps_fn)
    g = torch.Generator().manual_seed(41)
    x_T = torch.randn(2, 3, 16, 16, generator=g) * 1.5
    z = torch.randn(12, 2, 3, 16, 16, generator=g)        # make_ddim_timesteps(6) has 7 entries -> 8 draws
    skw = dict(_skw("plms", 6, 0.0), dtp=0.9)
    samples, inter = d.p_sample_loop("plms", (2, 3, 16, 16), skw, denoise_sample_fn_kwargs={}, condition_kwargs={},
                                     x_T=x_T, noise_fn=lambda j: z[j])
    want, pred, xi, _ = D.plms_sample(D.make_schedule(), eps_fn, x_T, lambda j: z[j], 6, dtp=0.9)
    assert rel_l2(inter["x_inter"].cpu(), xi) < 2e-5
    assert (samples.cpu().int() - D.to_uint8(want).int()).abs().max() <= 1
    loose, _, _, _ = D.plms_sample(D.make_schedule(), eps_fn, x_T, lambda j: z[j], 6, dtp=1.0)
    assert (loose - want).abs().max() > 1e-3                      # the threshold changed the trajectory


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("dtp", [0.9, 0.995, 0.5])
def test_x0_quantile_kernel_matches_torch_quantile(kind, dtp):
    """sgd_x0_quantile (exact order statistics by radix select + torch's interpolation) == torch.quantile on the same x0"""
    import ctypes as C
    from sgdm_amd import _lib as L
    from sgdm_amd.diffusion import _quantile_rank
    lib = L.load()
    g = torch.Generator().manual_seed(int(dtp * 1000) + kind)
    B, Cc, hw = 5, 3, 64 * 64
    x = torch.randn(B, Cc, hw, generator=g) * 2
    eps_c, eps_u = torch.randn(B, hw, Cc, generator=g), torch.randn(B, hw, Cc, generator=g)
    eps = torch.cat([eps_c, eps_u]).cuda()                                  # UNet output at 2B, NHWC
    w = 2.0
    guided = ((1 - w) * eps_u + w * eps_c).permute(0, 2, 1)                  # imagen scale type
    coef = (C.c_float * 5)(1.3, 0.8, 0.0, 0.0, 0.0) if kind == 0 else (C.c_float * 5)(0.6, 0.7, 0.0, 0.0, 0.0)
    x0 = coef[0] * x - coef[1] * guided if kind == 0 else (x - coef[0] * guided) / torch.tensor(coef[1]).sqrt()
    want = torch.quantile(x0.reshape(B, -1).abs(), dtp, dim=-1).clamp(min=1.0)
    lo, hi, frac = _quantile_rank(dtp, Cc * hw)
    s = torch.empty(B, device="cuda")
    xd = x.cuda()
    L.check(lib.sgd_x0_quantile(kind, C.c_void_p(xd.data_ptr()), C.c_void_p(eps.data_ptr()), 1, w, coef, B, Cc, hw, lo, hi,
                                frac, C.c_void_p(s.data_ptr()), torch.cuda.current_stream().cuda_stream), "quantile")
    assert max_rel(s.cpu(), want) < 2e-6


def test_ddim_interp_variant_runs_and_batches_like_the_reference():
    """vis.interp (ddim_plms_sampler.py:142-155): samples * n interpolated (slerp) float cond rows, one shared start noise"""
    from sgdm_amd.synth import synth_batch

    class NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    cond = torch.rand(4, 10).cuda() + 0.1
    vis = NS(interp=True, interp_c=NS(n=3, samples=2))
    torch.manual_seed(5)
    samples, inter = d.p_sample_loop("ddim", (4, 3, 16, 16), dict(_skw("ddim", 10, 0.0), vis=vis),
                                     denoise_sample_fn_kwargs=dict(cond=cond, layout=None, cond_scale=2.0), condition_kwargs={})
    assert tuple(samples.shape) == (6, 3, 16, 16) and samples.dtype == torch.uint8
    from sgdm_amd.util import batch_to_conditioninterp
    rows = batch_to_conditioninterp(cond, 3, 2)
    assert tuple(rows.shape) == (6, 10)
    assert torch.allclose(rows[0], cond[0], atol=1e-5) and torch.allclose(rows[2], cond[1], atol=1e-5)   # slerp ends = the pair


def test_graph_step_with_cluster_ids_follows_weight_updates():
    """with class / cluster IDS as `cond`, the dropped (unconditional) CFG rows take w . null_cond_emb + b,
    a projection cached on the device.  It used to be refreshed inside the launch program, which a captured hipGraph does
    not re-run: after an optimizer step / EMA swap / load_state_dict a cached graph kept the OLD projection.  Graph
    sample -> change the weights -> graph sample must equal the eager path on the new weights, bit for bit."""
    from sgdm_amd.synth import weights_from_seed
    m, entry = build_model("uf_cluster5000_c32_s16", "f16x3")
    d = _diffusion(m)
    ids = torch.tensor([17, 4999]).cuda()
    dkw = dict(cond=ids, layout=None, cond_scale=2.0)
    skw = _skw("native", 1000)

    def run(graph):
        torch.manual_seed(99)
        return d.p_sample_loop("native", (2, 3, 16, 16), dict(skw, hip_graph=graph), denoise_sample_fn_kwargs=dict(dkw),
                               condition_kwargs={}, step_indices=[999, 998, 997, 0])[0].cpu()

    assert torch.equal(run(True), run(False))
    first = run(True)
    with torch.no_grad():                                         # what an optimizer step / LitEma.copy_to does: in place
        for name in ("mlp_cond.0.weight", "mlp_cond.0.bias", "null_cond_emb"):
            p = m.P(name)
            p.add_(torch.randn(p.shape, generator=torch.Generator().manual_seed(5)).cuda() * 0.05)
    after_graph, after_eager = run(True), run(False)
    assert not torch.equal(after_eager, first)
    assert torch.equal(after_graph, after_eager)
    m.load_state_dict(weights_from_seed(entry["manifest"], entry["seed"]))
    assert torch.equal(run(True), first)


# ------------------------------------------------------------------------------------------------------------------
# sampling variants against trajectories recorded from the reference WITH its noise (tests/golden/vis.npz,
# make_golden_vis.py): the `vis` branches of the DDIM sampler and dynamic thresholding (dtp < 1)
# ------------------------------------------------------------------------------------------------------------------
class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _vis_check(tag, samples, inter, v, tol=2e-3):
    r = rel_l2(inter["x_inter"].cpu(), v[tag + ".x_inter"])
    d = (samples.cpu().int() - torch.from_numpy(v[tag + ".samples_u8"]).int()).abs()
    assert tuple(samples.shape) == tuple(v[tag + ".samples_u8"].shape)
    assert r < tol, (tag, r)                       # free-running 7-step DDIM: plumbing tolerance
    assert d.max() <= 2 and (d > 1).float().mean() < 1e-3, (tag, int(d.max()))


def test_ddim_vis_interp_vs_reference():
    v = load_npz("vis.npz")
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    z = torch.from_numpy(v["interp.z"])
    vis = _NS(interp=True, interp_c=_NS(n=3, samples=2))
    samples, inter = d.p_sample_loop("ddim", (4, 3, 16, 16), dict(_skw("ddim", 6, 0.5), vis=vis),
                                     denoise_sample_fn_kwargs=dict(cond=torch.from_numpy(v["interp.cond"]).cuda(), layout=None,
                                                                   cond_scale=2.0),
                                     condition_kwargs={}, vis_noise=torch.from_numpy(v["interp.vis_noise"]),
                                     noise_fn=lambda i: z[i])
    _vis_check("interp", samples, inter, v)


def test_ddim_vis_chainvis_vs_reference():
    v = load_npz("vis.npz")
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    z = torch.from_numpy(v["chainvis.z"])
    vis = _NS(chainvis=True, chainvis_c=_NS(samples=2))
    samples, inter = d.p_sample_loop("ddim", (3, 3, 16, 16), dict(_skw("ddim", 6, 0.5), vis=vis),
                                     denoise_sample_fn_kwargs=dict(cond=torch.from_numpy(v["chainvis.cond"]).cuda(), layout=None,
                                                                   cond_scale=2.0),
                                     condition_kwargs={}, vis_noise=torch.from_numpy(v["chainvis.vis_noise"]),
                                     noise_fn=lambda i: z[i])
    _vis_check("chainvis", samples, inter, v)


def test_ddim_vis_condscale_vs_reference():
    from sgdm_amd.synth import synth_batch
    v = load_npz("vis.npz")
    m, _ = build_model("ca_stego_c32_s16", "f32")
    d = _diffusion(m)
    batch = synth_batch("stegoclusterlayout", 2, 16, 27, 27, seed=23 + 44)
    cond8 = batch["cond"][:1].float().repeat(8, 1)
    z = torch.from_numpy(v["condscale.z"])
    vis = _NS(condscale=True, condscale_c=_NS(samples=1))
    samples, inter = d.p_sample_loop("ddim", (2, 3, 16, 16), dict(_skw("ddim", 6, 0.5), vis=vis),
                                     denoise_sample_fn_kwargs=dict(cond=cond8.cuda(), layout=batch["layout"].cuda(), cond_scale=2.0),
                                     condition_kwargs={}, vis_noise=torch.from_numpy(v["condscale.vis_noise"]),
                                     noise_fn=lambda i: z[i])
    _vis_check("condscale", samples, inter, v)


def test_dynamic_thresholding_trajectories_vs_reference():
    """dtp = 0.9 (diffusion_utils/util.py:70-82): DDIM-10 (eta = 1) with the recorded noise, and the native sampler over all
    1000 steps with the reference's RNG stream replayed on the CPU (x_T, then uniform_(2B) + randn per step)"""
    v = load_npz("vis.npz")
    m, _ = build_model("uf_label_c32_s16", "f32")
    d = _diffusion(m)
    z = torch.from_numpy(v["dtp_ddim.z"])
    samples, inter = d.p_sample_loop("ddim", (2, 3, 16, 16), dict(_skw("ddim", 10, 1.0), dtp=0.9),
                                     denoise_sample_fn_kwargs=dict(cond=_cond(), layout=None, cond_scale=2.0), condition_kwargs={},
                                     x_T=torch.from_numpy(v["dtp_ddim.x_T"]), noise_fn=lambda i: z[i])
    _vis_check("dtp_ddim", samples, inter, v)
    B, S = 2, 16
    torch.manual_seed(int(v["dtp_native.rng_seed"]))
    x_T = torch.randn(B, 3, S, S)

    def noise_fn(i):
        torch.zeros(2 * B).float().uniform_(0, 1)
        return torch.randn(B, 3, S, S)
    samples, inter = d.p_sample_loop("native", (B, 3, S, S), dict(_skw("native", 1000), dtp=0.9),
                                     denoise_sample_fn_kwargs=dict(cond=_cond(), layout=None, cond_scale=2.0), condition_kwargs={},
                                     x_T=x_T, noise_fn=noise_fn)
    diff = (samples.cpu().int() - torch.from_numpy(v["dtp_native.samples_u8"]).int()).abs()
    assert diff.max() <= 1 and (diff != 0).float().mean() < 1e-2
    assert rel_l2(inter["x_inter"].cpu(), v["dtp_native.x_inter"]) < 1e-3
