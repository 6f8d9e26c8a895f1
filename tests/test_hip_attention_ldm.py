"""dynamic/attention_ldm.py CrossAttention / LinearCrossAttention on the HIP kernels against block vectors
recorded from the reference classes (tests/golden/attention_ldm.npz, make_golden_attention_ldm.py) and, in grad mode, against
the gradients the reference's autograd gives them (attention_ldm_train.npz, the same script with --train).  GPU only."""
import pytest
import torch

from conftest import load_npz, max_rel

pytestmark = pytest.mark.gpu

CASES = {   # name: (class, dim, context_dim, dim_head, heads, norm_context)  -- as in the generator
    "ca_d64": ("CrossAttention", 128, 48, 64, 4, False),
    "ca_d64_normctx_mask": ("CrossAttention", 128, 48, 64, 4, True),
    "ca_d24_mask": ("CrossAttention", 96, 32, 24, 4, True),
    "lin_d32": ("LinearCrossAttention", 64, 40, 32, 2, False),
    "lin_d24_h1_mask": ("LinearCrossAttention", 48, 16, 24, 1, True),
}


@pytest.mark.parametrize("prec,tol", [("f32", 5e-6), ("f16x3", 2e-5)])
@pytest.mark.parametrize("name", sorted(CASES))
def test_attention_ldm_blocks_vs_reference(name, prec, tol):
    from sgdm_amd import attention_ldm as A
    from sgdm_amd.synth import weights_from_seed
    v = load_npz("attention_ldm.npz")
    cls, dim, cdim, dh, heads, nc = CASES[name]
    m = getattr(A, cls)(dim, context_dim=cdim, dim_head=dh, heads=heads, norm_context=nc)
    # same state_dict names / shapes / order as the reference module
    manifest = [(k, tuple(t.shape)) for k, t in m.state_dict().items()]
    assert [f"{k}:{','.join(map(str, s))}" for k, s in manifest] == list(v[name + ".manifest"])
    sd = weights_from_seed(manifest, 23)
    for k in sd:
        if k.endswith(".beta"):
            sd[k] = torch.zeros_like(sd[k])
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.hip_precision = prec
    x, ctx = torch.from_numpy(v[name + ".x"]).cuda(), torch.from_numpy(v[name + ".context"]).cuda()
    mask = torch.from_numpy(v[name + ".mask"]).cuda() if name + ".mask" in v else None
    with torch.no_grad():
        y = m(x, ctx, mask=mask)
    err = max_rel(y.cpu(), v[name + ".y"])
    assert err < tol, err


def _module(name, v, prec):
    from sgdm_amd import attention_ldm as A
    from sgdm_amd.synth import weights_from_seed
    cls, dim, cdim, dh, heads, nc = CASES[name]
    m = getattr(A, cls)(dim, context_dim=cdim, dim_head=dh, heads=heads, norm_context=nc)
    manifest = [(k, tuple(t.shape)) for k, t in m.state_dict().items()]
    assert [f"{k}:{','.join(map(str, s))}" for k, s in manifest] == list(v[name + ".manifest"])
    sd = weights_from_seed(manifest, 23)
    for k in sd:
        if k.endswith(".beta"):
            sd[k] = torch.zeros_like(sd[k])
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.hip_precision = prec
    return m


@pytest.mark.parametrize("prec,tol", [("f32", 5e-6), ("f16x3", 5e-6)])
@pytest.mark.parametrize("name", sorted(CASES))
def test_attention_ldm_train_vs_reference_autograd(name, prec, tol):
    """grad-mode call + loss.backward() against the reference classes' autograd (attention_ldm_train.npz: loss = sum(y * gy);
    gradients of x, context and every trainable parameter).  north_star's 1e-4 is the contract; measured worst tensor 1.2e-6
    (profiles/r6_attention_ldm_train_errors.txt), asserted at 5e-6."""
    v = load_npz("attention_ldm_train.npz")
    m = _module(name, v, prec)
    x = torch.from_numpy(v[name + ".x"]).cuda().requires_grad_(True)
    ctx = torch.from_numpy(v[name + ".context"]).cuda().requires_grad_(True)
    mask = torch.from_numpy(v[name + ".mask"]).cuda() if name + ".mask" in v else None
    gy = torch.from_numpy(v[name + ".gy"]).cuda()
    y = m(x, ctx, mask=mask)
    assert y.requires_grad
    (y * gy).sum().backward()
    assert max_rel(y.detach().cpu(), v[name + ".y"]) < tol
    errs = {"x": max_rel(x.grad.cpu(), v[name + ".g.x"]), "context": max_rel(ctx.grad.cpu(), v[name + ".g.context"])}
    for k, prm in m.named_parameters():
        assert prm.grad is not None and prm.grad.shape == prm.shape, k
        errs[k] = max_rel(prm.grad.cpu(), v[name + ".g." + k])
    assert max(errs.values()) < tol, errs
    # buffers (the LayerNorm betas) are not parameters and receive nothing; a second backward accumulates like autograd's
    g1 = {k: prm.grad.clone() for k, prm in m.named_parameters()}
    (m(x, ctx, mask=mask) * gy).sum().backward()
    for k, prm in m.named_parameters():
        assert torch.allclose(prm.grad, 2 * g1[k], rtol=1e-5, atol=1e-6 * float(g1[k].abs().max())), k


def test_attention_ldm_frozen_parameters_and_inputs_only():
    """requires_grad flags are honoured: frozen parameters get no .grad, an input that needs none gets none, and under
    no_grad the module returns a tensor outside the graph"""
    v = load_npz("attention_ldm_train.npz")
    name = "ca_d64_normctx_mask"
    m = _module(name, v, "f32")
    for k, prm in m.named_parameters():
        prm.requires_grad_(k.startswith("to_q"))
    x = torch.from_numpy(v[name + ".x"]).cuda()
    ctx = torch.from_numpy(v[name + ".context"]).cuda().requires_grad_(True)
    mask = torch.from_numpy(v[name + ".mask"]).cuda()
    gy = torch.from_numpy(v[name + ".gy"]).cuda()
    (m(x, ctx, mask=mask) * gy).sum().backward()
    assert x.grad is None and max_rel(ctx.grad.cpu(), v[name + ".g.context"]) < 1e-5
    for k, prm in m.named_parameters():
        if k.startswith("to_q"):
            assert max_rel(prm.grad.cpu(), v[name + ".g." + k]) < 1e-5
        else:
            assert prm.grad is None, k
    with torch.no_grad():
        assert not m(x, ctx, mask=mask).requires_grad


def test_attention_ldm_wide_heads_train():
    """dim_head 96 (zero-padded to the 128-wide core) and 128 in grad mode: input and parameter gradients against fp32 torch
    autograd of the reference's forward restated on the module's own parameters (attention_ldm.py:220-254)"""
    from sgdm_amd import attention_ldm as A
    for dh in (96, 128):
        torch.manual_seed(dh)
        heads, dim, cdim, b, n, mm = 2, 64, 32, 2, 40, 6
        m = A.CrossAttention(dim, context_dim=cdim, dim_head=dh, heads=heads, norm_context=True).cuda()
        m.hip_precision = "f32"
        x = torch.randn(b, n, dim, device="cuda", requires_grad=True)
        ctx = torch.randn(b, mm, cdim, device="cuda", requires_grad=True)
        mask = torch.rand(b, mm, device="cuda") > 0.3
        mask[:, 0] = True
        gy = torch.randn(b, n, dim, device="cuda")
        (m(x, ctx, mask=mask) * gy).sum().backward()
        got = {"x": x.grad.cpu(), "context": ctx.grad.cpu(), **{k: p.grad.cpu() for k, p in m.named_parameters()}}
        # the reference's forward in plain torch on CPU copies
        P = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in m.named_parameters()}
        xc, cc = x.detach().cpu().requires_grad_(True), ctx.detach().cpu().requires_grad_(True)
        F = torch.nn.functional
        xn = F.layer_norm(xc, (dim,), P["norm.gamma"], torch.zeros(dim))
        cn = F.layer_norm(cc, (cdim,), P["norm_context.gamma"], torch.zeros(cdim))
        q = xn @ P["to_q.weight"].t()
        k, v = (cn @ P["to_kv.weight"].t()).chunk(2, dim=-1)
        sp = lambda t_: t_.reshape(b, -1, heads, dh).permute(0, 2, 1, 3)
        q, k, v = sp(q), sp(k), sp(v)
        nk, nv = P["null_kv"][0].expand(b, heads, 1, dh), P["null_kv"][1].expand(b, heads, 1, dh)
        k, v = torch.cat((nk, k), -2), torch.cat((nv, v), -2)
        sim = torch.einsum("bhid,bhjd->bhij", q * dh ** -0.5, k)
        mk = F.pad(mask.cpu(), (1, 0), value=True)[:, None, None, :]
        sim = sim.masked_fill(~mk, -torch.finfo(torch.float32).max)
        out = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).permute(0, 2, 1, 3).reshape(b, n, heads * dh)
        y = F.layer_norm(out @ P["to_out.0.weight"].t(), (dim,), P["to_out.1.gamma"], torch.zeros(dim))
        (y * gy.cpu()).sum().backward()
        want = {"x": xc.grad, "context": cc.grad, **{k_: p.grad for k_, p in P.items()}}
        errs = {k_: max_rel(got[k_], want[k_]) for k_ in want}
        assert max(errs.values()) < 5e-6, (dh, errs)


def _heads(t, heads, d):
    b, n, _ = t.shape
    return t.reshape(b, n, heads, d).permute(0, 2, 1, 3)


@pytest.mark.parametrize("d", [16, 32, 64, 128])
def test_masked_attention_backward_vs_autograd(d):
    """sgd_attention_masked_bwd against fp32 torch autograd of softmax(masked_fill(q k^T scale)) v (attention_ldm.py:239-254):
    masked keys get exactly zero gradient"""
    import ctypes as C
    from sgdm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(11 + d)
    b, heads, t, j = 2, 3, 150, 77
    q = torch.randn(b, t, heads * d, generator=g)
    kv = torch.randn(b, j, 2 * heads * d, generator=g)
    go = torch.randn(b, t, heads * d, generator=g)
    mask = torch.rand(b, j, generator=g) > 0.4
    mask[:, 0] = True
    scale = d ** -0.5
    qa, kva = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    sim = torch.einsum("bhid,bhjd->bhij", _heads(qa, heads, d) * scale, _heads(kva[..., :heads * d], heads, d))
    sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(torch.float32).max)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), _heads(kva[..., heads * d:], heads, d))
    out = out.permute(0, 2, 1, 3).reshape(b, t, heads * d)
    (out * go).sum().backward()
    p = lambda t_: C.c_void_p(t_.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    qd, kvd, god, mk = q.cuda(), kv.cuda(), go.cuda(), mask.to(torch.uint8).cuda()
    o = torch.empty(b, t, heads * d, device="cuda")
    lse = torch.empty(b, heads, t, device="cuda")
    kp, vp = p(kvd), C.c_void_p(kvd.data_ptr() + 4 * heads * d)
    L.check(lib.sgd_attention_masked(p(qd), heads * d, d, kp, vp, 2 * heads * d, d, p(mk), b, heads, t, j, d, scale, p(o),
                                     heads * d, p(lse), st), "fwd")
    assert max_rel(o.cpu(), out.detach()) < 2e-6
    dq = torch.full_like(qd, float("nan"))
    dkv = torch.full_like(kvd, float("nan"))
    dvec = torch.empty(b, heads, t, device="cuda")
    L.check(lib.sgd_attention_masked_bwd(p(qd), heads * d, d, kp, vp, 2 * heads * d, d, p(mk), p(o), heads * d, p(god),
                                         heads * d, p(lse), p(dvec), b, heads, t, j, d, scale, p(dq), p(dkv),
                                         C.c_void_p(dkv.data_ptr() + 4 * heads * d), st), "bwd")
    torch.cuda.synchronize()
    assert max_rel(dq.cpu(), qa.grad) < 5e-6
    assert max_rel(dkv.cpu(), kva.grad) < 5e-6
    assert float(dkv.cpu()[~mask].abs().max()) == 0.0


@pytest.mark.parametrize("d,masked", [(24, True), (32, False), (64, True), (128, True)])
def test_linear_attention_backward_vs_autograd(d, masked):
    """sgd_linear_attention_bwd against fp32 torch autograd of the reference's einsums (attention_ldm.py:283-296), head widths up
    to 128 (148 KB of LDS), more query rows than one tile, masked keys (k = -FLT_MAX, v = 0 in the forward: zero gradient)"""
    import ctypes as C
    from sgdm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5 + d)
    b, heads, t, j = 2, 3, 83, 9
    q = torch.randn(b, t, heads * d, generator=g)
    kv = torch.randn(b, j, 2 * heads * d, generator=g)
    go = torch.randn(b, t, heads * d, generator=g)
    mask = torch.rand(b, j, generator=g) > 0.3
    mask[:, 0] = True
    scale = d ** -0.5
    qa, kva = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    kh, vh = _heads(kva[..., :heads * d], heads, d), _heads(kva[..., heads * d:], heads, d)
    if masked:
        m4 = mask[:, None, :, None]
        kh = kh.masked_fill(~m4, -torch.finfo(torch.float32).max)
        vh = vh.masked_fill(~m4, 0.0)
    ref = torch.einsum("bhnd,bhde->bhne", _heads(qa, heads, d).softmax(-1) * scale,
                       torch.einsum("bhnd,bhne->bhde", kh.softmax(-2), vh))
    ref = ref.permute(0, 2, 1, 3).reshape(b, t, heads * d)
    (ref * go).sum().backward()
    p = lambda t_: C.c_void_p(t_.data_ptr())
    qd, kvd, god = q.cuda(), kv.cuda(), go.cuda()
    mk = mask.to(torch.uint8).cuda() if masked else None
    dq = torch.full_like(qd, float("nan"))
    dkv = torch.full_like(kvd, float("nan"))
    L.check(lib.sgd_linear_attention_bwd(p(qd), heads * d, d, p(kvd), C.c_void_p(kvd.data_ptr() + 4 * heads * d),
                                         2 * heads * d, d, p(mk) if masked else None, p(god), heads * d, b, heads, t, j, d,
                                         scale, p(dq), p(dkv), C.c_void_p(dkv.data_ptr() + 4 * heads * d),
                                         torch.cuda.current_stream().cuda_stream), "sgd_linear_attention_bwd")
    torch.cuda.synchronize()
    assert max_rel(dq.cpu(), qa.grad) < 5e-6
    assert max_rel(dkv.cpu(), kva.grad) < 5e-6
    if masked:
        assert float(dkv.cpu()[~mask].abs().max()) == 0.0


def test_masked_attention_core_equals_dropping_the_keys():
    """sgd_attention_masked == sgd_attention on the compacted key set (masking = weight exactly 0)"""
    import ctypes as C
    from sgdm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    b, heads, t, j, d = 1, 2, 70, 45, 32
    q = torch.randn(b, t, heads * d, generator=g).cuda()
    kv = torch.randn(b, j, 2 * heads * d, generator=g).cuda()
    mask = torch.rand(b, j, generator=g) > 0.4
    mask[:, 0] = True
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t_: C.c_void_p(t_.data_ptr())
    out = torch.empty(b, t, heads * d, device="cuda")
    mk = mask.to(torch.uint8).cuda()
    L.check(lib.sgd_attention_masked(p(q), heads * d, d, p(kv), C.c_void_p(kv.data_ptr() + 4 * heads * d), 2 * heads * d, d,
                                     p(mk), b, heads, t, j, d, d ** -0.5, p(out), heads * d, None, st), "masked")
    kv2 = kv[:, mask[0]].contiguous()
    ref = torch.empty_like(out)
    L.check(lib.sgd_attention(p(q), heads * d, d, p(kv2), C.c_void_p(kv2.data_ptr() + 4 * heads * d), 2 * heads * d, d,
                              b, heads, t, kv2.shape[1], d, d ** -0.5, p(ref), heads * d, None, st), "plain")
    assert max_rel(out.cpu(), ref.cpu()) < 2e-6


@pytest.mark.parametrize("d", [24, 64, 128])
def test_linear_attention_core_head_dims(d):
    """sgd_linear_attention (dynamic/attention_ldm.py:261-298: q softmax over d, k softmax over the keys, out = q~ (k~^T v))
    against plain fp32 torch, including d = 128 -- 66 KB of dynamic LDS, above the 64 KB a kernel gets without asking
    (the launch failed there)"""
    import ctypes as C
    from sgdm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    b, heads, t, j = 2, 3, 50, 9
    q = torch.randn(b, t, heads * d, generator=g)
    kv = torch.randn(b, j, 2 * heads * d, generator=g)
    mask = torch.rand(b, j, generator=g) > 0.3
    mask[:, 0] = True
    scale = d ** -0.5
    qh = q.reshape(b, t, heads, d).permute(0, 2, 1, 3)
    kh = kv[..., :heads * d].reshape(b, j, heads, d).permute(0, 2, 1, 3)
    vh = kv[..., heads * d:].reshape(b, j, heads, d).permute(0, 2, 1, 3)
    m4 = mask[:, None, :, None]
    kh = kh.masked_fill(~m4, -torch.finfo(torch.float32).max)
    vh = vh.masked_fill(~m4, 0.0)
    ref = torch.einsum("bhnd,bhde->bhne", qh.softmax(-1) * scale, torch.einsum("bhnd,bhne->bhde", kh.softmax(-2), vh))
    ref = ref.permute(0, 2, 1, 3).reshape(b, t, heads * d)
    qd, kvd, mk = q.cuda(), kv.cuda(), mask.to(torch.uint8).cuda()
    out = torch.full((b, t, heads * d), float("nan"), device="cuda")
    p = lambda t_: C.c_void_p(t_.data_ptr())
    L.check(lib.sgd_linear_attention(p(qd), heads * d, d, p(kvd), C.c_void_p(kvd.data_ptr() + 4 * heads * d), 2 * heads * d, d,
                                     p(mk), b, heads, t, j, d, scale, p(out), heads * d, torch.cuda.current_stream().cuda_stream),
            "sgd_linear_attention")
    torch.cuda.synchronize()
    assert max_rel(out.cpu(), ref) < 2e-6
