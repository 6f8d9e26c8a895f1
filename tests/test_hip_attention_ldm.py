"""dynamic/attention_ldm.py CrossAttention / LinearCrossAttention on the HIP kernels against block vectors
recorded from the reference classes (tests/golden/attention_ldm.npz, make_golden_attention_ldm.py).  GPU only."""
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
    with pytest.raises(NotImplementedError):
        m(x, ctx, mask=mask)                                 # grad mode + trainable parameters: no silent detach


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
