"""Backward kernels of the training step through the C-ABI vs torch-CPU autograd of the same op.  GPU only."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import max_rel
from test_hip_kernels import _lib, _nhwc, _p, _stream

pytestmark = pytest.mark.gpu


def _igemm_args(L, x, x1=None, conv=None, m=0, rows_per_n=0, pa=None, pb=None, silu=0, stride=1, resample=0):
    a = L.IgemmArgs()
    a.x0, a.c0 = x.data_ptr(), x.shape[-1]
    if x1 is not None:
        a.x1, a.c1 = x1.data_ptr(), x1.shape[-1]
    if conv is not None:
        n, hi, wi, ho, wo = conv
        a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride, a.resample = L.MODE_CONV3, n, hi, wi, ho, wo, stride, resample
    else:
        a.mode, a.m, a.rows_per_n, a.stride = L.MODE_FLAT, m, rows_per_n, 1
    if pa is not None:
        a.pro, a.pa, a.pb = L.PRO_AFFINE_NC, pa.data_ptr(), pb.data_ptr()
    a.pro_silu = silu
    return a


@pytest.mark.parametrize("prec,tol", [("f32", 2e-6), ("f16x3", 2e-5)])
@pytest.mark.parametrize("shape", [(2, 64, 16, 128), (3, 128, 8, 96), (2, 32, 4, 32)])
def test_conv_dgrad_is_forward_kernel_on_adjoint_weights(shape, prec, tol):
    L, lib = _lib()
    n, cin, h, cout = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, h, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    gy = torch.randn(n, cout, h, h, generator=g)
    F.conv2d(x, w, padding=1).backward(gy)
    p = L.PREC_BY_NAME[prec]
    wd = w.cuda()
    buf = torch.empty(lib.sgd_packed_weight_bytes(cin, cout, 3, p) // 4, device="cuda")
    cp, op = C.c_int32(), C.c_int32()
    L.check(lib.sgd_pack_weight_dgrad(_p(wd), _p(buf), cout, cin, 3, p, C.byref(cp), C.byref(op), _stream()), "packT")
    gyd = _nhwc(gy).cuda()
    out = torch.full((n, h, h, cin), float("nan"), device="cuda")
    a = _igemm_args(L, gyd, conv=(n, h, h, h, h))
    a.w, a.cin_p, a.cout_p, a.y, a.cout, a.y_ld, a.prec = buf.data_ptr(), cp.value, op.value, out.data_ptr(), cin, cin, p
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "dgrad")
    assert max_rel(out.cpu().permute(0, 3, 1, 2), x.grad) < tol


def _wgrad(L, lib, fwd, gy, cout, cin, taps, ksplit, scratch=False):
    slabs = torch.full((ksplit, taps, cout, cin), float("nan"), device="cuda")
    nbytes = int(lib.sgd_wgrad_scratch_bytes(C.byref(fwd), cout)) if scratch else 0
    # partial rows of the bias gradient the launch writes: ksplit, or the row chunks of the planes form's pre-pass
    brows = int(lib.sgd_wgrad_bias_rows(C.byref(fwd), cout, gy.shape[-1], ksplit, nbytes))
    assert brows >= 1
    bsl = torch.full((brows + 1, cout), float("nan"), device="cuda")        # (+ a guard row that must stay untouched)
    if scratch:          # operands pre-split into 16-bit planes by one element-wise pre-pass (sgd_wgrad_scratch)
        assert nbytes > 0
        buf = torch.full((nbytes // 4 + 4,), float("nan"), device="cuda")
        L.check(lib.sgd_wgrad_scratch(C.byref(fwd), _p(gy), gy.shape[-1], cout, _p(slabs), ksplit, _p(bsl), _p(buf), nbytes,
                                      _stream()), "wgrad_scratch")
    else:
        L.check(lib.sgd_wgrad(C.byref(fwd), _p(gy), gy.shape[-1], cout, _p(slabs), ksplit, _p(bsl), _stream()), "wgrad")
    assert torch.isnan(bsl[brows]).all() and torch.isfinite(bsl[:brows]).all()
    # the bias gradient rides along: partial column sums of gy, folded like sgd_colsum's second stage
    db = torch.full((cout,), float("nan"), device="cuda")
    L.check(lib.sgd_colsum_fold(_p(bsl), brows, cout, _p(db), 0, 1.0, _stream()), "fold")
    ref_db = gy.reshape(-1, gy.shape[-1])[:, :cout].double().sum(0).float()
    assert max_rel(db.cpu(), ref_db.cpu()) < 2e-6
    dw = torch.full((cout, cin, taps), float("nan"), device="cuda")
    L.check(lib.sgd_wgrad_reduce(_p(slabs), ksplit, taps, cout, cin, _p(dw), 0, 1.0, _stream()), "reduce")
    L.check(lib.sgd_wgrad_reduce(_p(slabs), ksplit, taps, cout, cin, _p(dw), 1, 1.0, _stream()), "reduce+")     # accumulate
    # ... and the one-launch form of both folds (what the training program uses)
    dw2, db2 = torch.full((cout, cin, taps), float("nan"), device="cuda"), torch.full((cout,), float("nan"), device="cuda")
    L.check(lib.sgd_wgrad_reduce_bias(_p(slabs), ksplit, taps, cout, cin, _p(dw2), 0, 1.0, _p(bsl), brows, _p(db2), _stream()),
            "reduce_bias")
    assert torch.equal(db2, db) and torch.equal(dw2 * 2, dw)
    return dw.cpu() / 2


@pytest.mark.parametrize("mode", ["plain", "fused_concat", "down", "up"])
def test_conv_wgrad(mode):
    L, lib = _lib()
    g = torch.Generator().manual_seed(12)
    n, h, cout = 3, 16, 160
    c0, c1 = (64, 32) if mode == "fused_concat" else (96, 0)
    cin = c0 + c1
    x0 = torch.randn(n, c0, h, h, generator=g)
    x1 = torch.randn(n, c1, h, h, generator=g) if c1 else None
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).requires_grad_(True)
    pa, pb = torch.randn(n, cin, generator=g), torch.randn(n, cin, generator=g)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    if mode == "plain":
        u, ho, rs, fused = xin, h, 0, False
    else:
        u = F.silu(xin * pa[:, :, None, None] + pb[:, :, None, None])
        fused = True
        if mode == "down":
            u, ho, rs = F.avg_pool2d(u, 2), h // 2, 1
        elif mode == "up":
            u, ho, rs = F.interpolate(u, scale_factor=2, mode="nearest"), h * 2, 2
        else:
            ho, rs = h, 0
    gy = torch.randn(n, cout, ho, ho, generator=g)
    F.conv2d(u, w, padding=1).backward(gy)
    x0d = _nhwc(x0).cuda()
    x1d = _nhwc(x1).cuda() if c1 else None
    pad, pbd = pa.cuda(), pb.cuda()
    fwd = _igemm_args(L, x0d, x1d, conv=(n, h, h, ho, ho), pa=pad if fused else None, pb=pbd if fused else None,
                      silu=1 if fused else 0, resample=rs)
    gyd = _nhwc(gy).cuda()
    dw = _wgrad(L, lib, fwd, gyd, cout, cin, 9, 3)
    assert max_rel(dw.reshape(cout, cin, 3, 3), w.grad) < 5e-6


def _train_ksplit(taps, cout, cin, rows):
    """the split the training program uses (sgdm_amd/train.py: wgrad_ksplit)"""
    from sgdm_amd.train import wgrad_ksplit
    return wgrad_ksplit(taps, cout, cin, rows)


@pytest.mark.parametrize("prec,tol", [("f32", 5e-6), ("f16x3", 3e-5), ("bf16x3", 3e-4)])
@pytest.mark.parametrize("shape", [(20, 128, 128, 64), (80, 1024, 512, 16), (40, 384, 128, 32)])
def test_conv_wgrad_split_precision_at_production_shapes(shape, prec, tol):
    """the weight-gradient kernel in the arithmetic mode and at the shapes / K splits the bs=80 train step runs it in
    (wgrad_conv_kernel<split, all taps>: 128->128 @64x64, the 1024->512 concat conv @16x16, a 384->128 concat @32x32),
    GroupNorm-affine + SiLU recomputed in its loader; reference = float64 autograd"""
    L, lib = _lib()
    n, cin, cout, h = shape
    g = torch.Generator().manual_seed(21)
    c0 = cin // 2 if cin > 128 else cin                       # the big ones are skip-concat convs (two sources)
    c1 = cin - c0
    x0 = torch.randn(n, c0, h, h, generator=g)
    x1 = torch.randn(n, c1, h, h, generator=g) if c1 else None
    pa, pb = 1 + 0.3 * torch.randn(n, cin, generator=g), 0.3 * torch.randn(n, cin, generator=g)
    gy = torch.randn(n, cout, h, h, generator=g) / (n * h * h) ** 0.5
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).double().requires_grad_(True)
    xin = (torch.cat([x0, x1], 1) if c1 else x0).double()
    u = F.silu(xin * pa.double()[:, :, None, None] + pb.double()[:, :, None, None])
    F.conv2d(u, w, padding=1).backward(gy.double())
    x0d, x1d = _nhwc(x0).cuda(), (_nhwc(x1).cuda() if c1 else None)
    pad, pbd = pa.cuda(), pb.cuda()
    fwd = _igemm_args(L, x0d, x1d, conv=(n, h, h, h, h), pa=pad, pb=pbd, silu=1)
    fwd.prec = L.PREC_BY_NAME[prec]
    gyd = _nhwc(gy).cuda()
    dw = _wgrad(L, lib, fwd, gyd, cout, cin, 9, _train_ksplit(9, cout, cin, n * h * h))
    err = max_rel(dw.reshape(cout, cin, 3, 3), w.grad.float())
    assert err < tol, err
    if prec != "f32":    # the planes forms (what the training program launches): the same arithmetic, bit for bit
        # default rule (csrc/backward.hip: wgrad_planes_ok): both operands pre-split into planes iff cin * cout > 100 * (cin + cout)
        # (or the fused-average-pool form); narrow layers transform in the loaders
        dw2 = _wgrad(L, lib, fwd, gyd, cout, cin, 9, _train_ksplit(9, cout, cin, n * h * h), scratch=True)
        assert torch.equal(dw, dw2)
        fwd.tune = L.TUNE_WGRAD_PLANES_ALWAYS
        dw3 = _wgrad(L, lib, fwd, gyd, cout, cin, 9, _train_ksplit(9, cout, cin, n * h * h), scratch=True)
        assert torch.equal(dw, dw3)


@pytest.mark.parametrize("prec,tol", [("f16x3", 3e-5), ("bf16x3", 3e-4)])
@pytest.mark.parametrize("shape,fused", [((80, 128, 128, 64), False), ((40, 256, 256, 32), False), ((6, 96, 160, 16), True)])
def test_strided_conv_wgrad_per_tap_on_the_split_kernel(shape, fused, prec, tol):
    """Downsample (conv 3x3 stride 2, openaimodel_ca.py:167-174) weight gradient: round 5 runs it as nine 1x1-style split-precision
    weight gradients (one tap per block, the tap's strided / shifted input pixels as rows) instead of the exact-f32 per-tap kernel
    (43-45 TF at C5's shapes).  Against float64 autograd at the training shapes, and against the kernel it replaces."""
    L, lib = _lib()
    n, cin, cout, h = shape
    g = torch.Generator().manual_seed(41)
    ho = h // 2
    x = torch.randn(n, cin, h, h, generator=g)
    pa, pb = 1 + 0.3 * torch.randn(n, cin, generator=g), 0.3 * torch.randn(n, cin, generator=g)
    gy = torch.randn(n, cout, ho, ho, generator=g) / (n * ho * ho) ** 0.5
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).double().requires_grad_(True)
    u = x.double()
    if fused:
        u = F.silu(u * pa.double()[:, :, None, None] + pb.double()[:, :, None, None])
    F.conv2d(u, w, stride=2, padding=1).backward(gy.double())
    xd, pad, pbd = _nhwc(x).cuda(), pa.cuda(), pb.cuda()
    fwd = _igemm_args(L, xd, None, conv=(n, h, h, ho, ho), pa=pad if fused else None, pb=pbd if fused else None,
                      silu=1 if fused else 0, stride=2)
    fwd.prec = L.PREC_BY_NAME[prec]
    gyd = _nhwc(gy).cuda()
    ks = _train_ksplit(9, cout, cin, n * ho * ho)
    dw = _wgrad(L, lib, fwd, gyd, cout, cin, 9, ks)
    err = max_rel(dw.reshape(cout, cin, 3, 3), w.grad.float())
    assert err < tol, err
    fwd.tune = L.TUNE_WGRAD_F32                                  # the exact-f32 per-tap kernel it replaces
    dw_old = _wgrad(L, lib, fwd, gyd, cout, cin, 9, ks)
    assert max_rel(dw_old.reshape(cout, cin, 3, 3), w.grad.float()) < 6e-6
    assert max_rel(dw, dw_old) < tol


@pytest.mark.parametrize("prec,tol", [("f16x3", 3e-5), ("bf16x3", 3e-4)])
@pytest.mark.parametrize("m,cin,cout,beta", [(80 * 256, 512, 512, True), (80 * 256, 512, 128, True), (1000, 96, 64, False)])
def test_linear_wgrad_with_layernorm_prologue_on_the_split_kernel(m, cin, cout, beta, prec, tol):
    """Attention_LR's to_q / to_kv (crossattetion_lr.py:81-88): y = LN(x) W^T.  Round 5: their weight gradients run on the
    split-precision 1x1 kernel (the LayerNorm row statistics travel through load_coef like the GroupNorm coefficients) instead
    of the exact-f32 per-tap kernel (56 TF at C5's shape).  Against float64 autograd, and against the kernel it replaces."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(57)
    x = torch.randn(m, cin, generator=g) * 1.5 + 0.3
    gamma = 1 + 0.2 * torch.randn(cin, generator=g)
    bt = 0.2 * torch.randn(cin, generator=g) if beta else None
    w = (torch.randn(cout, cin, generator=g) / math.sqrt(cin)).double().requires_grad_(True)
    gy = torch.randn(m, cout, generator=g) / m ** 0.5
    xn = F.layer_norm(x.double(), (cin,), gamma.double(), bt.double() if beta else None, 1e-5)
    F.linear(xn, w).backward(gy.double())
    xd, gd = x.cuda(), gamma.cuda()
    btd = bt.cuda() if beta else None
    st = torch.empty(m, 2, device="cuda")
    L.check(lib.sgd_ln_stats(_p(xd), m, cin, 1e-5, _p(st), _stream()), "ln_stats")
    fwd = _igemm_args(L, xd, m=m)
    fwd.pro, fwd.pa, fwd.pb, fwd.pc = L.PRO_LN_ROW, st.data_ptr(), gd.data_ptr(), (btd.data_ptr() if beta else 0)
    fwd.prec = L.PREC_BY_NAME[prec]
    gyd = gy.cuda()
    ks = _train_ksplit(1, cout, cin, m)
    dw = _wgrad(L, lib, fwd, gyd, cout, cin, 1, ks)
    err = max_rel(dw.reshape(cout, cin), w.grad.float())
    assert err < tol, err
    fwd.tune = L.TUNE_WGRAD_F32
    dw_old = _wgrad(L, lib, fwd, gyd, cout, cin, 1, ks)
    assert max_rel(dw_old.reshape(cout, cin), w.grad.float()) < 6e-6
    assert max_rel(dw, dw_old) < tol


@pytest.mark.parametrize("prec,tol", [("f32", 6e-6), ("f16x3", 4e-5)])
def test_wgrad_random_configurations(prec, tol):
    """40 seeded random weight-gradient launches (3x3: plain / fused prologue / concat / avg-pool / nearest-up / stride 2;
    1x1 rows) against float64 autograd, through whichever kernel sgd_wgrad picks for the shape -- wave-specialised with
    or without operand planes, pooled planes, the narrow stem / head kernels, the pipelined 1x1 kernel, the generic one --
    with random K splits and a gradient buffer wider than cout"""
    import random
    L, lib = _lib()
    p = L.PREC_BY_NAME[prec]
    rnd = random.Random(20260106)
    done = 0
    while done < 40:
        g = torch.Generator().manual_seed(rnd.randint(0, 1 << 30))
        kind = rnd.choice(["plain", "fused", "concat", "down", "up", "stride2", "flat", "flat"])
        n = rnd.randint(1, 6)
        h = rnd.choice([4, 8, 16, 32])
        c0 = rnd.choice([3, 4, 32, 64, 96, 128, 160, 256])
        cout = rnd.choice([3, 32, 64, 100, 128, 192, 256])
        c1 = rnd.choice([32, 64]) if (kind == "concat" and c0 % 32 == 0) else 0
        cin = c0 + c1
        if n * h * h * max(cin, cout) > 2_500_000:
            continue
        fused = kind in ("fused", "concat", "down", "up") or (kind == "flat" and rnd.random() < 0.5)
        x = torch.randn(n, cin, h, h, generator=g, dtype=torch.float64)
        pa = 1 + 0.3 * torch.randn(n, cin, generator=g, dtype=torch.float64)
        pb = 0.3 * torch.randn(n, cin, generator=g, dtype=torch.float64)
        u = F.silu(x * pa[:, :, None, None] + pb[:, :, None, None]) if fused else x
        taps, stride, rs, ho = 9, 1, 0, h
        if kind == "down":
            u, ho, rs = F.avg_pool2d(u, 2), h // 2, 1
        elif kind == "up":
            u, ho, rs = F.interpolate(u, scale_factor=2, mode="nearest"), h * 2, 2
        elif kind == "stride2":
            stride, ho = 2, h // 2
        if ho < 2:
            continue
        if kind == "flat":
            taps = 1
            w = (torch.randn(cout, cin, generator=g, dtype=torch.float64) / math.sqrt(cin)).requires_grad_(True)
            ur = u.permute(0, 2, 3, 1).reshape(n * h * h, cin)
            gy = torch.randn(n * h * h, cout, generator=g, dtype=torch.float64)
            F.linear(ur, w).backward(gy)
            gy_rows = gy
        else:
            w = (torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / math.sqrt(cin * 9)).requires_grad_(True)
            gy = torch.randn(n, cout, ho, ho, generator=g, dtype=torch.float64)
            F.conv2d(u, w, stride=stride, padding=1).backward(gy)
            gy_rows = gy.permute(0, 2, 3, 1).reshape(n * ho * ho, cout)
        # ---- the launch
        xr = x.float()
        x0d = _nhwc(xr[:, :c0]).cuda()
        x1d = _nhwc(xr[:, c0:]).cuda() if c1 else None
        pad, pbd = pa.float().cuda(), pb.float().cuda()
        if kind == "flat":
            fwd = _igemm_args(L, x0d.reshape(n * h * h, c0), None, m=n * h * h, rows_per_n=h * h,
                              pa=pad if fused else None, pb=pbd if fused else None, silu=1 if fused else 0)
        else:
            fwd = _igemm_args(L, x0d, x1d, conv=(n, h, h, ho, ho), pa=pad if fused else None, pb=pbd if fused else None,
                              silu=1 if fused else 0, stride=stride, resample=rs)
        fwd.prec = p
        gy_ld = cout + rnd.choice([0, 0, 4, 32]) if cout % 4 == 0 else cout
        gyd = torch.full((gy_rows.shape[0], gy_ld), float("nan"), device="cuda")
        gyd[:, :cout] = gy_rows.float().cuda()
        if gy_ld > cout:
            gyd[:, cout:] = 7.0                                  # finite junk next to the gradient: must not be read as data
        rows = gy_rows.shape[0]
        ks = min(rnd.choice([1, 2, 3, _train_ksplit(taps, cout, cin, rows)]), (rows + 63) // 64)     # (<= K tiles of 64 rows)
        scratch = int(lib.sgd_wgrad_scratch_bytes(C.byref(fwd), cout)) > 0 and rnd.random() < 0.6
        desc = f"case {done}: {kind} n={n} h={h} c0={c0} c1={c1} cout={cout} fused={fused} gy_ld={gy_ld} ksplit={ks} scratch={scratch}"
        dw = _wgrad(L, lib, fwd, gyd, cout, cin, taps, ks, scratch=scratch)
        ref = w.grad.float().reshape(cout, cin, taps)
        err = max_rel(dw, ref)
        assert err < tol, (desc, err)
        done += 1


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("kind,n,h,ch", [("stem3", 5, 64, 128), ("stem4", 3, 32, 64), ("head", 5, 64, 128), ("head", 2, 16, 64),
                                         ("stem3", 2, 16, 256)])
def test_conv_wgrad_stem_and_head(kind, n, h, ch, prec, monkeypatch):
    """round 4: the stem (3 / 4 input channels) and the output head (3 output channels, GroupNorm + SiLU prologue) have
    weight-gradient kernels of their own (HBM-bound row walks in fp32 FMA, any arithmetic mode) instead of the generic
    per-tap MFMA kernel: both against float64 autograd, bias gradient included (checked inside _wgrad)"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(44)
    if kind.startswith("stem"):
        cin, cout, fused = int(kind[-1]), ch, False
    else:
        cin, cout, fused = ch, 3, True
    x = torch.randn(n, cin, h, h, generator=g)
    pa, pb = 1 + 0.3 * torch.randn(n, cin, generator=g), 0.3 * torch.randn(n, cin, generator=g)
    gy = torch.randn(n, cout, h, h, generator=g)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).double().requires_grad_(True)
    u = x.double()
    if fused:
        u = F.silu(u * pa.double()[:, :, None, None] + pb.double()[:, :, None, None])
    F.conv2d(u, w, padding=1).backward(gy.double())
    xd, pad, pbd = _nhwc(x).cuda(), pa.cuda(), pb.cuda()          # (kept alive: the descriptor holds raw pointers)
    fwd = _igemm_args(L, xd, None, conv=(n, h, h, h, h), pa=pad if fused else None, pb=pbd if fused else None,
                      silu=1 if fused else 0)
    fwd.prec = L.PREC_BY_NAME[prec]
    gyd = _nhwc(gy).cuda()
    ks = _train_ksplit(9, cout, cin, n * h * h)
    dw = _wgrad(L, lib, fwd, gyd, cout, cin, 9, ks)
    assert max_rel(dw.reshape(cout, cin, 3, 3), w.grad.float()) < 5e-6
    fwd.tune = L.TUNE_WGRAD_GENERIC_NARROW                    # the generic kernel they replace, same slabs
    dw_old = _wgrad(L, lib, fwd, gyd, cout, cin, 9, ks)
    assert max_rel(dw_old.reshape(cout, cin, 3, 3), w.grad.float()) < (5e-6 if prec == "f32" else 3e-5)


@pytest.mark.parametrize("prec,tol", [("f16x3", 3e-5), ("bf16x3", 3e-4)])
@pytest.mark.parametrize("shape", [(20, 128, 128, 64), (40, 256, 256, 32)])
def test_conv_wgrad_fused_avgpool_through_pooled_planes(shape, prec, tol, monkeypatch):
    """ResBlock(down) first conv (openaimodel.py:301-306): the conv reads avg_pool2d(SiLU(GN(x))).  Round 4: its weight
    gradient runs on the wave-specialised kernel -- act_split_kernel writes the operand planes at the POOLED resolution
    -- instead of the generic per-tap kernel; both must agree with float64 autograd, and with each other to rounding"""
    L, lib = _lib()
    n, cin, cout, h = shape
    g = torch.Generator().manual_seed(33)
    x = torch.randn(n, cin, h, h, generator=g)
    pa, pb = 1 + 0.3 * torch.randn(n, cin, generator=g), 0.3 * torch.randn(n, cin, generator=g)
    ho = h // 2
    gy = torch.randn(n, cout, ho, ho, generator=g) / (n * ho * ho) ** 0.5
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).double().requires_grad_(True)
    u = F.avg_pool2d(F.silu(x.double() * pa.double()[:, :, None, None] + pb.double()[:, :, None, None]), 2)
    F.conv2d(u, w, padding=1).backward(gy.double())
    xd, pad, pbd = _nhwc(x).cuda(), pa.cuda(), pb.cuda()
    fwd = _igemm_args(L, xd, None, conv=(n, h, h, ho, ho), pa=pad, pb=pbd, silu=1, resample=1)
    fwd.prec = L.PREC_BY_NAME[prec]
    gyd = _nhwc(gy).cuda()
    ks = _train_ksplit(9, cout, cin, n * ho * ho)
    dw = _wgrad(L, lib, fwd, gyd, cout, cin, 9, ks, scratch=True)            # pooled planes + wave-specialised kernel
    err = max_rel(dw.reshape(cout, cin, 3, 3), w.grad.float())
    assert err < tol, err
    fwd.tune = L.TUNE_WGRAD_NO_POOLED_PLANES
    dw_old = _wgrad(L, lib, fwd, gyd, cout, cin, 9, ks, scratch=True)        # the generic kernel it replaces
    assert max_rel(dw_old.reshape(cout, cin, 3, 3), w.grad.float()) < tol
    assert max_rel(dw, dw_old) < tol


@pytest.mark.parametrize("prec,tol", [("f16x3", 3e-5), ("bf16x3", 3e-4)])
@pytest.mark.parametrize("m,k,nout", [(80 * 256, 512, 1536), (13824, 768, 80)])
def test_linear_wgrad_split_precision_at_production_shapes(m, k, nout, prec, tol):
    """1x1 / linear weight gradient in split precision: attention qkv (512 -> 1536 over 80*256 rows), and the input gradient of
    all ResBlocks' emb_layers seen as a weight gradient (train.py, _film: "rows" = the 13,824 FiLM outputs of C2, "gy" = the
    FiLM gradient transposed, 80 columns = the batch)"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(22)
    x = torch.randn(m, k, generator=g)
    w = (torch.randn(nout, k, generator=g) / math.sqrt(k)).double().requires_grad_(True)
    gy = torch.randn(m, nout, generator=g) / m ** 0.5
    F.linear(x.double(), w).backward(gy.double())
    fwd = _igemm_args(L, x.cuda(), m=m)
    fwd.prec = L.PREC_BY_NAME[prec]
    dw = _wgrad(L, lib, fwd, gy.cuda(), nout, k, 1, _train_ksplit(1, nout, k, m))
    err = max_rel(dw.reshape(nout, k), w.grad.float())
    assert err < tol, err


def test_linear_wgrad_and_bias():
    L, lib = _lib()
    g = torch.Generator().manual_seed(13)
    m, k, nout = 333, 200, 72
    x = torch.randn(m, k, generator=g)
    w = (torch.randn(nout, k, generator=g) / math.sqrt(k)).requires_grad_(True)
    b = torch.zeros(nout, requires_grad=True)
    gy = torch.randn(m, nout, generator=g)
    F.linear(F.silu(x), w, b).backward(gy)
    xd, gyd = x.cuda(), gy.cuda()
    fwd = _igemm_args(L, xd, m=m, silu=1)
    dw = _wgrad(L, lib, fwd, gyd, nout, k, 1, 3)
    assert max_rel(dw.reshape(nout, k), w.grad) < 5e-6
    db = torch.zeros(nout, device="cuda")
    work = torch.empty(16, nout, device="cuda")
    L.check(lib.sgd_colsum(_p(gyd), m, nout, nout, _p(db), 0, 1.0, _p(work), 16, _stream()), "colsum")
    assert max_rel(db.cpu(), b.grad) < 2e-6


@pytest.mark.parametrize("gu_mode", [0, 1, 2])
@pytest.mark.parametrize("silu,film", [(1, True), (0, False)])
def test_groupnorm_silu_backward(gu_mode, silu, film):
    """x -> GN(+FiLM) [-> SiLU] [-> avgpool / nearest-up] -> gu ; plus an identity-skip gradient through the
    same resample (ResBlock up/down, openaimodel.py:301-306)"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(14)
    n, c, h = 3, 96, 8
    x = (torch.randn(n, c, h, h, generator=g) * 2 + 0.3).requires_grad_(True)
    gamma = torch.randn(c, generator=g).requires_grad_(True)
    beta = torch.randn(c, generator=g).requires_grad_(True)
    fl = torch.randn(n, 2 * c, generator=g).requires_grad_(True)
    y = F.group_norm(x, 32, gamma, beta, 1e-5)
    if film:
        y = y * (1 + fl[:, :c, None, None]) + fl[:, c:, None, None]
    u = F.silu(y) if silu else y
    rs = {0: lambda t: t, 1: lambda t: F.avg_pool2d(t, 2), 2: lambda t: F.interpolate(t, scale_factor=2, mode="nearest")}[gu_mode]
    u2, skip = rs(u), rs(x)
    gu, gs = torch.randn(u2.shape, generator=g), torch.randn(u2.shape, generator=g)
    (u2 * gu + skip * gs).sum().backward()
    # ---- HIP: forward stats/coefs, then the three backward kernels
    xd = _nhwc(x.detach()).cuda()
    hw = h * h
    sums = torch.empty(n, c, 2, device="cuda")
    L.check(lib.sgd_chan_stats(_p(xd), n, hw, c, _p(sums), c, 0, _stream()), "stats")
    gd, bd, fd = gamma.detach().cuda(), beta.detach().cuda(), fl.detach().cuda()
    a, b = torch.empty(n, c, device="cuda"), torch.empty(n, c, device="cuda")
    L.check(lib.sgd_gn_coef(_p(sums), _p(gd), _p(bd), _p(fd) if film else None, 2 * c, n, c, 32, hw, 1e-5, _p(a),
                            _p(b), _stream()), "coef")
    gud, gsd = _nhwc(gu).cuda(), _nhwc(gs).cuda()
    S = torch.empty(n, c, 2, device="cuda")
    L.check(lib.sgd_gn_bwd_reduce(_p(xd), n, h, h, c, c, 0, _p(a), _p(b), silu, _p(gud), c, gu_mode, 0.0, 0, _p(S), _stream()),
            "reduce")
    A, B, Cc = (torch.empty(n, c, device="cuda") for _ in range(3))
    dg, db = torch.empty(n, c, device="cuda"), torch.empty(n, c, device="cuda")
    dfilm = torch.zeros(n, 2 * c, device="cuda")
    L.check(lib.sgd_gn_bwd_coef(_p(S), 1, _p(sums), _p(gd), _p(bd), _p(fd) if film else None, 2 * c, n, c, 32, hw, 1e-5,
                                _p(A), _p(B), _p(Cc), _p(dg), _p(db), _p(dfilm) if film else None, _stream()), "bcoef")
    dx = torch.full((n, h, h, c), float("nan"), device="cuda")
    L.check(lib.sgd_gn_bwd_apply(_p(xd), n, h, h, c, c, 0, _p(a), _p(b), silu, _p(gud), c, gu_mode, 0.0, 0, _p(A), _p(B),
                                 _p(Cc), _p(gsd), c, gu_mode, _p(dx), c, 0, 0, _stream()), "apply")
    assert max_rel(dx.cpu().permute(0, 3, 1, 2), x.grad) < 2e-5
    assert max_rel(dg.cpu().sum(0), gamma.grad) < 2e-5
    assert max_rel(db.cpu().sum(0), beta.grad) < 2e-5
    if film:
        assert max_rel(dfilm.cpu(), fl.grad) < 2e-5


@pytest.mark.parametrize("n,c,film", [(80, 128, True), (80, 1024, True), (3, 96, False), (160, 384, False), (256, 512, True),
                                      (7, 32, True), (160, 1024, False)])
def test_groupnorm_backward_coefficients_and_column_sums_in_one_launch(n, c, film):
    """sgd_gn_bwd_coef_fold (round 5) against sgd_gn_bwd_coef + sgd_colsum_pair on the same statistics: A, B, C, dfilm and
    the scaled dgamma / dbeta -- every bit (the fold keeps the two-stage column sum's order of additions), plain and
    accumulating"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(n * 1000 + c)
    hw, groups = 4096, 32
    S = torch.randn(n, c, 2, generator=g).cuda()
    x1 = torch.randn(n, c, generator=g) * 40
    sums = torch.stack([x1, x1 * x1 / hw + hw * (0.5 + torch.rand(n, c, generator=g))], -1).cuda()     # var > 0
    gamma, beta = torch.randn(c, generator=g).cuda(), torch.randn(c, generator=g).cuda()
    fl = torch.randn(n, 2 * c, generator=g).cuda() if film else None
    scale = 1.0 / 4096.0
    A, B, Cc, dg, db = (torch.empty(n, c, device="cuda") for _ in range(5))
    dfilm = torch.zeros(n, 2 * c, device="cuda")
    L.check(lib.sgd_gn_bwd_coef(_p(S), 1, _p(sums), _p(gamma), _p(beta), _p(fl), 2 * c, n, c, groups, hw, 1e-5, _p(A), _p(B), _p(Cc),
                                _p(dg), _p(db), _p(dfilm) if film else None, _stream()), "coef")
    base = torch.randn(2, c, generator=g).cuda()
    for acc in (0, 1):
        o1, o2 = base[0].clone(), base[1].clone()
        L.check(lib.sgd_colsum_pair(_p(dg), _p(db), n, c, c, _p(o1), _p(o2), acc, scale, _stream()), "pair")
        A2, B2, C2 = (torch.full((n, c), float("nan"), device="cuda") for _ in range(3))
        dfilm2 = torch.zeros(n, 2 * c, device="cuda")
        p1, p2 = base[0].clone(), base[1].clone()
        L.check(lib.sgd_gn_bwd_coef_fold(_p(S), 1, _p(sums), _p(gamma), _p(beta), _p(fl), 2 * c, n, c, groups, hw, 1e-5, _p(A2),
                                         _p(B2), _p(C2), _p(dfilm2) if film else None, _p(p1), _p(p2), acc, scale, _stream()),
                "fold")
        torch.cuda.synchronize()
        assert torch.equal(A2, A) and torch.equal(B2, B) and torch.equal(C2, Cc) and torch.equal(dfilm2, dfilm)
        assert torch.equal(p1, o1) and torch.equal(p2, o2)
    # what does not fit the fold's LDS tables is refused, not truncated
    assert lib.sgd_gn_bwd_coef_fold(_p(S), 1, _p(sums), _p(gamma), _p(beta), None, 0, 257, c, groups, hw, 1e-5, _p(A), _p(B), _p(Cc),
                                    None, _p(o1), _p(o2), 0, scale, _stream()) == 1


# n, source channel counts, h, silu, film, dropout, accumulate
GN_ROWS_CASES = [(5, (128,), 64, 1, True, 0.0, 0),        # ResBlock GroupNorm at 64x64: 16 chunks of 4 windows
                 (3, (256, 128), 32, 1, False, 0.0, 1),   # decoder in_layers.0 over a concat: shared partial table, accumulate
                 (4, (512,), 32, 0, False, 0.0, 0),       # no SiLU: two channel sets per lane
                 (2, (1024,), 32, 1, True, 0.1, 0),       # four sets per lane, train-time dropout mask on the gradient
                 (7, (64,), 32, 1, False, 0.25, 0),       # 16 quads per row: four rows per wave instruction
                 (3, (256, 32), 32, 1, False, 0.0, 0)]    # sources whose own chunk counts differ (16 and 4)


@pytest.mark.parametrize("case", GN_ROWS_CASES, ids=["-".join(map(str, c)) for c in GN_ROWS_CASES])
def test_groupnorm_backward_row_stream_passes_equal_the_round1_passes(case):
    """sgd_gn_bwd_reduce_rows (round 6: contiguous 8 KiB pieces per wave, per-chunk partial sums folded by the coefficient launch)
    against sgd_gn_bwd_reduce on the same tensors: the folded statistics to the rounding of a different summation order, the
    coefficient launches on the chunked table against its host-side fold, and the whole chain (reduce_rows -> coef -> apply)
    against float64 autograd through GroupNorm(+FiLM)(+SiLU)(+dropout mask)"""
    n, cs, h, silu, film, drop, acc = case
    L, lib = _lib()
    ct, hw = sum(cs), h * h
    g = torch.Generator().manual_seed(17 + ct + h)
    xs = [(torch.randn(n, h, h, c, generator=g) * 2 + 0.3).cuda() for c in cs]
    gamma, beta = torch.randn(ct, generator=g).cuda(), torch.randn(ct, generator=g).cuda()
    fl = torch.randn(n, 2 * ct, generator=g).cuda() if film else None
    gu = torch.randn(n, h, h, ct, generator=g).cuda()
    gres = torch.randn(n, h, h, ct, generator=g).cuda()
    seed = 12345
    sums = torch.zeros(n, ct, 2, device="cuda")
    off = 0
    for x, c in zip(xs, cs):
        L.check(lib.sgd_chan_stats(_p(x), n, hw, c, _p(sums), ct, off, _stream()), "stats")
        off += c
    a, b = torch.empty(n, ct, device="cuda"), torch.empty(n, ct, device="cuda")
    L.check(lib.sgd_gn_coef(_p(sums), _p(gamma), _p(beta), _p(fl) if film else None, 2 * ct, n, ct, 32, hw, 1e-5, _p(a), _p(b),
                            _stream()), "coef")
    ks = [int(lib.sgd_gn_bwd_rows_chunks(n, h, h, c)) for c in cs]
    chunks = min(ks)
    assert chunks > 0 and all(k % chunks == 0 for k in ks), ks
    S0 = torch.full((n, ct, 2), float("nan"), device="cuda")
    P = torch.full((n, chunks, ct, 2), float("nan"), device="cuda")
    off = 0
    for x, c in zip(xs, cs):
        L.check(lib.sgd_gn_bwd_reduce(_p(x), n, h, h, c, ct, off, _p(a), _p(b), silu, _p(gu), ct, 0, drop, seed, _p(S0), _stream()), "r0")
        L.check(lib.sgd_gn_bwd_reduce_rows(_p(x), n, h, h, c, ct, off, _p(a), _p(b), silu, _p(gu), ct, drop, seed, chunks, _p(P),
                                           _stream()), "r1")
        off += c
    torch.cuda.synchronize()
    S1 = P.double().sum(1).float().double()                     # (the launches round the folded sums to float once)
    assert torch.isfinite(P).all() and max_rel(S1, S0.double()) < 2e-6
    # coefficients from the chunked table (folded inside the launch) == coefficients from its host-side fold
    outs = []
    for Sx, sch in ((P, chunks), (S1.float().contiguous(), 1)):
        A, B, Cc, dg, db = (torch.full((n, ct), float("nan"), device="cuda") for _ in range(5))
        dfilm = torch.zeros(n, 2 * ct, device="cuda")
        L.check(lib.sgd_gn_bwd_coef(_p(Sx), sch, _p(sums), _p(gamma), _p(beta), _p(fl) if film else None, 2 * ct, n, ct, 32, hw, 1e-5,
                                    _p(A), _p(B), _p(Cc), _p(dg), _p(db), _p(dfilm) if film else None, _stream()), "bcoef")
        outs.append((A, B, Cc, dg, db, dfilm))
    for u, v in zip(outs[0], outs[1]):
        assert max_rel(u, v) < 1e-6
    A, B, Cc, dg, db, dfilm = outs[0]
    if n <= 256 and 16 * n * (ct // 32) + 32 * n + 4 + 2048 <= 64 * 1024:
        A2, B2, C2 = (torch.full((n, ct), float("nan"), device="cuda") for _ in range(3))
        df2, o1, o2 = torch.zeros(n, 2 * ct, device="cuda"), torch.zeros(ct, device="cuda"), torch.zeros(ct, device="cuda")
        L.check(lib.sgd_gn_bwd_coef_fold(_p(P), chunks, _p(sums), _p(gamma), _p(beta), _p(fl) if film else None, 2 * ct, n, ct, 32, hw,
                                         1e-5, _p(A2), _p(B2), _p(C2), _p(df2) if film else None, _p(o1), _p(o2), 0, 1.0, _stream()), "fold")
        torch.cuda.synchronize()
        assert torch.equal(A2, A) and torch.equal(B2, B) and torch.equal(C2, Cc) and torch.equal(df2, dfilm)
    off, dxs = 0, []
    for x, c in zip(xs, cs):
        base = torch.randn(n, h, h, c, generator=g).cuda()
        d0 = base.clone()
        L.check(lib.sgd_gn_bwd_apply(_p(x), n, h, h, c, ct, off, _p(a), _p(b), silu, _p(gu), ct, 0, drop, seed, _p(A), _p(B), _p(Cc),
                                     _p(gres), ct, 0, _p(d0), c, 0, acc, _stream()), "a0")
        torch.cuda.synchronize()
        dxs.append(d0 - base if acc else d0)
        off += c
    # float64 autograd of the whole chain (the dropout mask is the kernels' own: element (row, channel) of the concatenated tensor)
    xcat = torch.cat([x.double() for x in xs], -1).permute(0, 3, 1, 2).cpu().requires_grad_(True)
    gam, bet = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    y = F.group_norm(xcat, 32, gam, bet, 1e-5)
    if film:
        f64 = fl.double().cpu()
        y = y * (1 + f64[:, :ct, None, None]) + f64[:, ct:, None, None]
    u = F.silu(y) if silu else y
    guc = gu.double().cpu().permute(0, 3, 1, 2)
    if drop > 0:
        import numpy as np
        from test_hip_train import _host_keep_mask
        keep = torch.from_numpy(_host_keep_mask(seed, n * hw, ct, drop).astype(np.float64)).reshape(n, h, h, ct).permute(0, 3, 1, 2)
        guc = guc * keep / (1.0 - drop)
    (u * guc + xcat * gres.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    got = torch.cat(dxs, -1).cpu().permute(0, 3, 1, 2)
    assert max_rel(got, xcat.grad) < 2e-5
    assert max_rel(dg.cpu().sum(0), gam.grad) < 2e-5 and max_rel(db.cpu().sum(0), bet.grad) < 2e-5


def test_groupnorm_backward_row_stream_shapes():
    """which shapes the row-stream reduce serves (no launch): >= 32 x 32 pixels, powers of two in 16 .. 1024 channels, whole 32 KiB
    windows"""
    L, lib = _lib()
    f = lib.sgd_gn_bwd_rows_chunks
    assert f(80, 64, 64, 128) == 16 and f(80, 32, 32, 256) == 16 and f(80, 32, 32, 32) == 4 and f(80, 32, 32, 16) == 2
    assert f(80, 16, 16, 512) == 0 and f(1, 8, 8, 128) == 0
    assert f(80, 64, 64, 96) == 0 and f(80, 64, 64, 224) == 0 and f(80, 64, 64, 2048) == 0
    assert lib.sgd_gn_bwd_reduce_rows(None, 1, 8, 8, 128, 128, 0, None, None, 1, None, 128, 0.0, 0, 1, None, None) == 1


@pytest.mark.parametrize("core,gscale", [("exact", 1.0), ("split", 1.0), ("split", 1e-2), ("split", 1e3)])
@pytest.mark.parametrize("b,heads,t,d", [(2, 8, 256, 64), (2, 4, 64, 32), (1, 8, 16, 16), (1, 2, 100, 64)])
def test_attention_backward(b, heads, t, d, core, gscale):
    """exact fp32 core and (round 4) the split-precision core of the f16x3 engine; the latter also with output gradients
    of different magnitudes inside fp16's range (where the backward program's power-of-two loss scale keeps them, like
    every gradient the f16x3 dgrad launches split): dS has no a-priori range, the kernel carries it times a running
    power of two"""
    L, lib = _lib()
    bwd = lib.sgd_attention_bwd if core == "exact" else lib.sgd_attention_bwd_split
    g = torch.Generator().manual_seed(15)
    ch = heads * d
    qkv = torch.randn(b, t, 3 * ch, generator=g).requires_grad_(True)        # channel-last legacy layout
    x = qkv.reshape(b, t, heads, 3, d)
    q, k, v = x[:, :, :, 0].permute(0, 2, 1, 3), x[:, :, :, 1].permute(0, 2, 1, 3), x[:, :, :, 2].permute(0, 2, 1, 3)
    w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), -1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(b, t, ch)
    go = torch.randn(b, t, ch, generator=g) * gscale
    o.backward(go)
    qd = qkv.detach().cuda()
    out = torch.empty(b, t, ch, device="cuda")
    lse = torch.empty(b, heads, t, device="cuda")
    kp, vp = C.c_void_p(qd.data_ptr() + 4 * d), C.c_void_p(qd.data_ptr() + 8 * d)
    L.check(lib.sgd_attention(_p(qd), 3 * ch, 3 * d, kp, vp, 3 * ch, 3 * d, b, heads, t, t, d, 1 / math.sqrt(d),
                              _p(out), ch, _p(lse), _stream()), "attn")
    assert max_rel(out.cpu(), o.detach()) < 3e-6
    gqkv = torch.full((b, t, 3 * ch), float("nan"), device="cuda")
    dvec = torch.empty(b, heads, t, device="cuda")
    god = go.cuda()
    L.check(bwd(_p(qd), 3 * ch, 3 * d, kp, vp, 3 * ch, 3 * d, _p(out), ch, _p(god), ch, _p(lse),
                _p(dvec), b, heads, t, t, d, 1 / math.sqrt(d), _p(gqkv),
                C.c_void_p(gqkv.data_ptr() + 4 * d), C.c_void_p(gqkv.data_ptr() + 8 * d), _stream()),
            "attn_bwd")
    assert torch.isfinite(gqkv).all()
    x3 = gqkv.cpu().reshape(b, t, heads, 3, d)
    r3 = qkv.grad.reshape(b, t, heads, 3, d)
    for i, nm in enumerate("qkv"):                       # each of dq / dk / dv against ITS OWN maximum
        assert max_rel(x3[:, :, :, i], r3[:, :, :, i]) < 1e-5, (core, nm)


def test_attention_backward_128_wide_heads():
    """config/dynamic/unet_fast_s64.yaml: 1024 channels / 8 heads at 16 x 16 -- the exact kernels' D = 128 instances (32-row LDS
    tiles); the split-precision backward has none and refuses"""
    test_attention_backward(2, 8, 256, 128, "exact", 1.0)
    test_attention_backward(1, 2, 70, 128, "exact", 1.0)
    L, lib = _lib()
    assert lib.sgd_attention_bwd_split(None, 0, 0, None, None, 0, 0, None, 0, None, 0, None, None, 1, 1, 1, 1, 128, 1.0, None, None,
                                       None, None) != 0


def test_q_sample_and_mse_loss():
    L, lib = _lib()
    from oracle import diffusion_ref as D
    g = torch.Generator().manual_seed(16)
    b, c, hw = 5, 3, 64
    x0, noise = torch.randn(b, c, 8, 8, generator=g), torch.randn(b, c, 8, 8, generator=g)
    t = torch.tensor([0, 999, 500, 1, 37])
    sched = D.make_schedule()
    ref = D.q_sample(sched, x0, t, noise)
    out = torch.empty(b, c, 8, 8, device="cuda")
    sa, s1 = sched["sqrt_alphas_cumprod"].cuda(), sched["sqrt_one_minus_alphas_cumprod"].cuda()
    x0d, nd, td = x0.cuda(), noise.cuda(), t.cuda()
    L.check(lib.sgd_q_sample(_p(x0d), _p(nd), _p(td), _p(sa), _p(s1), b, c * hw, _p(out), _stream()), "q_sample")
    assert max_rel(out.cpu(), ref) < 1e-6
    eps = torch.randn(b, c, 8, 8, generator=g).requires_grad_(True)
    per = ((noise - eps) ** 2).reshape(b, -1).mean(1)
    per.mean().backward()
    ed = _nhwc(eps.detach()).cuda()
    ps, ge = torch.empty(b, device="cuda"), torch.empty(b, 8, 8, c, device="cuda")
    L.check(lib.sgd_mse_loss(_p(ed), _p(nd), b, c, hw, _p(ps), _p(ge), _stream()), "mse")
    assert max_rel(ps.cpu(), per.detach()) < 1e-6
    assert max_rel(ge.cpu().permute(0, 3, 1, 2), eps.grad) < 1e-6


@pytest.mark.parametrize("core", ["exact", "split"])
def test_attention_backward_multiquery(core):
    """Attention_LR core (crossattetion_lr.py:115-137): 8 heads share K/V of 16+1+256 rows; dK/dV summed over heads"""
    L, lib = _lib()
    bwd = lib.sgd_attention_bwd if core == "exact" else lib.sgd_attention_bwd_split
    g = torch.Generator().manual_seed(17)
    b, heads, t, j, d = 2, 8, 256, 273, 64
    q = torch.randn(b, t, heads * d, generator=g).requires_grad_(True)
    kv = torch.randn(b, j, 2 * d, generator=g).requires_grad_(True)
    qh = q.reshape(b, t, heads, d).permute(0, 2, 1, 3) * d ** -0.5
    attn = torch.einsum("bhid,bjd->bhij", qh, kv[..., :d]).softmax(-1)
    o = torch.einsum("bhij,bjd->bhid", attn, kv[..., d:]).permute(0, 2, 1, 3).reshape(b, t, heads * d)
    go = torch.randn(b, t, heads * d, generator=g)
    o.backward(go)
    qd, kvd, god = q.detach().cuda(), kv.detach().cuda(), go.cuda()
    out = torch.empty(b, t, heads * d, device="cuda")
    lse = torch.empty(b, heads, t, device="cuda")
    vp = C.c_void_p(kvd.data_ptr() + 4 * d)
    L.check(lib.sgd_attention(_p(qd), heads * d, d, _p(kvd), vp, 2 * d, 0, b, heads, t, j, d, d ** -0.5, _p(out),
                              heads * d, _p(lse), _stream()), "attn")
    gq = torch.full((b, t, heads * d), float("nan"), device="cuda")
    gkv = torch.full((b, j, 2 * d), float("nan"), device="cuda")
    dvec = torch.empty(b, heads, t, device="cuda")
    L.check(bwd(_p(qd), heads * d, d, _p(kvd), vp, 2 * d, 0, _p(out), heads * d, _p(god), heads * d,
                _p(lse), _p(dvec), b, heads, t, j, d, d ** -0.5, _p(gq), _p(gkv),
                C.c_void_p(gkv.data_ptr() + 4 * d), _stream()), "attn_bwd")
    assert max_rel(gq.cpu(), q.grad) < 1e-5
    assert max_rel(gkv.cpu()[..., :d], kv.grad[..., :d]) < 1e-5
    assert max_rel(gkv.cpu()[..., d:], kv.grad[..., d:]) < 1e-5


@pytest.mark.parametrize("rows,c", [(300, 512), (48, 32)])
def test_layernorm_backward(rows, c):
    L, lib = _lib()
    g = torch.Generator().manual_seed(18)
    x = (torch.randn(rows, c, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = torch.randn(c, generator=g).requires_grad_(True)
    beta = torch.randn(c, generator=g).requires_grad_(True)
    gy = torch.randn(rows, c, generator=g)
    F.layer_norm(x, (c,), gamma, beta, 1e-5).backward(gy)
    xd, gd, gmd = x.detach().cuda(), gy.cuda(), gamma.detach().cuda()
    dx = torch.ones(rows, c, device="cuda")
    gxh = torch.empty(rows, c, device="cuda")
    L.check(lib.sgd_ln_bwd(_p(xd), _p(gd), _p(gmd), rows, c, 1e-5, _p(dx), 1, _p(gxh), None, _stream()), "ln_bwd")
    assert max_rel(dx.cpu() - 1, x.grad) < 2e-5
    assert max_rel(gxh.cpu().sum(0), gamma.grad) < 2e-5


@pytest.mark.parametrize("prec,tol", [("f32", 2e-6), ("f16x3", 2e-5)])
def test_strided_conv_dgrad_by_zero_insertion(prec, tol):
    """Downsample conv (stride 2, openaimodel_ca.py:167-174): input gradient = forward kernel on adjoint weights over
    the zero-upsampled output gradient"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(19)
    n, c, h = 2, 64, 16
    x = torch.randn(n, c, h, h, generator=g, requires_grad=True)
    w = (torch.randn(c, c, 3, 3, generator=g) / math.sqrt(c * 9)).requires_grad_(True)
    gy = torch.randn(n, c, h // 2, h // 2, generator=g)
    F.conv2d(x, w, stride=2, padding=1).backward(gy)
    p = L.PREC_BY_NAME[prec]
    wd = w.detach().cuda()
    buf = torch.empty(lib.sgd_packed_weight_bytes(c, c, 3, p) // 4, device="cuda")
    cp, op = C.c_int32(), C.c_int32()
    L.check(lib.sgd_pack_weight_dgrad(_p(wd), _p(buf), c, c, 3, p, C.byref(cp), C.byref(op), _stream()), "packT")
    gyd = _nhwc(gy).cuda()
    out = torch.full((n, h, h, c), float("nan"), device="cuda")
    a = _igemm_args(L, gyd, conv=(n, h // 2, h // 2, h, h), resample=L.RS_ZEROUP2)
    a.w, a.cin_p, a.cout_p, a.y, a.cout, a.y_ld, a.prec = buf.data_ptr(), cp.value, op.value, out.data_ptr(), c, c, p
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "dgrad_s2")
    assert max_rel(out.cpu().permute(0, 3, 1, 2), x.grad) < tol
    # and its weight gradient (stride 2 in the wgrad loader)
    xd = _nhwc(x.detach()).cuda()
    fwd = _igemm_args(L, xd, conv=(n, h, h, h // 2, h // 2), stride=2)
    dw = _wgrad(L, lib, fwd, gyd, c, c, 9, 2)
    assert max_rel(dw.reshape(c, c, 3, 3), w.grad) < 5e-6


def test_resample_adjoints():
    L, lib = _lib()
    g = torch.Generator().manual_seed(20)
    n, c, h = 2, 32, 8
    x = torch.randn(n, c, h, h, generator=g, requires_grad=True)
    gu = torch.randn(n, c, 2 * h, 2 * h, generator=g)
    F.interpolate(x, scale_factor=2, mode="nearest").backward(gu)
    dst = torch.zeros(n, h, h, c, device="cuda")
    gud = _nhwc(gu).cuda()
    L.check(lib.sgd_resample_bwd(_p(gud), n, h, h, c, L.RS_UP2, _p(dst), 0, _stream()), "up_adj")
    assert max_rel(dst.cpu().permute(0, 3, 1, 2), x.grad) < 2e-6
    x.grad = None
    gp = torch.randn(n, c, h // 2, h // 2, generator=g)
    F.avg_pool2d(x, 2).backward(gp)
    gpd = _nhwc(gp).cuda()
    L.check(lib.sgd_resample_bwd(_p(gpd), n, h, h, c, L.RS_AVGPOOL2, _p(dst), 0, _stream()), "pool_adj")
    assert max_rel(dst.cpu().permute(0, 3, 1, 2), x.grad) < 2e-6


@pytest.mark.parametrize("prec,tol", [("f16x3", 2e-5), ("bf16x3", 3e-4)])
@pytest.mark.parametrize("xs,ws", [(1e3, 1e-4), (1.0, 1e-6), (3e-3, 30.0), (1.0, 1.0)])
def test_split_precision_out_of_range_magnitudes(xs, ws, prec, tol):
    """f16x3 outside the comfortable range: activations x1e3 / weights x1e-4 (and friends).  fp16 halves go subnormal
    below 6e-5 (2-3 significant bits left in `lo`) and overflow above 65504; the packed weights therefore carry a
    per-tensor power-of-two scale (sgd_pack_weight_scaled, undone exactly in the epilogue) and the training program
    scales gradients the same way.  Forward conv, its dgrad (adjoint-packed weights) and the split-precision wgrad vs
    float64; the (1, 1) row is the in-range control."""
    L, lib = _lib()
    n, cin, h, cout = 4, 128, 16, 128
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(n, cin, h, h, generator=g) * xs).double().requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9) * ws).double().requires_grad_(True)
    gy = torch.randn(n, cout, h, h, generator=g).double()
    y = F.conv2d(x, w, padding=1)
    y.backward(gy)
    p = L.PREC_BY_NAME[prec]

    def pack(transpose):
        co, ci = (cin, cout) if transpose else (cout, cin)
        buf = torch.empty(lib.sgd_packed_weight_bytes(co, ci, 3, p) // 4, device="cuda")
        amax = torch.zeros(1, dtype=torch.int32, device="cuda")
        sinv = torch.ones(1, device="cuda")
        cp, op = C.c_int32(), C.c_int32()
        wd = w.detach().float().cuda()
        L.check(lib.sgd_weight_amax(_p(wd), wd.numel(), _p(amax), _stream()), "amax")
        L.check(lib.sgd_pack_weight_scaled(_p(wd), _p(buf), cout, cin, 3, p, int(transpose), _p(amax), _p(sinv), C.byref(cp),
                                           C.byref(op), _stream()), "pack")
        return buf, sinv, cp.value, op.value
    # forward
    buf, sinv, cp, op = pack(False)
    xd = _nhwc(x.detach().float()).cuda()
    out = torch.full((n, h, h, cout), float("nan"), device="cuda")
    a = _igemm_args(L, xd, conv=(n, h, h, h, h))
    a.w, a.cin_p, a.cout_p, a.y, a.cout, a.y_ld, a.prec, a.w_scale_inv = buf.data_ptr(), cp, op, out.data_ptr(), cout, cout, p, sinv.data_ptr()
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "fwd")
    assert max_rel(out.cpu().permute(0, 3, 1, 2), y.detach().float()) < tol
    s = float(sinv.item())
    assert s == 2.0 ** round(math.log2(s)) and 1.0 <= float(w.detach().abs().max()) / s < 2.0      # amax * 2^k in [1, 2)
    # dgrad (gradients of O(1): the training program keeps them there with its own power-of-two loss scale)
    bufT, sinvT, cpT, opT = pack(True)
    gyd = _nhwc(gy.float()).cuda()
    gx = torch.full((n, h, h, cin), float("nan"), device="cuda")
    b = _igemm_args(L, gyd, conv=(n, h, h, h, h))
    b.w, b.cin_p, b.cout_p, b.y, b.cout, b.y_ld, b.prec, b.w_scale_inv = bufT.data_ptr(), cpT, opT, gx.data_ptr(), cin, cin, p, sinvT.data_ptr()
    L.check(lib.sgd_igemm(C.byref(b), _stream()), "dgrad")
    assert max_rel(gx.cpu().permute(0, 3, 1, 2), x.grad.float()) < tol
    # wgrad reads the raw activations: their magnitude is the tensor's own (activations x1e3 stay below fp16's 65504)
    fwd = _igemm_args(L, xd, conv=(n, h, h, h, h))
    fwd.prec = p
    dw = _wgrad(L, lib, fwd, gyd, cout, cin, 9, 3)
    assert max_rel(dw.reshape(cout, cin, 3, 3), w.grad.float()) < max(tol, 3e-5)
