"""Kernel-level parity of the C-ABI entry points (through ctypes) against plain fp32 torch-CPU
restatements of the same op.  GPU only."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import max_rel

pytestmark = pytest.mark.gpu

PRECS = [("f32", 2e-6), ("f16x3", 2e-5), ("bf16x3", 1e-4)]


def _lib():
    from sgdm_amd import _lib as L
    return L, L.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _pack(w, ks, prec):
    L, lib = _lib()
    cout, cin = w.shape[0], w.shape[1]
    nbytes = lib.sgd_packed_weight_bytes(cout, cin, ks, prec)
    buf = torch.empty(nbytes // 4, device="cuda")
    cin_p, cout_p = C.c_int32(), C.c_int32()
    L.check(lib.sgd_pack_weight(_p(w.contiguous()), _p(buf), cout, cin, ks, prec, C.byref(cin_p), C.byref(cout_p),
                                _stream()), "pack")
    return buf, cin_p.value, cout_p.value


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _run_conv(x, w, bias, prec_name, stride=1, resample=0, x1=None, pa=None, pb=None, silu=0, res=None, res_mode=0):
    """x, x1: NCHW cpu tensors. returns NCHW cpu result of the HIP kernel."""
    L, lib = _lib()
    prec = L.PREC_BY_NAME[prec_name]
    n, c0, hi, wi = x.shape
    c1 = x1.shape[1] if x1 is not None else 0
    hc, wc = (hi // 2, wi // 2) if resample == 1 else ((hi * 2, wi * 2) if resample == 2 else (hi, wi))
    ho, wo = (hc // 2, wc // 2) if stride == 2 else (hc, wc)
    cout = w.shape[0]
    wd = w.cuda()
    buf, cin_p, cout_p = _pack(wd, 3, prec)
    xd = _nhwc(x).cuda()
    x1d = _nhwc(x1).cuda() if x1 is not None else None
    y = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
    a = L.IgemmArgs()
    a.x0, a.x1, a.c0, a.c1 = xd.data_ptr(), (x1d.data_ptr() if x1d is not None else 0), c0, c1
    a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride, a.resample = L.MODE_CONV3, n, hi, wi, ho, wo, stride, resample
    keep = []
    if pa is not None:
        pad, pbd = pa.cuda(), pb.cuda()
        keep += [pad, pbd]
        a.pro, a.pa, a.pb = L.PRO_AFFINE_NC, pad.data_ptr(), pbd.data_ptr()
    a.pro_silu = silu
    a.w, a.cin_p, a.cout_p = buf.data_ptr(), cin_p, cout_p
    bd = bias.cuda() if bias is not None else None
    a.bias = bd.data_ptr() if bd is not None else 0
    rd = _nhwc(res).cuda() if res is not None else None
    a.res, a.res_mode = (rd.data_ptr() if rd is not None else 0), res_mode
    a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout, cout, prec
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
    torch.cuda.synchronize()
    return y.cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("prec,tol", PRECS)
@pytest.mark.parametrize("shape", [(2, 32, 16, 16, 128), (3, 64, 8, 8, 64), (1, 3, 16, 16, 32), (2, 96, 4, 4, 3),
                                   (1, 128, 32, 32, 128), (5, 30, 8, 8, 128)])
def test_conv3x3_plain(shape, prec, tol):
    n, cin, h, w_, cout = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w_, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    got = _run_conv(x, w, b, prec)
    assert max_rel(got, ref) < tol


@pytest.mark.parametrize("prec,tol", PRECS)
def test_conv3x3_fused_prologue_epilogue(prec, tol):
    """GN-apply(+FiLM)+SiLU prologue, virtual concat, residual epilogue (ResBlock conv, openaimodel.py:300-320)"""
    g = torch.Generator().manual_seed(2)
    n, c0, c1, h, cout = 3, 64, 32, 16, 128
    x0, x1 = torch.randn(n, c0, h, h, generator=g), torch.randn(n, c1, h, h, generator=g)
    pa, pb = torch.randn(n, c0 + c1, generator=g), torch.randn(n, c0 + c1, generator=g)
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / math.sqrt((c0 + c1) * 9)
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, h, generator=g)
    xin = torch.cat([x0, x1], 1)
    act = F.silu(xin * pa[:, :, None, None] + pb[:, :, None, None])
    ref = F.conv2d(act, w, b, padding=1) + res
    got = _run_conv(x0, w, b, prec, x1=x1, pa=pa, pb=pb, silu=1, res=res)
    assert max_rel(got, ref) < tol


@pytest.mark.parametrize("prec,tol", PRECS[:2])
@pytest.mark.parametrize("mode", ["down", "up", "stride2", "upconv"])
def test_conv3x3_resample(mode, prec, tol):
    g = torch.Generator().manual_seed(3)
    n, c, h = 2, 64, 16
    x = torch.randn(n, c, h, h, generator=g)
    pa, pb = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) / math.sqrt(c * 9)
    b = torch.randn(c, generator=g)
    act = F.silu(x * pa[:, :, None, None] + pb[:, :, None, None])
    if mode == "down":       # ResBlock(down=True): conv(avgpool(act)) + avgpool(x)
        ref = F.conv2d(F.avg_pool2d(act, 2), w, b, padding=1) + F.avg_pool2d(x, 2)
        got = _run_conv(x, w, b, prec, resample=1, pa=pa, pb=pb, silu=1, res=x, res_mode=1)
    elif mode == "up":       # ResBlock(up=True)
        ref = F.conv2d(F.interpolate(act, scale_factor=2, mode="nearest"), w, b, padding=1) \
            + F.interpolate(x, scale_factor=2, mode="nearest")
        got = _run_conv(x, w, b, prec, resample=2, pa=pa, pb=pb, silu=1, res=x, res_mode=2)
    elif mode == "stride2":  # Downsample conv (openaimodel_ca.py:167-174)
        ref = F.conv2d(x, w, b, stride=2, padding=1)
        got = _run_conv(x, w, b, prec, stride=2)
    else:                    # Upsample nearest + conv (openaimodel_ca.py:128-131)
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
        got = _run_conv(x, w, b, prec, resample=2)
    assert got.shape == ref.shape
    assert max_rel(got, ref) < tol


@pytest.mark.parametrize("prec,tol", PRECS)
@pytest.mark.parametrize("m,k,nout", [(160, 128, 512), (7, 5000, 256), (300, 96, 96), (1024, 512, 1536)])
def test_linear_flat(m, k, nout, prec, tol):
    L, lib = _lib()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(nout, k, generator=g) / math.sqrt(k)
    b = torch.randn(nout, generator=g)
    ref = F.linear(F.silu(x), w, b)
    p = L.PREC_BY_NAME[prec]
    buf, cin_p, cout_p = _pack(w.cuda(), 1, p)
    xd, bd = x.cuda(), b.cuda()
    y = torch.full((m, nout), float("nan"), device="cuda")
    a = L.IgemmArgs()
    a.x0, a.c0, a.mode, a.m, a.stride, a.pro_silu = xd.data_ptr(), k, L.MODE_FLAT, m, 1, 1
    a.w, a.cin_p, a.cout_p, a.bias = buf.data_ptr(), cin_p, cout_p, bd.data_ptr()
    a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), nout, nout, p
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
    assert max_rel(y.cpu(), ref) < tol


@pytest.mark.parametrize("n,h,w_,cin,cout,pad", [(5, 64, 64, 3, 128, 0), (2, 32, 32, 4, 224, 0), (3, 16, 16, 3, 256, 8), (2, 8, 8, 4, 1024, 0),
                                                  (1, 4, 4, 3, 32, 4), (2, 16, 64, 3, 64, 0), (3, 6, 10, 4, 128, 0)])
def test_stem_conv_narrow_kernel(n, h, w_, cin, cout, pad):
    """sgd_conv3_narrow_in (round 4): the UNet stem as a plain fp32 kernel -- output against float64 conv2d, the GroupNorm
    partial statistics against the exact sums of the output it wrote, a strided destination left untouched outside its
    columns; maps that are not powers of two included (this kernel has no tile geometry to satisfy)"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(71)
    x = torch.randn(n, cin, h, w_, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    xd, wd, bd = _nhwc(x).cuda(), wt.cuda(), b.cuda()
    y_ld = cout + pad
    y = torch.full((n, h, w_, y_ld), float("nan"), device="cuda")
    parts = lib.sgd_conv3_narrow_in_parts(h, w_)
    assert parts >= 1
    st = torch.full((n, parts, 2, cout), float("nan"), device="cuda")
    L.check(lib.sgd_conv3_narrow_in(_p(xd), _p(wd), _p(bd), _p(y), _p(st), n, h, w_, cin, cout, y_ld, 0, _stream()), "stem")
    torch.cuda.synchronize()
    got = y[..., :cout].cpu().permute(0, 3, 1, 2)
    assert max_rel(got, ref) < 2e-6
    if pad:
        assert torch.isnan(y[..., cout:]).all()
    yy = y[..., :cout].double().reshape(n, h * w_, cout).cpu()
    exact = torch.stack([yy.sum(1), (yy * yy).sum(1)], 1)                     # [n, 2, cout]
    folded = st.cpu().double().sum(1)
    assert float((folded - exact).abs().max() / exact.abs().max()) < 2e-6
    # without statistics and without bias
    y2 = torch.full((n, h, w_, y_ld), float("nan"), device="cuda")
    L.check(lib.sgd_conv3_narrow_in(_p(xd), _p(wd), None, _p(y2), None, n, h, w_, cin, cout, y_ld, 0, _stream()), "stem")
    ref2 = F.conv2d(x.double(), wt.double(), None, padding=1)
    assert max_rel(y2[..., :cout].cpu().permute(0, 3, 1, 2), ref2) < 2e-6
    assert lib.sgd_conv3_narrow_in(_p(xd), _p(wd), None, _p(y2), None, n, h, w_, 5, cout, y_ld, 0, _stream()) == 1     # cin 3 / 4 only
    # adjoint: the input gradient of a conv with `cin` OUTPUT channels (the head) -- x plays the output gradient
    wh = (torch.randn(cin, cout, 3, 3, generator=g) / math.sqrt(cout * 9)).double()
    u = torch.randn(n, cout, h, w_, generator=g, dtype=torch.float64, requires_grad=True)
    F.conv2d(u, wh, padding=1).backward(x.double())
    whd = wh.float().cuda()
    y3 = torch.full((n, h, w_, y_ld), float("nan"), device="cuda")
    L.check(lib.sgd_conv3_narrow_in(_p(xd), _p(whd), None, _p(y3), None, n, h, w_, cin, cout, y_ld, 1, _stream()), "head dgrad")
    assert max_rel(y3[..., :cout].cpu().permute(0, 3, 1, 2), u.grad) < 2e-6


def _random_conv_cases(count, seed):
    """random but valid conv launches: channel counts off the 4 / 32 grid, concat, prologue, every resample / residual mode,
    few blocks (grid_cap: many tiles per block, partial last rounds -> the balanced tail) and capped grids"""
    import random
    rnd = random.Random(seed)
    cases = []
    while len(cases) < count:
        kind = rnd.choice(["plain", "plain", "res", "down", "up", "stride2", "upconv"])
        n = rnd.randint(1, 5)
        h, w_ = rnd.choice([4, 8, 16, 32]), rnd.choice([4, 8, 16, 32])      # (sgd_igemm: power-of-two output maps)
        c0 = rnd.choice([3, 4, 20, 32, 33, 64, 96, 100, 128, 160, 200])
        c1 = rnd.choice([0, 0, 0, 4, 32, 36]) if c0 % 32 == 0 else 0
        cout = rnd.choice([3, 4, 30, 32, 64, 100, 128, 132, 256, 300])
        pro = rnd.random() < 0.6
        silu = int(rnd.random() < 0.6)
        if kind in ("down", "up"):
            c1 = 0
            cout = c0                      # the resampled identity skip adds x itself
            pro = True
        if n * h * w_ * max(c0 + c1, cout) > 3_000_000:     # keep the float64 reference quick
            continue
        cases.append(dict(kind=kind, n=n, h=h, w=w_, c0=c0, c1=c1, cout=cout, pro=pro, silu=silu,
                          max_grid=rnd.choice([0, 0, 8, 16, 24, 40]), grid_cap=rnd.choice([0, 0, 0, 8, 72, 200]),
                          work=rnd.random() < 0.7, seed=rnd.randint(0, 1 << 30)))
    return cases


@pytest.mark.parametrize("prec,tol", [("f32", 4e-6), ("f16x3", 4e-5)])
def test_conv3x3_random_configurations(monkeypatch, prec, tol):
    """48 seeded random launches against float64: the schedule paths a fixed shape list does not reach (tiles per block,
    partial rounds with and without the workspace, capped grids) crossed with the loader / epilogue modes"""
    L, lib = _lib()
    p = L.PREC_BY_NAME[prec]
    work = torch.zeros(int(lib.sgd_igemm_work_bytes()) // 4, device="cuda")
    for ci, cs in enumerate(_random_conv_cases(48, 20260104)):
        g = torch.Generator().manual_seed(cs["seed"])
        n, h, w_, c0, c1, cout, kind = cs["n"], cs["h"], cs["w"], cs["c0"], cs["c1"], cs["cout"], cs["kind"]
        cin = c0 + c1
        x = torch.randn(n, cin, h, w_, generator=g, dtype=torch.float64)
        wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / math.sqrt(cin * 9)
        b = torch.randn(cout, generator=g, dtype=torch.float64)
        pa = torch.randn(n, cin, generator=g, dtype=torch.float64)
        pb = torch.randn(n, cin, generator=g, dtype=torch.float64)
        act = x
        if cs["pro"]:
            act = act * pa[:, :, None, None] + pb[:, :, None, None]
        if cs["silu"]:
            act = F.silu(act)
        resample, stride, res, res_mode = 0, 1, None, 0
        if kind == "down":
            ref = F.conv2d(F.avg_pool2d(act, 2), wt, b, padding=1) + F.avg_pool2d(x, 2)
            resample, res, res_mode = 1, x, 1
        elif kind == "up":
            ref = F.conv2d(F.interpolate(act, scale_factor=2, mode="nearest"), wt, b, padding=1) \
                + F.interpolate(x, scale_factor=2, mode="nearest")
            resample, res, res_mode = 2, x, 2
        elif kind == "stride2":
            ref = F.conv2d(act, wt, b, stride=2, padding=1)
            stride = 2
        elif kind == "upconv":
            ref = F.conv2d(F.interpolate(act, scale_factor=2, mode="nearest"), wt, b, padding=1)
            resample = 2
        else:
            ref = F.conv2d(act, wt, b, padding=1)
            if kind == "res":
                res = torch.randn(ref.shape, generator=g, dtype=torch.float64)
                ref = ref + res
        # ---- the launch
        hc, wc = (h // 2, w_ // 2) if resample == 1 else ((h * 2, w_ * 2) if resample == 2 else (h, w_))
        ho, wo = ((hc + 1) // 2, (wc + 1) // 2) if stride == 2 else (hc, wc)
        assert ref.shape[2:] == (ho, wo), (cs, ref.shape)
        buf, cin_p, cout_p = _pack(wt.float().cuda(), 3, p)
        x0d = _nhwc(x[:, :c0].float()).cuda()
        x1d = _nhwc(x[:, c0:].float()).cuda() if c1 else None
        pad, pbd, bd = pa.float().cuda(), pb.float().cuda(), b.float().cuda()
        rd = _nhwc(res.float()).cuda() if res is not None else None
        y = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
        a = L.IgemmArgs()
        a.x0, a.x1, a.c0, a.c1 = x0d.data_ptr(), (x1d.data_ptr() if c1 else 0), c0, c1
        a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride, a.resample = L.MODE_CONV3, n, h, w_, ho, wo, stride, resample
        if cs["pro"]:
            a.pro, a.pa, a.pb = L.PRO_AFFINE_NC, pad.data_ptr(), pbd.data_ptr()
        a.pro_silu = cs["silu"]
        a.w, a.cin_p, a.cout_p, a.bias = buf.data_ptr(), cin_p, cout_p, bd.data_ptr()
        a.res, a.res_mode = (rd.data_ptr() if rd is not None else 0), res_mode
        a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout, cout, p
        # (two draws kept from the time one of them travelled through the environment: the smaller cap applies)
        a.grid_cap = min([v for v in (cs["grid_cap"], cs["max_grid"]) if v] or [0])
        if cs["work"]:
            a.work, a.work_bytes = work.data_ptr(), work.numel() * 4
        L.check(lib.sgd_igemm(C.byref(a), _stream()), f"igemm case {ci}: {cs}")
        torch.cuda.synchronize()
        got = y.cpu().permute(0, 3, 1, 2).double()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < tol, (ci, cs, err)
    off = int(lib.sgd_igemm_work_status_offset())
    assert int(work.view(torch.int32)[off // 4]) == 0, "a finisher's bounded poll expired"


@pytest.mark.parametrize("prec,tol", [("f32", 4e-6), ("f16x3", 4e-5)])
def test_linear_random_configurations(monkeypatch, prec, tol):
    """48 seeded random 1x1 / linear launches against float64: no / per-image affine / LayerNorm-row prologue, SiLU,
    concat, residual, strided and row-remapped outputs, few blocks and capped grids, with and without the workspace"""
    import random
    L, lib = _lib()
    p = L.PREC_BY_NAME[prec]
    work = torch.zeros(int(lib.sgd_igemm_work_bytes()) // 4, device="cuda")
    rnd = random.Random(20260105)
    for ci in range(48):
        g = torch.Generator().manual_seed(rnd.randint(0, 1 << 30))
        pro = rnd.choice(["none", "none", "affine", "affine", "ln"])
        rows_per_n = rnd.choice([16, 64, 100, 128, 256])
        n = rnd.randint(1, 6)
        m = n * rows_per_n if pro == "affine" else rnd.choice([7, 80, 128, 333, 1024, 1500])
        c0 = rnd.choice([4, 20, 32, 64, 96, 128, 200, 256, 512])
        if pro == "ln" or rnd.random() < 0.3:
            c0 = rnd.choice([32, 64, 128, 512])
        c1 = rnd.choice([0, 0, 32, 36]) if (c0 % 32 == 0 and pro != "ln") else 0
        cin = c0 + c1
        cout = rnd.choice([3, 32, 64, 100, 128, 256, 384, 512, 1536])
        silu = int(rnd.random() < 0.5)
        x = torch.randn(m, cin, generator=g, dtype=torch.float64) * 1.5 + 0.2
        wt = torch.randn(cout, cin, generator=g, dtype=torch.float64) / math.sqrt(cin)
        b = torch.randn(cout, generator=g, dtype=torch.float64)
        act = x
        a = L.IgemmArgs()
        keep = []
        if pro == "affine":
            pa = torch.randn(n, cin, generator=g, dtype=torch.float64)
            pb = torch.randn(n, cin, generator=g, dtype=torch.float64)
            act = (x.reshape(n, rows_per_n, cin) * pa[:, None] + pb[:, None]).reshape(m, cin)
            pad, pbd = pa.float().cuda(), pb.float().cuda()
            keep += [pad, pbd]
            a.pro, a.pa, a.pb, a.rows_per_n = L.PRO_AFFINE_NC, pad.data_ptr(), pbd.data_ptr(), rows_per_n
        elif pro == "ln":
            gamma = torch.randn(cin, generator=g, dtype=torch.float64)
            beta = torch.randn(cin, generator=g, dtype=torch.float64) if rnd.random() < 0.7 else None
            act = F.layer_norm(x, (cin,), gamma, beta, 1e-5)
        if silu:
            act = F.silu(act)
        ref = act @ wt.t() + b
        res = torch.randn(m, cout, generator=g, dtype=torch.float64) if rnd.random() < 0.4 else None
        if res is not None:
            ref = ref + res
        # ---- the launch
        x0d = x[:, :c0].float().contiguous().cuda()
        x1d = x[:, c0:].float().contiguous().cuda() if c1 else None
        if pro == "ln":
            st = torch.empty(m, 2, device="cuda")
            L.check(lib.sgd_ln_stats(_p(x0d), m, cin, C.c_float(1e-5), _p(st), _stream()), "ln_stats")
            gd = gamma.float().cuda()
            btd = beta.float().cuda() if beta is not None else None
            keep += [st, gd, btd]
            a.pro, a.pa, a.pb, a.pc = L.PRO_LN_ROW, st.data_ptr(), gd.data_ptr(), (btd.data_ptr() if btd is not None else 0)
        buf, cin_p, cout_p = _pack(wt.float().reshape(cout, cin, 1, 1).cuda(), 1, p)
        bd = b.float().cuda()
        rd = res.float().cuda() if res is not None else None
        # output: leading dimension and column offset of a wider buffer, optionally rows re-mapped group by group
        remap = rnd.random() < 0.3 and m % 16 == 0 and (cout % 4 == 0)
        y_ld = cout + rnd.choice([0, 0, 4, 64]) if cout % 4 == 0 else cout
        if remap:
            rin, rout, roff = 16, 16 + rnd.choice([1, 3]), rnd.choice([0, 1])
            yrows = m // rin * rout
        else:
            rin = rout = roff = 0
            yrows = m
        y = torch.full((yrows, y_ld), float("nan"), device="cuda")
        a.x0, a.x1, a.c0, a.c1 = x0d.data_ptr(), (x1d.data_ptr() if c1 else 0), c0, c1
        a.mode, a.m, a.stride, a.pro_silu = L.MODE_FLAT, m, 1, silu
        if pro != "affine":
            a.rows_per_n = 0
        a.w, a.cin_p, a.cout_p, a.bias = buf.data_ptr(), cin_p, cout_p, bd.data_ptr()
        a.res = rd.data_ptr() if rd is not None else 0
        a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout, y_ld, p
        a.orows_in, a.orows_out, a.orow_off = rin, rout, roff
        cap = rnd.choice([0, 0, 0, 8, 72, 200])
        if rnd.random() < 0.7:
            a.work, a.work_bytes = work.data_ptr(), work.numel() * 4
        mg = rnd.choice([0, 0, 8, 16, 40])
        a.grid_cap = min([v for v in (cap, mg) if v] or [0])     # (two draws, as when one travelled through the environment)
        desc = f"case {ci}: pro={pro} m={m} c0={c0} c1={c1} cout={cout} silu={silu} res={res is not None} y_ld={y_ld} remap={(rin, rout, roff)} grid_cap={a.grid_cap} max_grid={mg} work={bool(a.work)}"
        L.check(lib.sgd_igemm(C.byref(a), _stream()), desc)
        torch.cuda.synchronize()
        got = y.cpu().double()
        if remap:
            rows = (torch.arange(m) // rin) * rout + roff + torch.arange(m) % rin
            out = got[rows, :cout]
            untouched = torch.ones(yrows, dtype=torch.bool)
            untouched[rows] = False
            assert torch.isnan(got[untouched]).all(), desc
        else:
            out = got[:, :cout]
        assert torch.isnan(got[:, cout:]).all(), desc
        err = float((out - ref).abs().max() / ref.abs().max())
        assert err < tol, (desc, err)
    off = int(lib.sgd_igemm_work_status_offset())
    assert int(work.view(torch.int32)[off // 4]) == 0, "a finisher's bounded poll expired"


def test_igemm_rejects_bad_args():
    L, lib = _lib()
    a = L.IgemmArgs()
    assert lib.sgd_igemm(C.byref(a), _stream()) == 1
    assert lib.sgd_igemm(None, _stream()) == 1


@pytest.mark.parametrize("n,c,hw", [(3, 128, 256), (2, 96, 64), (1, 1024, 16), (2, 32, 4096)])
def test_groupnorm_coefficients(n, c, hw):
    L, lib = _lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, hw, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    film = torch.randn(n, 2 * c + 8, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, 1e-5) * (1 + film[:, :c, None]) + film[:, c:2 * c, None]
    xd = x.permute(0, 2, 1).contiguous().cuda()
    c_half = c // 2
    sums = torch.zeros(n, c, 2, device="cuda")
    # two calls emulate a virtual concat of channel halves: strided view is not allowed, so copy halves
    xa, xb = xd[:, :, :c_half].contiguous(), xd[:, :, c_half:].contiguous()
    L.check(lib.sgd_chan_stats(_p(xa), n, hw, c_half, _p(sums), c, 0, _stream()), "stats")
    L.check(lib.sgd_chan_stats(_p(xb), n, hw, c - c_half, _p(sums), c, c_half, _stream()), "stats")
    a, b = torch.empty(n, c, device="cuda"), torch.empty(n, c, device="cuda")
    fd, gd, bd = film.cuda(), gamma.cuda(), beta.cuda()      # keep the device tensors alive across the launch
    L.check(lib.sgd_gn_coef(_p(sums), _p(gd), _p(bd), _p(fd), 2 * c + 8, n, c, 32, hw, 1e-5,
                            _p(a), _p(b), _stream()), "coef")
    got = x * a.cpu()[:, :, None] + b.cpu()[:, :, None]
    assert max_rel(got, ref) < 5e-6


@pytest.mark.parametrize("rows,c", [(512, 512), (37, 32), (100, 128)])
def test_layernorm(rows, c):
    L, lib = _lib()
    g = torch.Generator().manual_seed(6)
    x, res = torch.randn(rows, c, generator=g) * 3 + 1, torch.randn(rows, c, generator=g)
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    ref = res + F.layer_norm(x, (c,), gamma, beta, 1e-5)
    out = torch.empty(rows, c, device="cuda")
    xd, gd, bd, rd = x.cuda(), gamma.cuda(), beta.cuda(), res.cuda()
    L.check(lib.sgd_ln_apply(_p(xd), _p(gd), _p(bd), _p(rd), rows, c, 1e-5, _p(out), _stream()), "ln")
    assert max_rel(out.cpu(), ref) < 2e-6
    st = torch.empty(rows, 2, device="cuda")
    L.check(lib.sgd_ln_stats(_p(xd), rows, c, 1e-5, _p(st), _stream()), "lnstats")
    assert max_rel(st.cpu()[:, 0], x.mean(1)) < 2e-6
    assert max_rel(st.cpu()[:, 1], 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)) < 2e-6


# the two attention cores: exact fp32 MFMA, and the split-precision one the f16x3 engine launches (three f16 products
# per fp32 product, fp32 accumulate: same tolerance class, measured ~1e-6)
ATTN_CORES = [("sgd_attention", 3e-6), ("sgd_attention_split", 5e-6)]


@pytest.mark.parametrize("core,tol", ATTN_CORES)
@pytest.mark.parametrize("b,heads,t,d", [(2, 8, 256, 64), (3, 4, 64, 32), (2, 8, 16, 16), (1, 2, 200, 64), (2, 2, 100, 128)])
def test_attention_legacy(b, heads, t, d, core, tol):
    """QKVAttentionLegacy (openaimodel.py:403-420) on the channel-last qkv layout"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(7)
    ch = heads * d
    qkv = torch.randn(b, 3 * ch, t, generator=g)
    q, k, v = qkv.reshape(b * heads, 3 * d, t).split(d, dim=1)
    s = 1 / math.sqrt(math.sqrt(d))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s), -1)
    ref = torch.einsum("bts,bcs->bct", w, v).reshape(b, ch, t)
    qd = qkv.permute(0, 2, 1).contiguous().cuda()                       # [b, t, 3ch]
    out = torch.full((b, t, ch), float("nan"), device="cuda")
    lse = torch.empty(b, heads, t, device="cuda")
    L.check(getattr(lib, core)(_p(qd), 3 * ch, 3 * d, C.c_void_p(qd.data_ptr() + 4 * d),
                               C.c_void_p(qd.data_ptr() + 8 * d), 3 * ch, 3 * d, b, heads, t, t, d,
                               1 / math.sqrt(d), _p(out), ch, _p(lse), _stream()), "attn")
    assert max_rel(out.cpu().permute(0, 2, 1), ref) < tol
    want_lse = torch.logsumexp(torch.einsum("bct,bcs->bts", q * s, k * s), -1).reshape(b, heads, t)
    assert (lse.cpu() - want_lse).abs().max() < 2e-5


def test_attention_split_every_row_within_tolerance():
    """regression: with a scale that is not a power of two (d = 32) the q * scale product was split with two different
    `hi` roundings (v_fma_mixlo_f16 at one use, v_cvt_f16_f32 at the other) on 1 element in 2^13 -> one query row in
    ~400 off by an fp16 ulp.  Every row is checked, against float64."""
    L, lib = _lib()
    b, heads, t, d = 4, 4, 512, 32
    ch = heads * d
    for seed in (0, 1):
        g = torch.Generator().manual_seed(seed)
        qkv = torch.randn(b, t, 3 * ch, generator=g)
        v5 = qkv.double().reshape(b, t, heads, 3, d)
        w = torch.softmax(torch.einsum("bthd,bshd->bhts", v5[:, :, :, 0], v5[:, :, :, 1]) * d ** -0.5, -1)
        ref = torch.einsum("bhts,bshd->bthd", w, v5[:, :, :, 2]).reshape(b, t, ch)
        qd = qkv.cuda()
        out = torch.empty(b, t, ch, device="cuda")
        L.check(lib.sgd_attention_split(_p(qd), 3 * ch, 3 * d, C.c_void_p(qd.data_ptr() + 4 * d),
                                        C.c_void_p(qd.data_ptr() + 8 * d), 3 * ch, 3 * d, b, heads, t, t, d,
                                        d ** -0.5, _p(out), ch, None, _stream()), "attn")
        row_err = ((out.cpu().double() - ref).abs() / ref.abs().max()).reshape(b * t, ch).amax(1)
        assert int((row_err > 5e-6).sum()) == 0, float(row_err.max())


@pytest.mark.parametrize("core,tol", ATTN_CORES)
def test_attention_multiquery_273_keys(core, tol):
    """Attention_LR core (crossattetion_lr.py:115-137): 8 heads share one K/V of 16+1+256 rows"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(8)
    b, heads, t, j, d = 2, 8, 256, 273, 64
    q = torch.randn(b, t, heads * d, generator=g)
    kv = torch.randn(b, j, 2 * d, generator=g)
    qh = q.reshape(b, t, heads, d).permute(0, 2, 1, 3) * d ** -0.5
    attn = torch.einsum("bhid,bjd->bhij", qh, kv[..., :d]).softmax(-1)
    ref = torch.einsum("bhij,bjd->bhid", attn, kv[..., d:]).permute(0, 2, 1, 3).reshape(b, t, heads * d)
    qd, kvd = q.cuda(), kv.cuda()
    out = torch.full((b, t, heads * d), float("nan"), device="cuda")
    L.check(getattr(lib, core)(_p(qd), heads * d, d, _p(kvd), C.c_void_p(kvd.data_ptr() + 4 * d), 2 * d, 0, b, heads,
                               t, j, d, d ** -0.5, _p(out), heads * d, None, _stream()), "attn")
    assert max_rel(out.cpu(), ref) < tol


@pytest.mark.parametrize("core,tol", ATTN_CORES)
def test_attention_softmax_large_logits(core, tol):
    """force the online-softmax rescale branch: one key dominates late in the sequence"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(9)
    b, heads, t, d = 1, 1, 128, 32
    q = torch.randn(b, t, d, generator=g)
    k = torch.randn(b, t, d, generator=g)
    v = torch.randn(b, t, d, generator=g)
    k[0, 100] = q[0, 5] * 8          # spike for query 5 in the last key tile
    k[0, 3] = q[0, 70] * 8           # spike in the first key tile
    ref = torch.softmax(q @ k.transpose(1, 2) * d ** -0.5, -1) @ v
    kv = torch.cat([k, v], -1).contiguous().cuda()
    qd = q.cuda()
    out = torch.empty(b, t, d, device="cuda")
    L.check(getattr(lib, core)(_p(qd), d, d, _p(kv), C.c_void_p(kv.data_ptr() + 4 * d), 2 * d, 0, b, heads, t, t, d,
                               d ** -0.5, _p(out), d, None, _stream()), "attn")
    assert max_rel(out.cpu(), ref) < tol


def test_boundary_kernels():
    L, lib = _lib()
    from oracle.unet_ref import timestep_embedding
    g = torch.Generator().manual_seed(10)
    t = torch.tensor([0, 1, 500, 999], dtype=torch.int64)
    for dim in (32, 128):
        out = torch.empty(8, dim, device="cuda")
        from sgdm_amd.unet import timestep_freqs
        td, fd = t.cuda(), timestep_freqs(dim).cuda()
        L.check(lib.sgd_timestep_embedding(_p(td), _p(fd), 4, 8, dim, _p(out), _stream()), "temb")
        ref = timestep_embedding(torch.cat([t, t]), dim)
        assert (out.cpu() - ref).abs().max() < 2e-6            # |cos|,|sin| <= 1: absolute tolerance
    # cond select with int64 one-hot + batch doubling
    cond = F.one_hot(torch.tensor([3, 1]), 5)
    mask = torch.tensor([False, False, True, True])
    null = torch.full((5,), 0.25)
    out = torch.empty(4, 5, device="cuda")
    cd, md, nd = cond.cuda(), mask.cuda().view(torch.uint8), null.cuda()
    L.check(lib.sgd_cond_select(_p(cd), 1, _p(md), _p(nd), 2, 4, 5, _p(out), _stream()), "cond")
    ref = torch.cat([cond.float(), null.expand(2, 5)])
    assert torch.equal(out.cpu(), ref)
    # pack input: x + masked layout, NCHW -> NHWC, doubled
    x = torch.randn(2, 3, 8, 8, generator=g)
    lay = torch.randn(2, 4, 8, 8, generator=g)
    nl = torch.randn(1, 1, 8, 8, generator=g)
    out = torch.empty(4, 8, 8, 7, device="cuda")
    xd, ld, nld = x.cuda(), lay.cuda(), nl.cuda()
    L.check(lib.sgd_pack_input(_p(xd), _p(ld), _p(md), _p(nld), 2, 4, 3, 4, 8, 8, _p(out), _stream()), "pack_input")
    xx, ll = torch.cat([x, x]), torch.cat([lay, lay])
    ref = torch.cat([xx, torch.where(mask[:, None, None, None], nl, ll)], 1).permute(0, 2, 3, 1)
    assert torch.equal(out.cpu(), ref)
    back = torch.empty(4, 7, 8, 8, device="cuda")
    L.check(lib.sgd_nhwc_to_nchw(_p(out), 4, 8, 8, 7, _p(back), _stream()), "nchw")
    assert torch.equal(back.cpu(), ref.permute(0, 3, 1, 2))


@pytest.mark.parametrize("shape", [(3, 16, 16, 64, 128), (2, 32, 32, 32, 256), (2, 16, 16, 128, 32), (40, 32, 32, 64, 128)])
def test_epilogue_statistics_match_chan_stats(shape):
    """sgd_igemm's fused GroupNorm statistics of its output + sgd_stats_reduce == sgd_chan_stats(y)"""
    n, h, w, cin, cout = shape
    L, lib = _lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    res = torch.randn(n, h, w, cout, generator=g).cuda()
    y = torch.empty(n, h, w, cout, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    prec = L.PREC_F32
    buf = torch.empty(lib.sgd_packed_weight_bytes(cout, cin, 3, prec) // 4, device="cuda")
    cp, op = C.c_int32(), C.c_int32()
    L.check(lib.sgd_pack_weight(C.c_void_p(wt.data_ptr()), C.c_void_p(buf.data_ptr()), cout, cin, 3, prec, C.byref(cp),
                                C.byref(op), st), "pack")
    a = L.IgemmArgs()
    a.x0, a.c0, a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride = x.data_ptr(), cin, L.MODE_CONV3, n, h, w, h, w, 1
    a.w, a.cin_p, a.cout_p, a.bias, a.res = buf.data_ptr(), cp.value, op.value, bias.data_ptr(), res.data_ptr()
    a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout, cout, prec
    parts = lib.sgd_igemm_stats_parts(C.byref(a))
    # one slot per compute-wave row of a 128-row tile: one row of waves on the 128- / 256-column tile, four on the 32-column
    # instance -- which also serves multiples of 128 channels when the launch is small (round 6: at most 64 tiles of 128 columns)
    small = (n * h * w // 128) * ((cout + 127) // 128) * 4 <= 256
    assert parts == (h * w // 128) * (1 if cout % 128 == 0 and not small else 4)
    a.tune = L.TUNE_NO_SMALL
    assert lib.sgd_igemm_stats_parts(C.byref(a)) == (h * w // 128) * (1 if cout % 128 == 0 else 4)
    a.tune = 0
    partial = torch.full((n, parts, 2, cout), float("nan"), device="cuda")
    a.stats = partial.data_ptr()
    L.check(lib.sgd_igemm(C.byref(a), st), "igemm")
    sums = torch.zeros(n, cout + 8, 2, device="cuda")
    L.check(lib.sgd_stats_reduce(C.c_void_p(partial.data_ptr()), n, parts, cout, C.c_void_p(sums.data_ptr()), cout + 8, 8, st),
            "reduce")
    ref = torch.zeros(n, cout + 8, 2, device="cuda")
    L.check(lib.sgd_chan_stats(C.c_void_p(y.data_ptr()), n, h * w, cout, C.c_void_p(ref.data_ptr()), cout + 8, 8, st), "stats")
    torch.cuda.synchronize()
    yy = y.double().reshape(n, h * w, cout).cpu()
    exact = torch.stack([yy.sum(1), (yy * yy).sum(1)], -1)
    assert float((sums[:, 8:].cpu().double() - exact).abs().max() / exact.abs().max()) < 2e-6
    assert float((ref[:, 8:].cpu().double() - exact).abs().max() / exact.abs().max()) < 2e-6
    assert float(sums[:, :8].abs().max()) == 0.0


@pytest.mark.parametrize("n,c0,parts0,c1,parts1,film", [(3, 128, 32, 0, 0, True), (2, 256, 8, 128, 8, False), (4, 64, 128, 0, 0, False),
                                                         (2, 512, 2, 512, 0, True), (1, 1024, 5, 0, 0, False), (3, 96, 37, 32, 3, True),
                                                         (2, 128, 1, 0, 0, False),
                                                         (1, 4096, 3, 0, 0, False)])      # 96 KB of dynamic LDS (> the 64 KB default)
def test_groupnorm_coefficients_from_partial_statistics(n, c0, parts0, c1, parts1, film):
    """sgd_gn_coef_parts (partial statistics of up to two concatenated producers -> GroupNorm + FiLM coefficients in one
    launch; several threads per channel since round 4) against float64, and against sgd_stats_reduce + sgd_gn_coef.
    parts1 == 0 with c1 > 0: that source's sums are already in place (a sgd_chan_stats source)."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(61)
    c, hw, groups = c0 + c1, 4096, 32
    x = torch.randn(n, hw, c, generator=g, dtype=torch.float64) * 1.7 + 0.4          # the tensor the statistics describe

    def partials(xs, parts):          # [n, parts, 2, cs]: sums over `parts` row ranges of uneven length
        cuts = sorted(torch.randperm(hw - 1, generator=g)[:parts - 1].add(1).tolist()) if parts > 1 else []
        edges = [0] + cuts + [hw]
        out = torch.empty(n, parts, 2, xs.shape[-1], dtype=torch.float64)
        for k in range(parts):
            seg = xs[:, edges[k]:edges[k + 1]]
            out[:, k, 0], out[:, k, 1] = seg.sum(1), (seg * seg).sum(1)
        return out.float().cuda()

    p0 = partials(x[..., :c0], parts0)
    p1 = partials(x[..., c0:], parts1) if (c1 and parts1) else None
    gamma, beta = torch.randn(c, generator=g).cuda(), torch.randn(c, generator=g).cuda()
    fl = torch.randn(n, 2 * c, generator=g).cuda() if film else None
    sums = torch.full((n, c, 2), float("nan"), device="cuda")
    if c1 and not parts1:              # second source: sums already in place
        xs = x[..., c0:]
        sums[:, c0:, 0], sums[:, c0:, 1] = xs.sum(1).float().cuda(), (xs * xs).sum(1).float().cuda()
    sums_in = sums.clone()
    a, b = torch.empty(n, c, device="cuda"), torch.empty(n, c, device="cuda")
    L.check(lib.sgd_gn_coef_parts(_p(p0), parts0, c0, _p(p1), parts1 if p1 is not None else 0, c1, _p(sums), _p(gamma), _p(beta),
                                  _p(fl), 2 * c, n, groups, hw, 1e-5, _p(a), _p(b), _stream()), "coef_parts")
    torch.cuda.synchronize()
    # float64 reference of the affine (a, b): y = a * x + b == (GroupNorm(x) * (1 + scale) + shift)
    xg = x.reshape(n, hw, groups, c // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = xg.var(dim=(1, 3), unbiased=False, keepdim=True)
    rstd = (1.0 / torch.sqrt(var + 1e-5)).expand(n, 1, groups, c // groups).reshape(n, c)
    mu = mean.expand(n, 1, groups, c // groups).reshape(n, c)
    ga = gamma.cpu().double()[None] * rstd
    be = beta.cpu().double()[None] - mu * ga
    if film:
        sc, sh = 1 + fl.cpu().double()[:, :c], fl.cpu().double()[:, c:]
        ga, be = ga * sc, be * sc + sh
    assert max_rel(a.cpu().double(), ga) < 2e-5 and max_rel(b.cpu().double(), be) < 2e-5
    assert torch.isfinite(sums).all()
    # the two-launch route on the same partials
    s2 = sums_in.clone()
    L.check(lib.sgd_stats_reduce(_p(p0), n, parts0, c0, _p(s2), c, 0, _stream()), "reduce0")
    if p1 is not None:
        L.check(lib.sgd_stats_reduce(_p(p1), n, parts1, c1, _p(s2), c, c0, _stream()), "reduce1")
    a2, b2 = torch.empty(n, c, device="cuda"), torch.empty(n, c, device="cuda")
    L.check(lib.sgd_gn_coef(_p(s2), _p(gamma), _p(beta), _p(fl), 2 * c, n, c, groups, hw, 1e-5, _p(a2), _p(b2), _stream()), "coef")
    assert max_rel(sums, s2) < 1e-6 and max_rel(a, a2) < 1e-6 and max_rel(b, b2) < 1e-6


@pytest.mark.parametrize("m,n,k,ksplit", [(7, 100, 1234, 5), (160, 256, 5000, 39), (256, 64, 33, 1)])
def test_linear_splitk(m, n, k, ksplit):
    """skinny split-K linear (mlp_cond.0 at K = 5000) vs float64; ragged m / n / k and a padded x row stride"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(m, k + 3, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    work = torch.full((ksplit, m, n), float("nan"), device="cuda")
    y = torch.full((m, n + 2), float("nan"), device="cuda")
    L.check(lib.sgd_linear_splitk(_p(xd), k + 3, _p(wd), _p(bd), m, n, k, _p(work), ksplit, _p(y), n + 2, _stream()), "lin")
    ref = (x[:, :k].double() @ w.double().t() + b.double()).float()
    assert max_rel(y[:, :n].cpu(), ref) < 2e-6
    assert torch.isnan(y[:, n:]).all()
    assert lib.sgd_linear_splitk(_p(xd), k + 3, _p(wd), _p(bd), 257, n, k, _p(work), ksplit, _p(y), n + 2, _stream()) == 1


# --------------------------------------------------------------------------------------------------------------------
# canary for the quarter-wave zero product of the LayerNorm-row prologue (profiles/r4_ln_hazard.txt)
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f16x3", "bf16x3", "f32"])
@pytest.mark.parametrize("shape", [(160 * 256, 512, 512), (160 * 256, 512, 128), (16 * 256, 512, 128), (128, 512, 128)],
                         ids=["c5_to_q", "c5_to_kv", "b8_to_kv", "one_workgroup"])
def test_layernorm_prologue_canary_on_the_shipped_instance(shape, prec):
    """The LayerNorm-row prologue (Attention_LR's to_q / to_kv, crossattetion_lr.py:81-88) on the SHIPPED one-plane 1x1
    instance, at the shapes C5 launches (UNet batch 160, 16x16 maps, 512 channels; to_kv has 128 outputs) and at the
    one-workgroup size where the two-plane experiment failed on every launch.  The fault found in round 4 on the opt-in
    two-plane instance returned, in the last quarter-wave of a loader wave, the LayerNorm value with a zero product --
    EXACTLY beta -- and came and went with unrelated code motion.  With W = I the output IS the staged operand, so the
    signature is directly visible: over 200 launches no cell may equal beta exactly where the true value is not beta, and
    every cell is within the split's rounding of ((x - mean) * rstd) * gamma + beta.  Seconds on the GPU; it fails first
    if a future edit of csrc/igemm.hip arms the fault in the instance the product runs."""
    L, lib = _lib()
    m, cin, cout = shape
    p = L.PREC_BY_NAME[prec]
    g = torch.Generator().manual_seed(404)
    x = (torch.randn(m, cin, generator=g) * 1.3 + 0.2).cuda()
    gamma = (1.0 + 0.25 * torch.randn(cin, generator=g)).cuda()
    beta = (0.5 + 0.25 * torch.randn(cin, generator=g)).cuda()               # no zeros: exact-beta cells are unambiguous
    w = torch.zeros(cout, cin)
    w[torch.arange(cout), torch.arange(cout)] = 1.0                           # first `cout` rows of the identity
    wbuf, cin_p, cout_p = _pack(w.cuda(), 1, p)
    st = torch.empty(m, 2, device="cuda")
    L.check(lib.sgd_ln_stats(_p(x), m, cin, 1e-5, _p(st), _stream()), "ln_stats")
    y = torch.empty(m, cout, device="cuda")
    a = L.IgemmArgs()
    a.x0, a.c0 = x.data_ptr(), cin
    a.mode, a.m, a.stride = L.MODE_FLAT, m, 1
    a.pro, a.pa, a.pb, a.pc = L.PRO_LN_ROW, st.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    a.w, a.cin_p, a.cout_p = wbuf.data_ptr(), cin_p, cout_p
    a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout, cout, p
    work = torch.zeros(int(lib.sgd_igemm_work_bytes()) // 4, device="cuda")
    a.work, a.work_bytes = work.data_ptr(), work.numel() * 4
    torch.cuda.synchronize()
    want = ((x[:, :cout] - st[:, :1]) * st[:, 1:2]) * gamma[:cout] + beta[:cout]
    far = (want - beta[:cout]).abs() > 1e-3                                    # cells whose true value is visibly not beta
    tol = {"f32": 2e-6, "f16x3": 4e-6, "bf16x3": 6e-5}[prec] * float(want.abs().max())
    bad_beta = bad_val = 0
    for rep in range(200):
        y.fill_(float("nan"))
        L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm")
        bad_beta += int(((y == beta[:cout]) & far).sum())
        bad_val += int(((y - want).abs() > tol).sum())
    torch.cuda.synchronize()
    assert bad_beta == 0, f"{bad_beta} cells returned exactly beta (zero LayerNorm product) in 200 launches"
    assert bad_val == 0, f"{bad_val} cells off by more than {tol:.1e} in 200 launches"


@pytest.mark.parametrize("n,h,w_,cin,cout,pro,silu", [(3, 64, 64, 128, 3, True, 1), (2, 20, 40, 64, 3, True, 1), (1, 8, 8, 32, 4, False, 0),
                                                       (5, 16, 48, 256, 3, True, 0), (2, 33, 31, 96, 4, False, 1)])
def test_head_conv_narrow_out(n, h, w_, cin, cout, pro, silu):
    """sgd_conv3_narrow_out (the output head, openaimodel.py:830-835: GroupNorm + SiLU -> 3x3 conv with 3 / 4 output channels)
    against float64: tiles that end inside the map, no-prologue and SiLU-only forms, strided output rows"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(n * 100 + h)
    x = torch.randn(n, cin, h, w_, generator=g, dtype=torch.float64)
    wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    pa = 1 + 0.3 * torch.randn(n, cin, generator=g, dtype=torch.float64)
    pb = 0.3 * torch.randn(n, cin, generator=g, dtype=torch.float64)
    act = x * pa[:, :, None, None] + pb[:, :, None, None] if pro else x
    if silu:
        act = F.silu(act)
    ref = F.conv2d(act, wt, b, padding=1)
    xd = _nhwc(x.float()).cuda()
    w9 = wt.float().permute(2, 3, 0, 1).reshape(9, cout, cin).contiguous().cuda()
    pad, pbd, bd = pa.float().cuda(), pb.float().cuda(), b.float().cuda()
    y_ld = cout + 2
    y = torch.full((n, h, w_, y_ld), float("nan"), device="cuda")
    L.check(lib.sgd_conv3_narrow_out(_p(xd), _p(pad) if pro else None, _p(pbd) if pro else None, silu, _p(w9), _p(bd), _p(y), n, h, w_,
                                     cin, cout, y_ld, _stream()), "narrow_out")
    torch.cuda.synchronize()
    assert torch.isnan(y[..., cout:]).all()
    got = y[..., :cout].cpu().permute(0, 3, 1, 2).double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 6e-6     # (an fp32 FMA chain over 9 * cin terms)
    # argument validation: widths the kernel has no form for are refused
    assert lib.sgd_conv3_narrow_out(_p(xd), None, None, 0, _p(w9), None, _p(y), n, h, w_, cin, 5, y_ld, _stream()) == 1
    assert lib.sgd_conv3_narrow_out(_p(xd), None, None, 0, _p(w9), None, _p(y), n, h, w_, cin + 4, cout, y_ld, _stream()) == 1


def test_device_probes_run_and_report_plausible_rates():
    """csrc/tools/probe.hip (libsgdm_hip_tools.so): the in-run device calibration of bench.py and the stream probes behind DESIGN
    section 9 -- every entry point launches, fills its sink with finite values and lands in a plausible range (a broken probe
    would silently skew roofline.frac_of_device_ceiling)"""
    L, _ = _lib()
    lib = L.load_tools()
    st = _stream()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    sink = torch.full((4096 * 256,), float("nan"), device="cuda")

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3

    iters = 20000
    for variant in (0, 1, 2):
        t = timed(lambda: L.check(lib.sgd_debug_mfma_probe(cus, iters, 7, variant, _p(sink), st), "mfma probe"))
        tf = float(lib.sgd_debug_mfma_probe_flops(cus, iters, variant)) / t / 1e12
        assert 500.0 < tf < 2600.0, (variant, tf)              # 2.5 PF is the dense 16-bit peak of the chip
    assert torch.isfinite(sink[:cus * 256]).all()
    for rb, wps in ((0, 1), (8, 1), (4, 2)):
        t = timed(lambda: L.check(lib.sgd_debug_mfma_lds_probe(cus, 9000, 7, rb, wps, _p(sink), st), "lds probe"))
        tf = cus * 4.0 * wps * 9000 * 48 * 16384 / t / 1e12
        assert 500.0 < tf < 2600.0, (rb, wps, tf)
    wbuf = torch.randn(1 << 20, device="cuda").half()
    abuf = torch.randn(4 << 20, device="cuda")
    for ex in (0, 1, 3, 7, 11, 16 + 7):
        t = timed(lambda: L.check(lib.sgd_debug_mfma_stream_probe(cus, 9000, 7, ex, _p(wbuf), _p(abuf), 1 << 20, _p(sink), st), "stream probe"))
        tf = cus * (8.0 if ex & 16 else 4.0) * 9000 * 48 * 16384 / t / 1e12
        assert 300.0 < tf < 2600.0, (ex, tf)
    assert lib.sgd_debug_mfma_stream_probe(cus, 9000, 7, 4, _p(wbuf), _p(abuf), 1 << 20, _p(sink), st) != 0      # barrier without loaders
    src = torch.randn(1 << 24, device="cuda")
    dst = torch.empty_like(src)
    for variant in (0, 1, 2, 3, 4):
        dst.zero_()
        t = timed(lambda: L.check(lib.sgd_debug_copy_probe(_p(src), _p(dst), src.numel(), variant, 0, st), "copy probe"))
        assert torch.equal(src, dst) and 1.0 < 2 * src.numel() * 4 / t / 1e12 < 8.0, variant
    t = timed(lambda: L.check(lib.sgd_debug_copy_probe(_p(src), _p(dst), src.numel(), 5, 0, st), "read probe"))
    assert 1.0 < src.numel() * 4 / t / 1e12 < 8.0
    assert abs(float(dst[:2048 * 256 * 4].double().sum()) - float(src.double().sum())) < 1e-3 * src.numel() ** 0.5
    t = timed(lambda: L.check(lib.sgd_debug_copy_probe(_p(src), _p(dst), src.numel(), 6, 0, st), "write probe"))
    assert 1.0 < src.numel() * 4 / t / 1e12 < 8.0 and float(dst[3::4].min()) == 1.0
