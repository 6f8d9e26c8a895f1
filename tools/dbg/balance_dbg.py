import sys, os, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "self-guided-diffusion-models_amd")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from test_hip_fullsize import _pack, _lib, _nhwc, _p, _stream, _work
n, cin, cout, h, taps = [int(v) for v in sys.argv[1].split("x")]
flags = sys.argv[2] if len(sys.argv) > 2 else ""      # p: prologue, r: residual, s: stats
prec = "f32"
L, lib = _lib(); p = L.PREC_BY_NAME[prec]
g = torch.Generator().manual_seed(53)
x = torch.randn(n, cin, h, h, generator=g); ks = 3 if taps == 9 else 1
w = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * taps); b = torch.randn(cout, generator=g)
pa, pb = 1 + 0.3 * torch.randn(n, cin, generator=g), 0.3 * torch.randn(n, cin, generator=g)
res = torch.randn(n, cout, h, h, generator=g)
act = x.double()
if "p" in flags: act = F.silu(act * pa.double()[:, :, None, None] + pb.double()[:, :, None, None])
ref = F.conv2d(act, w.double(), b.double(), padding=ks // 2)
if "r" in flags: ref = ref + res.double()
wbuf, cp, op = _pack(w.cuda(), ks, p)
xd, bd, rd, pad, pbd = _nhwc(x).cuda(), b.cuda(), _nhwc(res).cuda(), pa.cuda(), pb.cuda()
work, nbytes = _work()
def run(ww):
    a = L.IgemmArgs(); a.x0, a.c0 = xd.data_ptr(), cin
    if taps == 9: a.mode, a.n, a.hi, a.wi, a.ho, a.wo, a.stride = L.MODE_CONV3, n, h, h, h, h, 1
    else: a.mode, a.m, a.rows_per_n, a.stride = L.MODE_FLAT, n * h * h, h * h, 1
    if "p" in flags: a.pro, a.pa, a.pb, a.pro_silu = L.PRO_AFFINE_NC, pad.data_ptr(), pbd.data_ptr(), 1
    if "r" in flags: a.res = rd.data_ptr()
    a.w, a.cin_p, a.cout_p, a.bias = wbuf.data_ptr(), cp, op, bd.data_ptr()
    y = torch.full((n, h, h, cout), float("nan"), device="cuda"); a.y, a.cout, a.y_ld, a.prec = y.data_ptr(), cout, cout, p
    if ww: a.work, a.work_bytes = work.data_ptr(), nbytes
    if "s" in flags:
        parts = lib.sgd_igemm_stats_parts(C.byref(a))
        partial = torch.full((n, parts, 2, cout), float("nan"), device="cuda"); a.stats = partial.data_ptr()
    L.check(lib.sgd_igemm(C.byref(a), _stream()), "igemm"); torch.cuda.synchronize()
    return y.cpu().reshape(-1, cout)
r = ref.permute(0, 2, 3, 1).reshape(-1, cout)
for ww in (False, True, True):
    print("work" if ww else "plain", flags, end=" ", flush=True)
    y = run(ww)
    err = (y.double() - r).abs().reshape(-1, 128, cout).amax((1, 2)) / r.abs().max()
    bad = [(i, round(float(e), 3)) for i, e in enumerate(err) if not e < 1e-5]
    print("bad tiles:", bad[:24], "of", len(err), "counters!=0:", int((work.view(torch.int32)[:512] != 0).sum()), flush=True)
# ---- which K part is missing / doubled in a bad tile?
if taps == 1:
    xs = act.permute(0, 2, 3, 1).reshape(-1, cin)
    wm = w.double().reshape(cout, cin)
    nch = cin // 32
    for rep in range(4):
        if rep == 2:
            work.zero_(); torch.cuda.synchronize(); print("(workspace zeroed)")
        y = run(True)
        err = (y.double() - r).abs().reshape(-1, 128, cout).amax((1, 2)) / r.abs().max()
        bad = [i for i, e in enumerate(err) if not e < 1e-5]
        msg = []
        for t in bad[:4]:
            rows = slice(t * 128, t * 128 + 128)
            d = y[rows].double() - r[rows]
            for part in range(4):
                c0, c1 = part * nch // 4 * 32, (part + 1) * nch // 4 * 32
                pp = xs[rows, c0:c1] @ wm[:, c0:c1].t()
                for sign, nm in ((1, "+"), (-1, "-")):
                    if float((d - sign * pp).abs().max()) < 1e-3 * float(r.abs().max()):
                        msg.append(f"tile {t}: {nm}part{part}")
        print("rep", rep, "bad", bad, msg, flush=True)
