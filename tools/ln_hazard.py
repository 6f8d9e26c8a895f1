#!/usr/bin/env python3
"""LayerNorm-row prologue (Attention_LR's to_q / to_kv) on the 1x1 / linear kernel: how often does each instance return a wrong
row?  (rounds 4-6: the two-plane instance compiled with packed-f32 code generation returns exactly beta in
whole quarter-waves of the staged operand, on every launch; since round 6 every LayerNorm launch of a split mode runs on
instances compiled without packed-f32 instructions.)  Four routes of the SAME library, `reps` launches each, every output row
checked against float64:
    nopk1   tune = 0                        one plane per barrier, no packed f32      <- what the product launches
    nopk2   tune = FLAT2                    two planes per barrier, no packed f32
    pk1     tune = LN_PACKED                one plane, packed f32 (the product's instance up to round 5)
    pk2     tune = FLAT2 | LN_PACKED        two planes, packed f32                    <- the failing combination of round 4
    python tools/ln_hazard.py [--prec f16x3] [--reps 30] [--shapes 40960,512,512 128,512,128 ...]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="+", default=["40960,512,512", "81920,256,256", "128,512,128", "20480,1024,1024"])
ap.add_argument("--prec", default="f16x3"); ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--nobeta", action="store_true")
a = ap.parse_args()
lib = L.load(); prec = L.PREC_BY_NAME[a.prec]
st = torch.cuda.current_stream().cuda_stream
ROUTES = [("nopk1", 0), ("nopk2", L.TUNE_FLAT2), ("pk1", L.TUNE_LN_PACKED), ("pk2", L.TUNE_FLAT2 | L.TUNE_LN_PACKED)]
for shp in a.shapes:
    m, cin, cout = (int(v) for v in shp.split(","))
    torch.manual_seed(0)
    x = torch.randn(m, cin, device="cuda")
    w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
    gamma, beta = torch.randn(cin, device="cuda"), torch.randn(cin, device="cuda")
    stats = torch.empty(m, 2, device="cuda")
    L.check(lib.sgd_ln_stats(C.c_void_p(x.data_ptr()), m, cin, C.c_float(1e-5), C.c_void_p(stats.data_ptr()), st), "ln_stats")
    buf = torch.empty(lib.sgd_packed_weight_bytes(cout, cin, 1, prec) // 4, device="cuda")
    cp, op = C.c_int32(), C.c_int32()
    L.check(lib.sgd_pack_weight(C.c_void_p(w.data_ptr()), C.c_void_p(buf.data_ptr()), cout, cin, 1, prec, C.byref(cp), C.byref(op), st), "pack")
    xd = x.double()
    xn = (xd - xd.mean(1, keepdim=True)) / (xd.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * gamma.double()
    if not a.nobeta:
        xn = xn + beta.double()
    ref = xn @ w.double().reshape(cout, cin).t()
    scale = float(ref.abs().max())
    y = torch.empty(m, cout, device="cuda")
    g = L.IgemmArgs()
    g.x0, g.c0 = x.data_ptr(), cin
    g.mode, g.m, g.rows_per_n, g.stride = L.MODE_FLAT, m, m, 1
    g.pro, g.pa, g.pb = L.PRO_LN_ROW, stats.data_ptr(), gamma.data_ptr()
    if not a.nobeta:
        g.pc = beta.data_ptr()
    g.w, g.cin_p, g.cout_p = buf.data_ptr(), cp.value, op.value
    g.y, g.cout, g.y_ld, g.prec = y.data_ptr(), cout, cout, prec
    line = f"m={m} {cin}->{cout} {a.prec}:"
    for name, tune in ROUTES:
        g.tune = tune
        bad_launches, bad_rows, worst, mod8 = 0, 0, 0.0, {}
        for _ in range(a.reps):
            y.fill_(float("nan"))
            L.check(lib.sgd_igemm(C.byref(g), st), "igemm")
            err = ((y.double() - ref).abs().amax(1) / scale)
            err = torch.nan_to_num(err, nan=1.0)
            rows = (err > 2e-4).nonzero().flatten()
            worst = max(worst, float(err.max()))
            if rows.numel():
                bad_launches += 1
                bad_rows += int(rows.numel())
                for r in (rows % 8).tolist():
                    mod8[r] = mod8.get(r, 0) + 1
        line += f"  {name}: {bad_launches}/{a.reps} launches wrong, {bad_rows} rows, worst {worst:.1e}" + (f", rows mod 8 {dict(sorted(mod8.items()))}" if mod8 else "")
    print(line, flush=True)
