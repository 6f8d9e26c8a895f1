#!/usr/bin/env python3
"""Micro-benchmark of ONE fused conv launch (the dominant kernel) through the C-ABI.
    python tools/bench_conv.py --n 80 --cin 128 --cout 128 --hw 64 --prec f16x3 [--reps 20] [--plain]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=80); ap.add_argument("--cin", type=int, default=128)
ap.add_argument("--cout", type=int, default=128); ap.add_argument("--hw", type=int, default=64)
ap.add_argument("--prec", default="f16x3"); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--plain", action="store_true", help="no GN/SiLU prologue, no residual")
ap.add_argument("--ks", type=int, default=3)
ap.add_argument("--dbg", type=int, default=-1, help="use the probe library with this SGDM_DBG ablation mask (no stamps)")
ap.add_argument("--zeros", action="store_true", help="all-zero operands (DVFS diagnostic: clock under load vs data)")
a = ap.parse_args()
if a.dbg >= 0:
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libsgdm_hip_probe.so"); os.environ["SGDM_DBG"] = str(a.dbg)
lib = L.load(); prec = L.PREC_BY_NAME[a.prec]
dev = "cuda"
n, cin, cout, hw = a.n, a.cin, a.cout, a.hw
x = torch.randn(n, hw, hw, cin, device=dev)
w = torch.randn(cout, cin, a.ks, a.ks, device=dev) / (cin * a.ks * a.ks) ** 0.5
bias = torch.randn(cout, device=dev); res = torch.randn(n, hw, hw, cout, device=dev)
pa, pb = torch.randn(n, cin, device=dev), torch.randn(n, cin, device=dev)
y = torch.empty(n, hw, hw, cout, device=dev)
if a.zeros:
    for t_ in (x, w, bias, res, pa, pb): t_.zero_()
buf = torch.empty(lib.sgd_packed_weight_bytes(cout, cin, a.ks, prec) // 4, device=dev)
cp, op = C.c_int32(), C.c_int32()
st = torch.cuda.current_stream().cuda_stream
L.check(lib.sgd_pack_weight(C.c_void_p(w.data_ptr()), C.c_void_p(buf.data_ptr()), cout, cin, a.ks, prec, C.byref(cp), C.byref(op), st), "pack")
g = L.IgemmArgs()
g.x0, g.c0 = x.data_ptr(), cin
if a.ks == 3:
    g.mode, g.n, g.hi, g.wi, g.ho, g.wo, g.stride = L.MODE_CONV3, n, hw, hw, hw, hw, 1
else:
    g.mode, g.m, g.rows_per_n, g.stride = L.MODE_FLAT, n * hw * hw, hw * hw, 1
if not a.plain:
    g.pro, g.pro_silu, g.pa, g.pb = L.PRO_AFFINE_NC, 1, pa.data_ptr(), pb.data_ptr()
    g.res = res.data_ptr()
g.w, g.cin_p, g.cout_p, g.bias = buf.data_ptr(), cp.value, op.value, bias.data_ptr()
g.y, g.cout, g.y_ld, g.prec = y.data_ptr(), cout, cout, prec
for _ in range(3): L.check(lib.sgd_igemm(C.byref(g), st), "igemm")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(a.reps): lib.sgd_igemm(C.byref(g), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
fl = 2.0 * n * hw * hw * cout * cin * a.ks * a.ks
print(f"n={n} cin={cin} cout={cout} hw={hw} ks={a.ks} prec={a.prec} plain={a.plain} zeros={a.zeros} dbg={a.dbg}: {ms:.4f} ms  {fl/ms/1e9:.1f} TFLOP/s")
