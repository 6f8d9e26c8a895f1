#!/usr/bin/env python3
"""Per-wave cycle accounting of ONE conv launch with the diagnostic library (build.py --probe).
    python tools/probe_conv.py --cin 256 --cout 256 --hw 32 [--dbg N] [--prec f16x3]
Prints, per role (compute / input-tile loader / weight loader): cycles per K step and the share spent waiting
inside the per-step barrier.  The role with the smallest barrier share paces the block."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libsgdm_hip_probe.so")

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=80); ap.add_argument("--cin", type=int, default=128)
ap.add_argument("--cout", type=int, default=128); ap.add_argument("--hw", type=int, default=64)
ap.add_argument("--prec", default="f16x3"); ap.add_argument("--ks", type=int, default=3)
ap.add_argument("--trace", action="store_true"); ap.add_argument("--dbg", type=int, default=0); ap.add_argument("--plain", action="store_true")
a = ap.parse_args()
os.environ["SGDM_DBG"] = str(a.dbg)
lib = L.load(); prec = L.PREC_BY_NAME[a.prec]
dev = "cuda"
n, cin, cout, hw = a.n, a.cin, a.cout, a.hw
x = torch.randn(n, hw, hw, cin, device=dev)
w = torch.randn(cout, cin, a.ks, a.ks, device=dev) / (cin * a.ks * a.ks) ** 0.5
bias = torch.randn(cout, device=dev); res = torch.randn(n, hw, hw, cout, device=dev)
pa, pb = torch.randn(n, cin, device=dev), torch.randn(n, cin, device=dev)
y = torch.empty(n, hw, hw, cout, device=dev)
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
for _ in range(20): L.check(lib.sgd_igemm(C.byref(g), st), "igemm")
NW = 8                                     # waves per block (4 MFMA + 4 input loaders)
stamps = torch.zeros(8192 * NW * 4, dtype=torch.int64, device=dev)
os.environ["SGDM_STAMP_PTR"] = hex(stamps.data_ptr())
trace = torch.zeros(16 * NW * 512 * 2, dtype=torch.int64, device=dev)
if a.trace:
    os.environ["SGDM_TRACE_PTR"] = hex(trace.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
L.check(lib.sgd_igemm(C.byref(g), st), "igemm")
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
s = stamps.cpu().view(-1, NW, 4)
s = s[s[:, 0, 0] > 0]                      # blocks that ran
fl = 2.0 * n * hw * hw * cout * cin * a.ks * a.ks
print(f"cin={cin} cout={cout} hw={hw} ks={a.ks} prec={a.prec} dbg={a.dbg}: {ms:.4f} ms {fl/ms/1e9:.1f} TF  blocks={len(s)}")
for role, name, waves in ((0, "compute", slice(0, 4)), (1, "loader", slice(4, 8))):
    r = s[:, waves, :].double()
    tot, bar, nb = r[..., 0].mean(), r[..., 1].mean(), r[..., 2].mean()
    print(f"  {name:9s}: total {tot:9.0f} cyc  barriers {nb:6.0f}  per-barrier-interval {tot/nb:7.0f} cyc  in-barrier {bar/nb:7.0f} cyc ({100*bar/tot:4.1f}%)  work {(tot-bar)/nb:7.0f}  epilogue total {r[..., 3].mean():8.0f} ({100*r[..., 3].mean()/tot:4.1f}%)")

r = s.double()
print("  per-wave work/step:", " ".join(f"{((r[:, w, 0] - r[:, w, 1]) / r[:, w, 2]).mean():6.0f}" for w in range(NW)))
print("  per-wave in-barrier:", " ".join(f"{(r[:, w, 1] / r[:, w, 2]).mean():6.0f}" for w in range(NW)))

if a.trace:
    tr = trace.cpu().view(16, NW, 512, 2).double()
    nb = int(s[0, 0, 2])
    # work of wave w in interval i = arrival(i) - release(i-1); step phase = (i - 1) % 9 for the conv stream
    work = tr[:, :, 1:nb, 0] - tr[:, :, 0:nb - 1, 1]           # [blk, wave, nb-1]
    wait = tr[:, :, 1:nb, 1] - tr[:, :, 1:nb, 0]
    span = tr[:, :, 1:nb, 1] - tr[:, :, 0:nb - 1, 1]
    ph = torch.arange(nb - 1) % 9
    print("  phase (tap):        " + " ".join(f"{p:6d}" for p in range(9)))
    for w in range(NW):
        print(f"  wave {w:2d} work      : " + " ".join(f"{work[:, w, ph == p].mean():6.0f}" for p in range(9)))
    print("  step span (wave 0): " + " ".join(f"{span[:, 0, ph == p].mean():6.0f}" for p in range(9)))
    print("  min wait over waves:" + " ".join(f"{wait[:, :, ph == p].min(dim=1).values.mean():6.0f}" for p in range(9)))
