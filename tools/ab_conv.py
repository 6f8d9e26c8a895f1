#!/usr/bin/env python3
"""A/B of the fused conv kernel across builds / launch modes IN ONE PROCESS (cdna_hip_programming.md rule 24): every
variant is a (library path, per-call overrides) pair -- `tune=<SGD_TUNE_* bits>` / `grid_cap=<n>` set the fields of
sgd_igemm_args (the library reads no environment), anything else is put into the environment for libraries older than
ABI 15; rounds are interleaved, median and min reported, outputs of all variants compared with the first.

    python tools/ab_conv.py --variants base=lib/libsgdm_hip_base.so new=lib/libsgdm_hip.so new256=lib/libsgdm_hip.so:tune=2 \
        --shapes 80,256,256,64 80,512,512,32 ... [--rounds 7] [--reps 20] [--prec f16x3] [--plain]
shape = n,cin,cout,hw[,ks]"""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--variants", nargs="+", required=True)
ap.add_argument("--shapes", nargs="+", required=True)
ap.add_argument("--rounds", type=int, default=7); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--prec", default="f16x3"); ap.add_argument("--plain", action="store_true")
a = ap.parse_args()
LIBDIR = os.path.dirname(L.LIB_PATH)


def load(path):
    lib = C.CDLL(path if os.path.isabs(path) else os.path.join(os.path.dirname(LIBDIR), path))
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    return lib


variants = []
for v in a.variants:
    name, rest = v.split("=", 1)
    path, *envs = rest.split(":")
    variants.append((name, load(path), dict(e.split("=") for e in envs)))
prec = L.PREC_BY_NAME[a.prec]
st = torch.cuda.current_stream().cuda_stream
WORK = {}
for shp in a.shapes:
    f = [int(x) for x in shp.split(",")]
    n, cin, cout, hw = f[:4]; ks = f[4] if len(f) > 4 else 3
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(n, hw, hw, cin, device="cuda", generator=g)
    w = torch.randn(cout, cin, ks, ks, device="cuda", generator=g) / (cin * ks * ks) ** 0.5
    bias = torch.randn(cout, device="cuda", generator=g); res = torch.randn(n, hw, hw, cout, device="cuda", generator=g)
    pa, pb = 1 + 0.3 * torch.randn(n, cin, device="cuda", generator=g), 0.3 * torch.randn(n, cin, device="cuda", generator=g)
    runs = []
    for name, lib, env in variants:
        buf = torch.empty(lib.sgd_packed_weight_bytes(cout, cin, ks, prec) // 4, device="cuda")
        cp, op = C.c_int32(), C.c_int32()
        L.check(lib.sgd_pack_weight(C.c_void_p(w.data_ptr()), C.c_void_p(buf.data_ptr()), cout, cin, ks, prec, C.byref(cp), C.byref(op), st), "pack")
        y = torch.full((n, hw, hw, cout), float("nan"), device="cuda")
        q = L.IgemmArgs()
        q.x0, q.c0 = x.data_ptr(), cin
        if ks == 3:
            q.mode, q.n, q.hi, q.wi, q.ho, q.wo, q.stride = L.MODE_CONV3, n, hw, hw, hw, hw, 1
        else:
            q.mode, q.m, q.rows_per_n, q.stride = L.MODE_FLAT, n * hw * hw, hw * hw, 1
        if not a.plain:
            q.pro, q.pro_silu, q.pa, q.pb = L.PRO_AFFINE_NC, 1, pa.data_ptr(), pb.data_ptr()
            q.res = res.data_ptr()
        q.w, q.cin_p, q.cout_p, q.bias = buf.data_ptr(), cp.value, op.value, bias.data_ptr()
        q.y, q.cout, q.y_ld, q.prec = y.data_ptr(), cout, cout, prec
        q.tune, q.grid_cap = int(env.get("tune", 0)), int(env.get("grid_cap", 0))
        env = {k: v for k, v in env.items() if k not in ("tune", "grid_cap")}     # (a copy: the variant serves every shape)
        if hasattr(lib, "sgd_igemm_work_bytes"):            # balanced tail (tune=16 turns it off)
            wb = int(lib.sgd_igemm_work_bytes())
            if name not in WORK:
                WORK[name] = torch.zeros(wb // 4, device="cuda")
            q.work, q.work_bytes = WORK[name].data_ptr(), wb
        runs.append((name, lib, env, q, y, buf, []))

    def launch(lib, env, q, reps):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        for _ in range(reps):
            rc = lib.sgd_igemm(C.byref(q), st)
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        return rc

    for name, lib, env, q, y, buf, ts in runs:
        L.check(launch(lib, env, q, 3), name)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for name, lib, env, q, y, buf, ts in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch(lib, env, q, a.reps); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / a.reps)
    fl = 2.0 * n * hw * hw * cout * cin * ks * ks
    ref = runs[0][4]
    line = f"n={n} cin={cin} cout={cout} hw={hw} ks={ks}:"
    for name, lib, env, q, y, buf, ts in runs:
        med, mn = statistics.median(ts), min(ts)
        diff = float((y - ref).abs().max() / ref.abs().max())
        line += f"  {name} {med:.4f} ms ({fl / med / 1e9:.0f} TF, min {mn:.4f}, d={diff:.1e})"
    print(line, flush=True)
