#!/usr/bin/env python3
"""What HBM rate does THIS device deliver, and to which access shape?  Sweeps the copy probe of the diagnostics library
(csrc/tools/probe.hip: variant x grid x buffer size), interleaved in one process, next to torch's own copy / add / fill.
    python tools/hbm_probe_sweep.py [--rounds 5]"""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L
ap = argparse.ArgumentParser(); ap.add_argument("--rounds", type=int, default=5); a = ap.parse_args()
lib = L.load_tools()
st = torch.cuda.current_stream().cuda_stream
NAMES = {0: "stride4", 1: "stride8", 2: "stride4_nt", 3: "piece8k", 4: "piece8k_nt", 5: "read", 6: "write"}


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


for mib in (256, 1024, 4096):
    n = mib << 18
    src, dst = torch.randn(n, device="cuda"), torch.empty(n, device="cuda")
    reps = max(2, 2560 // mib)
    cases = [(v, b) for v in range(7) for b in (1024, 2048, 4096, 8192)]
    res = {c: [] for c in cases}
    tor = {"torch_copy": [], "torch_add": [], "torch_fill": []}
    for _ in range(a.rounds):
        for v, b in cases:
            t = timed(lambda: L.check(lib.sgd_debug_copy_probe(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, v, b, st), "probe"), reps)
            res[(v, b)].append((1 if v >= 5 else 2) * 4.0 * n / t / 1e12)
        tor["torch_copy"].append(2 * 4.0 * n / timed(lambda: dst.copy_(src), reps) / 1e12)
        tor["torch_add"].append(3 * 4.0 * n / timed(lambda: torch.add(src, dst, out=dst), reps) / 1e12)
        tor["torch_fill"].append(4.0 * n / timed(lambda: dst.fill_(1.0), reps) / 1e12)
    print(f"# {mib} MiB per buffer, TB/s of bytes read + written (median of {a.rounds}, max)")
    for v in range(7):
        print(f"{NAMES[v]:>11}: " + "  ".join(f"grid {b}: {statistics.median(res[(v, b)]):.2f} ({max(res[(v, b)]):.2f})" for b in (1024, 2048, 4096, 8192)))
    print("  ".join(f"{k}: {statistics.median(x):.2f} ({max(x):.2f})" for k, x in tor.items()))
    del src, dst
