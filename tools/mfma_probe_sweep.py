#!/usr/bin/env python3
"""What does the bare-MFMA probe (csrc/probe.hip) measure?  Sweeps variant (16x16x32 random / zeros, 32x32x16 random),
launch duration and the number of active compute units, interleaved in one process.
    python tools/mfma_probe_sweep.py [--lds [--quarter]]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-guided-diffusion-models_amd"))
import torch
from sgdm_amd import _lib as L
lib = L.load_tools()
st = torch.cuda.current_stream().cuda_stream
cus = torch.cuda.get_device_properties(0).multi_processor_count
sink = torch.empty(4096 * 256, device="cuda")


def run(blocks, iters, variant):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.sgd_debug_mfma_probe(blocks, iters, 777, variant, C.c_void_p(sink.data_ptr()), st), "probe")
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    return float(lib.sgd_debug_mfma_probe_flops(blocks, iters, variant)) / t / 1e12, t


def run_lds(blocks, iters, row_blocks, wps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.sgd_debug_mfma_lds_probe(blocks, iters, 777, row_blocks, wps, C.c_void_p(sink.data_ptr()), st), "lds probe")
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    return blocks * 4.0 * wps * iters * 48 * 16384 / t / 1e12, t


if "--quarter" in sys.argv:      # with --lds: a quarter of the CUs (no power limit: what each structure reaches at the full clock)
    cus = cus // 4
if "--lds" in sys.argv:
    # the conv kernel's compute-wave stream alone: 48 MFMAs per step, 16 / 8 / 0 LDS fragment reads per step, 1 or 2 waves per SIMD
    for rb, wps in ((0, 1), (8, 1), (4, 1), (0, 2), (8, 2), (4, 2)):
        run_lds(cus, 2000, rb, wps)
    for rnd in range(3):
        for rb in (0, 8, 4):
            for wps in (1, 2):
                tf, t = run_lds(cus, 40000 // wps, rb, wps)
                print(f"round {rnd} row blocks {rb} ({2 * rb:2d} LDS reads per 48 MFMAs) waves/SIMD {wps}: {t * 1e3:8.2f} ms  {tf:7.1f} TF raw "
                      f"= {tf / 3:6.1f} TF per fp32 product", flush=True)
    # ... and the rest of the conv kernel's per-step traffic added to the shipped tile's stream piece by piece
    wbuf = torch.randn(1 << 19, device="cuda").half()                       # 1 MiB of weight fragments (stays in L2)
    arows = 1 << 24
    abuf = torch.randn(arows * 4, device="cuda")                           # 256 MiB of 16-byte input rows (streamed)
    label = {0: "stream only (16 LDS reads)", 1: "+ weight fragments from L2", 2: "+ loader waves", 3: "+ weights + loaders",
             6: "+ loaders + barrier per chunk", 7: "+ weights + loaders + barrier", 10: "+ loaders + LDS counters",
             11: "+ weights + loaders + LDS counters"}

    wbuf = torch.randn(1 << 20, device="cuda").half()                       # (2 MiB: the 8-wave variant's steps are twice as wide)

    def run_ex(ex, iters=36000):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.sgd_debug_mfma_stream_probe(cus, iters, 777, ex, C.c_void_p(wbuf.data_ptr()), C.c_void_p(abuf.data_ptr()), arows,
                                                C.c_void_p(sink.data_ptr()), st), "stream probe")
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
        return cus * (8.0 if ex & 16 else 4.0) * (iters // 9 * 9) * 48 * 16384 / t / 1e12, t
    for ex in (0, 1, 3, 7):
        label[16 + ex] = "8 waves of 64x64: " + label[ex].replace("stream only (16 LDS reads)", "stream only (8 reads)")
    for ex in label:
        run_ex(ex, 1800)
    for rnd in range(3):
        for ex in label:
            tf, t = run_ex(ex, 18000 if ex & 16 else 36000)
            print(f"round {rnd} {label[ex]:34s}: {t * 1e3:8.2f} ms  {tf:7.1f} TF raw = {tf / 3:6.1f} TF per fp32 product", flush=True)
    sys.exit(0)

names = {0: "16x16x32 random", 1: "16x16x32 zeros", 2: "32x32x16 random"}
for v in (0, 1, 2):
    run(cus, 2000, v)
for rnd in range(2):
    for blocks in (cus, cus // 2, cus // 4, 2 * cus):
        for v in (0, 1, 2):
            for iters in (2000, 20000, 200000, 2000000):
                it = iters if v != 2 else iters // 2
                tf, t = run(blocks, it, v)
                print(f"round {rnd} blocks {blocks:4d} {names[v]:16s} iters {it:8d}: {t * 1e3:9.2f} ms  {tf:7.1f} TF  "
                      f"(implied clock at 16 cyc / 16x16x32: {tf * 1e12 / (blocks * 4 * 1024.0) / 1e9:.2f} GHz)", flush=True)
