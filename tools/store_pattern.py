#!/usr/bin/env python3
"""Micro-benchmark: how fast does one MI355X write (and read) a [rows x 128] fp32 matrix in the access patterns an MFMA
accumulator tile offers?  (compiled on the box with hipcc; tools only)

  P0  lane = pixel row, 16 bytes per lane, a 128-byte line filled by 4 store instructions   (accumulator [channel x pixel])
  P1  lane = channel, 4 bytes per lane: one store instruction = two full 128-byte lines     (accumulator [pixel x channel])
  P2  8 lanes x 16 bytes per row: one store instruction = eight full lines                   (after a register transpose)
"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
SRC = r'''
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int P, bool RD>
__global__ __launch_bounds__(256) void k(float* __restrict__ y, const float* __restrict__ r, int ntile, int cols) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        float* yb = y + (size_t)t * 128 * cols + w * 32;
        const float* rb = r + (size_t)t * 128 * cols + w * 32;
        for (int mt = 0; mt < 4; ++mt) {
            if (P == 0) {
                f4 v[4];
                for (int g = 0; g < 4; ++g) { v[g] = f4{1.f, 2.f, 3.f, (float)g}; if (RD) v[g] += *(const f4*)(rb + (size_t)(mt * 32 + li) * cols + 8 * g + 4 * lh); }
                for (int g = 0; g < 4; ++g) *(f4*)(yb + (size_t)(mt * 32 + li) * cols + 8 * g + 4 * lh) = v[g];
            } else if (P == 1) {
                float v[16];
                for (int i = 0; i < 16; ++i) { v[i] = (float)i; if (RD) v[i] += rb[(size_t)(mt * 32 + 8 * (i >> 2) + 4 * lh + (i & 3)) * cols + li]; }
                for (int i = 0; i < 16; ++i) yb[(size_t)(mt * 32 + 8 * (i >> 2) + 4 * lh + (i & 3)) * cols + li] = v[i];
            } else {
                f4 v[4];
                for (int i = 0; i < 4; ++i) { v[i] = f4{1.f, 2.f, 3.f, (float)i}; if (RD) v[i] += *(const f4*)(rb + (size_t)(mt * 32 + i * 8 + (lane >> 3)) * cols + 4 * (lane & 7)); }
                for (int i = 0; i < 4; ++i) *(f4*)(yb + (size_t)(mt * 32 + i * 8 + (lane >> 3)) * cols + 4 * (lane & 7)) = v[i];
            }
        }
    }
}
extern "C" void run(int p, int rd, float* y, const float* r, int ntile, int cols, int grid, void* st) {
    hipStream_t s = (hipStream_t)st;
#define L(P, R) hipLaunchKernelGGL((k<P, R>), dim3(grid), dim3(256), 0, s, y, r, ntile, cols)
    if (p == 0) { if (rd) L(0, true); else L(0, false); }
    else if (p == 1) { if (rd) L(1, true); else L(1, false); }
    else { if (rd) L(2, true); else L(2, false); }
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "k.hip"), "w").write(SRC)
so = os.path.join(d, "k.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(d, "k.hip"), "-o", so])
lib = C.CDLL(so)
rows, cols = 80 * 4096, 128
y = torch.empty(rows, cols, device="cuda"); r = torch.randn(rows, cols, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for grid in (256, 512, 2560):
    for p in (0, 1, 2):
        for rd in (0, 1):
            f = lambda: lib.run(p, rd, C.c_void_p(y.data_ptr()), C.c_void_p(r.data_ptr()), rows // 128, cols, grid, C.c_void_p(st))
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            gb = rows * cols * 4 * (1 + rd) / 1e9
            print(f"grid {grid:5d} P{p} {'read+write' if rd else 'write     '}: {ms*1e3:7.1f} us  {gb/ms*1e3/1e3:6.2f} TB/s", flush=True)
