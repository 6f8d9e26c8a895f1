// 3x3 convolution of a 3- or 4-channel map (the UNet stem, openaimodel.py:560-566 / openaimodel_ca.py:735-741:
// conv_nd(dims, in_channels, model_channels, 3, padding=1)) as a plain fp32 FMA kernel.
//
// On the implicit-GEMM kernel this layer pads its 3 input channels to a 32-channel K step per tap: 9 K steps of which
// 29/32 are zeros, 0.112 ms at 20 TF/s for a layer whose real work is writing 168 MB (UNet batch 80: 0.03 ms at the
// HBM rate) with 27 multiply-adds per output.  Here a thread owns FOUR output channels and keeps their 27 x 4 weights in
// registers; a block stages the input rows it needs (halo included, zero padding resolved once) in LDS and walks its
// pixels: per pixel 9 broadcast 16-byte LDS reads, 108 FMAs, one 16-byte store -- 512 contiguous bytes per pixel across
// the 32 threads that share it.  The GroupNorm statistics of the output (sum y, sum y^2 per image and channel) leave as
// one partial pair per block, in the layout sgd_igemm's epilogue writes (include/sgdm_hip.h: sgd_igemm_args.stats).
// Exact fp32 whatever the engine's arithmetic mode.
#include <hip/hip_runtime.h>

#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"

namespace {

__device__ __forceinline__ f32x4 ld4n(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int CIN>
__global__ __launch_bounds__(256) void conv3_narrow_in_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              float* __restrict__ stats, int h, int w, int cout, int y_ld,
                                                              int rows_per_block, int parts, int adjoint) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [(R + 2) * (w + 2)][4] input, then [PL][Q][8] sums
    const int Q = cout >> 2, PL = blockDim.x / Q;
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q;
    const int n = blockIdx.x / parts, part = blockIdx.x % parts;
    const int r0 = part * rows_per_block, r1 = min(h, r0 + rows_per_block);
    const int wp = w + 2, hrows = r1 - r0 + 2;
    // ---- input rows r0 - 1 .. r1 (zero outside the map), channel-padded to 4
    for (int i = threadIdx.x; i < hrows * wp; i += blockDim.x) {
        const int yy = r0 - 1 + i / wp, xx = i % wp - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            const float* p = x + (((long)n * h + yy) * w + xx) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; ++c) v[c] = p[c];
        }
        *reinterpret_cast<f32x4*>(lds + (size_t)i * 4) = v;
    }
    // ---- this thread's weights: W[co][ci][tap], co = 4 q .. 4 q + 3; adjoint (input gradient of a conv with 3 / 4
    // OUTPUT channels, i.e. the head): the parameter is W[ci][co][tap] and the taps are flipped
    typedef float f32x2 __attribute__((ext_vector_type(2)));          // pairs: v_pk_fma_f32, two FMAs per issue slot
    f32x2 wr[9][CIN][2];
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    const bool live = pl < PL && q < Q;                               // (blockDim is Q * PL: always true, kept for clarity)
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    wr[t][c][j >> 1][j & 1] = adjoint ? wt[((long)c * cout + q * 4 + j) * 9 + (8 - t)]
                                                      : wt[((long)(q * 4 + j) * CIN + c) * 9 + t];
        if (bias) bq = ld4n(bias + q * 4);
    }
    __syncthreads();
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    const int npix = (r1 - r0) * w;
    // (row, column) of the thread's pixel advance by PL pixels per step: no division per pixel
    int oy = pl / w, ox = pl - oy * w;
    const int dy = PL / w, dx = PL - dy * w;
    for (int p = pl; p < npix; p += PL) {
        f32x2 a0 = {bq[0], bq[1]}, a1 = {bq[2], bq[3]};
        const float* tile = lds + (size_t)(oy * wp + ox) * 4;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 in = *reinterpret_cast<const f32x4*>(tile + (size_t)((t / 3) * wp + t % 3) * 4);
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const f32x2 xi = {in[c], in[c]};
                a0 = __builtin_elementwise_fma(xi, wr[t][c][0], a0);
                a1 = __builtin_elementwise_fma(xi, wr[t][c][1], a1);
            }
        }
        const f32x4 acc = {a0[0], a0[1], a1[0], a1[1]};
        *reinterpret_cast<f32x4*>(y + (((long)n * h + r0 + oy) * w + ox) * y_ld + q * 4) = acc;
        s1 += acc;
        s2 += acc * acc;
        oy += dy; ox += dx;
        if (ox >= w) { ox -= w; ++oy; }
    }
    if (!stats) return;
    // ---- GroupNorm partial sums of the block: fold the PL pixel lanes of every channel quad in lane order
    __syncthreads();                                                  // the input tile is dead
    float* red = lds;                                                 // [PL][Q][8]
    *reinterpret_cast<f32x4*>(red + ((size_t)pl * Q + q) * 8) = s1;
    *reinterpret_cast<f32x4*>(red + ((size_t)pl * Q + q) * 8 + 4) = s2;
    __syncthreads();
    if (pl == 0) {
        f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1;
        for (int k = 0; k < PL; ++k) {
            t1 += *reinterpret_cast<const f32x4*>(red + ((size_t)k * Q + q) * 8);
            t2 += *reinterpret_cast<const f32x4*>(red + ((size_t)k * Q + q) * 8 + 4);
        }
        float* sp = stats + ((long)n * parts + part) * 2 * cout + q * 4;
        *reinterpret_cast<f32x4*>(sp) = t1;
        *reinterpret_cast<f32x4*>(sp + cout) = t2;
    }
}

// output rows per block: about 512 pixels, whole rows (the 108 weight loads and the input staging of a block are its fixed
// cost: 256 pixels 122 us, 512 pixels 107 us, 1024 pixels 112 us at UNet batch 160, tools/bench_narrow.py)
inline int narrow_rows_per_block(int h, int w) {
    int r = 512 / (w > 0 ? w : 1);
    if (r < 1) r = 1;
    return r > h ? h : r;
}

}  // namespace

extern "C" int sgd_conv3_narrow_in_parts(int32_t h, int32_t w) {
    if (h <= 0 || w <= 0) return 0;
    const int r = narrow_rows_per_block(h, w);
    return (h + r - 1) / r;
}

extern "C" int sgd_conv3_narrow_in(const float* x, const float* w, const float* bias, float* y, float* stats, int32_t n,
                                   int32_t h, int32_t wd, int32_t cin, int32_t cout, int32_t y_ld, int32_t adjoint, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !w || !y || n <= 0 || h <= 0 || wd <= 0 || (cin != 3 && cin != 4) || cout <= 0 || (cout & 3) || cout > 1024 ||
        y_ld < cout || (y_ld & 3))
        return SGD_ERR_ARG;
    const int Q = cout / 4;
    const int PL = 256 / Q > 0 ? 256 / Q : 1;
    const int R = narrow_rows_per_block(h, wd), parts = (h + R - 1) / R;
    const size_t in_bytes = (size_t)(R + 2) * (wd + 2) * 16, red_bytes = (size_t)PL * Q * 32;
    const size_t smem = in_bytes > red_bytes ? in_bytes : red_bytes;
    if (smem > 64 * 1024) return SGD_ERR_ARG;
    const dim3 grid((unsigned)((long)n * parts)), block((unsigned)(Q * PL));
    if (cin == 3)
        hipLaunchKernelGGL((conv3_narrow_in_kernel<3>), grid, block, smem, (hipStream_t)stream, x, w, bias, y, stats, h, wd, cout,
                           y_ld, R, parts, adjoint);
    else
        hipLaunchKernelGGL((conv3_narrow_in_kernel<4>), grid, block, smem, (hipStream_t)stream, x, w, bias, y, stats, h, wd, cout,
                           y_ld, R, parts, adjoint);
    return sgd_check_launch();
}

// =============================================================================================================================
// The output head (openaimodel.py:830-835: normalization, SiLU, zero_module(conv_nd(dims, model_channels, out_channels, 3,
// padding=1))): a 3x3 conv with 3 (or 4) OUTPUT channels behind a GroupNorm + SiLU.  On the implicit-GEMM kernel its 3 columns
// ride a 32-column tile (0.121 ms at 18.6 TF at UNet batch 80, round 4); the layer's real work is one read of the 168 MB input
// and 3,456 fp32 FMAs per pixel.  The stem kernel's pattern, transposed: a thread owns FOUR INPUT channels (quad q of every
// 32-channel chunk) and keeps their 9 x COUT x 4 weights of the chunk in registers; a block (8 x 32 output pixels) stages the
// activated chunk -- GroupNorm affine + SiLU once per element, zero padding resolved -- as a 10 x 34 halo tile in LDS (pixel
// pitch 36 floats: conflict-free 16-byte reads) and every thread walks 8 pixels: per pixel 9 LDS reads of its own quad and
// 36 * COUT FMAs into that pixel's COUT partial sums, which stay in registers across the chunks and are folded over the eight
// quad lanes (three shuffle steps) once at the end.  Exact fp32 in every arithmetic mode.
// (A first version with one pixel per thread and the weights as scalar operands was bound by its scalar-load waits: 0.107 ms.)
// =============================================================================================================================
namespace {

constexpr int NO_TX = 32, NO_TY = 8, NO_HW = NO_TX + 2, NO_HH = NO_TY + 2, NO_PIX = NO_HW * NO_HH, NO_PITCH = 36;
constexpr int NO_ITEMS = (NO_PIX * 8 + 255) / 256;        // 16-byte items per thread per 32-channel chunk (11)

template <int COUT>
__global__ __launch_bounds__(256) void conv3_narrow_out_kernel(const float* __restrict__ x, const float* __restrict__ pa,
                                                               const float* __restrict__ pb, int silu,
                                                               const float* __restrict__ w9, const float* __restrict__ bias,
                                                               float* __restrict__ y, int h, int w, int cin, int y_ld,
                                                               int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float tile[NO_PIX * NO_PITCH];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int img = b / tiles_y;
    const int y0 = ty * NO_TY, x0 = tx * NO_TX;
    const int q = tid & 7, px = tid >> 3;                  // channel quad, output column; the thread's pixels are rows 0 .. 7
    // ---- this thread's staging items: halo pixel (hy, hx), channel quad q (the same for all of them)
    long goff[NO_ITEMS];                                   // element offset of the source pixel's row (-1: padding / past the tile)
    int loff[NO_ITEMS];
#pragma unroll
    for (int k = 0; k < NO_ITEMS; ++k) {
        const int pix = (tid + k * 256) >> 3;
        const int hy = pix / NO_HW, hx = pix - hy * NO_HW;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool in = pix < NO_PIX && yy >= 0 && yy < h && xx >= 0 && xx < w;
        goff[k] = in ? (((long)img * h + yy) * w + xx) * cin + q * 4 : -1;
        loff[k] = pix < NO_PIX ? pix * NO_PITCH + q * 4 : -1;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));          // (even, odd) channel pairs: v_pk_fma_f32 on register pairs
    f32x2 acc[NO_TY][COUT];
#pragma unroll
    for (int i = 0; i < NO_TY; ++i)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[i][co] = f32x2{0.f, 0.f};
    const float* const ap = pa ? pa + (long)img * cin + q * 4 : nullptr;
    const float* const bp = pa ? pb + (long)img * cin + q * 4 : nullptr;
    for (int ch = 0; ch < cin; ch += 32) {
        // ---- requests first (clamped addresses): the chunk's raw rows, its coefficients, this thread's weights
        f32x4 raw[NO_ITEMS];
#pragma unroll
        for (int k = 0; k < NO_ITEMS; ++k) raw[k] = ld4n(x + (goff[k] >= 0 ? goff[k] : (long)q * 4) + ch);
        f32x4 ka = {1.f, 1.f, 1.f, 1.f}, kb = {0.f, 0.f, 0.f, 0.f};
        if (ap) { ka = ld4n(ap + ch); kb = ld4n(bp + ch); }
        f32x4 wr[9][COUT];                                 // w9[tap][co][ch + 4 q .. + 3]
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int co = 0; co < COUT; ++co) wr[t][co] = ld4n(w9 + ((long)t * COUT + co) * cin + ch + q * 4);
        __syncthreads();                                   // every wave is past the previous chunk's pixels
#pragma unroll
        for (int k = 0; k < NO_ITEMS; ++k) {
            f32x4 v = raw[k] * ka + kb;
            if (silu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sgd_silu(v[e]);
            }
            if (goff[k] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (loff[k] >= 0) *reinterpret_cast<f32x4*>(tile + loff[k]) = v;
        }
        __syncthreads();
        // ---- 8 pixels (column px, rows 0 .. 7) x 9 taps: this thread's quad of each halo pixel against its weights
#pragma unroll
        for (int i = 0; i < NO_TY; ++i) {
            const float* row = tile + (i * NO_HW + px) * NO_PITCH + q * 4;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(row + ((t / 3) * NO_HW + t % 3) * NO_PITCH);
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    acc[i][co] = __builtin_elementwise_fma(f32x2{wr[t][co][0], wr[t][co][1]}, f32x2{xv[0], xv[1]}, acc[i][co]);
                    acc[i][co] = __builtin_elementwise_fma(f32x2{wr[t][co][2], wr[t][co][3]}, f32x2{xv[2], xv[3]}, acc[i][co]);
                }
            }
        }
    }
    // ---- fold the eight quad lanes of a pixel (lanes tid ^ 1, ^ 2, ^ 4), lane q == 0 stores
#pragma unroll
    for (int i = 0; i < NO_TY; ++i)
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float a = acc[i][co][0] + acc[i][co][1];
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            a += __shfl_xor(a, 4, 64);
            acc[i][co][0] = a;
        }
    const int ox = x0 + px;
    if (q == 0 && ox < w) {
#pragma unroll
        for (int i = 0; i < NO_TY; ++i) {
            const int oy = y0 + i;
            if (oy >= h) break;
            float* yp = y + (((long)img * h + oy) * w + ox) * y_ld;
#pragma unroll
            for (int co = 0; co < COUT; ++co) yp[co] = acc[i][co][0] + (bias ? bias[co] : 0.f);
        }
    }
}

}  // namespace

// w9: the conv's weight as [tap 0..8][cout][cin] (a transposed copy of the OIHW parameter the caller keeps in step with it)
extern "C" int sgd_conv3_narrow_out(const float* x, const float* pa, const float* pb, int32_t silu, const float* w9,
                                    const float* bias, float* y, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout,
                                    int32_t y_ld, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !w9 || !y || n <= 0 || h <= 0 || wd <= 0 || cin <= 0 || (cin & 31) || (cout != 3 && cout != 4) || y_ld < cout ||
        (pa != nullptr) != (pb != nullptr))
        return SGD_ERR_ARG;
    const int tiles_x = (wd + NO_TX - 1) / NO_TX, tiles_y = (h + NO_TY - 1) / NO_TY;
    const long grid = (long)n * tiles_x * tiles_y;
    if (grid > 0x7fffffffL) return SGD_ERR_ARG;
    if (cout == 3)
        hipLaunchKernelGGL((conv3_narrow_out_kernel<3>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, pa, pb, silu, w9,
                           bias, y, h, wd, cin, y_ld, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL((conv3_narrow_out_kernel<4>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, pa, pb, silu, w9,
                           bias, y, h, wd, cin, y_ld, tiles_x, tiles_y);
    return sgd_check_launch();
}
