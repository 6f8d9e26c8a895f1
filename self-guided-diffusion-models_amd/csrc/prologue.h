// Loader-side helpers shared by the forward implicit-GEMM kernel and the weight-gradient kernel:
// raw input fetch over a virtual channel concat and the fused prologue (GroupNorm apply / LayerNorm, SiLU).
#pragma once
#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// 16-bit element type of the split-precision modes (fp32 x = hi + lo, both 16-bit; three MFMA products)
template <int PREC> struct Split;
template <> struct Split<SGD_PREC_F16X3> {
    typedef _Float16 T;
    static __device__ __forceinline__ T hi(float v) { return (T)v; }
    static __device__ __forceinline__ float back(T h) { return (float)h; }
    // v = h + l (+ 2^-22 |v|); SGD_ROUNDED (sgdm_common.h): h must be the same number at both of its uses
    static __device__ __forceinline__ void split(float v, T& h, T& l) {
        SGD_ROUNDED(v);
        h = (T)v;
        l = (T)(v - (float)h);
    }
};
template <> struct Split<SGD_PREC_BF16X3> {
    typedef __bf16 T;
    static __device__ __forceinline__ T hi(float v) { return (T)v; }
    static __device__ __forceinline__ float back(T h) { return (float)h; }
    static __device__ __forceinline__ void split(float v, T& h, T& l) {
        SGD_ROUNDED(v);
        h = (T)v;
        l = (T)(v - (float)h);
    }
};

// Four fp32 values -> packed f16 hi pairs and lo pairs in 8 vector instructions: v_cvt_pk_f16_f32 (RNE, two values per
// instruction) for hi, v_fma_mix_f32 for lo_f32 = v - float(hi) reading the f16 half directly (no v_cvt_f32_f16), and
// v_cvt_pk_f16_f32 again for lo.  The compiler's own sequence for the same arithmetic is 16 instructions (it converts hi
// twice: scalar for the subtraction, packed for the store); loader issue slots are what paces the conv kernel.
// h = {hi(v0) | hi(v1) << 16, hi(v2) | hi(v3) << 16}, l likewise.  Bit-identical to Split<F16X3>::split per element.
struct alignas(8) u32x2 { uint32_t x, y; };
__device__ __forceinline__ void split4_f16(f32x4 v, u32x2& h, u32x2& l) {
#if defined(__HIP_DEVICE_COMPILE__)
    float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], l0, l1, l2, l3;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h.x) : "v"(v0), "v"(v1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h.y) : "v"(v2), "v"(v3));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h.x), "v"(v0));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h.x), "v"(v1));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(h.y), "v"(v2));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(h.y), "v"(v3));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l.x) : "v"(l0), "v"(l1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l.y) : "v"(l2), "v"(l3));
#else
    h.x = h.y = l.x = l.y = 0;
    (void)v;
#endif
}

// raw input vector: 4 consecutive channels starting at c of source row `row` (virtual concat x0|x1)
template <bool VEC>
__device__ __forceinline__ f32x4 load_raw(const sgd_igemm_args& a, long row, int c) {
    f32x4 v;
    if (VEC) {
        // pointer select, ONE unconditional load: a branch around a load would make every later
        // s_waitcnt vmcnt of the calling loader conservative (the counter is in-order)
        const float* p = (c < a.c0) ? a.x0 + row * a.c0 + c : a.x1 + row * a.c1 + (c - a.c0);
        v = ld4(p);
    } else {
        const int ct = a.c0 + a.c1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int cc = c + j;
            float s = 0.f;
            if (cc < ct) s = (cc < a.c0) ? a.x0[row * a.c0 + cc] : a.x1[row * a.c1 + (cc - a.c0)];
            v[j] = s;
        }
    }
    return v;
}

// prologue coefficients of one item
struct Coef {
    f32x4 p, q;       // AFFINE_NC: a, b       LN_ROW: p[0] = mean, p[1] = rstd
};

template <bool VEC>
__device__ __forceinline__ Coef load_coef(const sgd_igemm_args& a, int n, long row, int c) {
    Coef k;
    k.p = f32x4{0.f, 0.f, 0.f, 0.f};
    k.q = k.p;
    const int ct = a.c0 + a.c1;
    if (a.pro == SGD_PRO_AFFINE_NC) {
        if (VEC) {
            k.p = ld4(a.pa + (long)n * ct + c);
            k.q = ld4(a.pb + (long)n * ct + c);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bool ok = c + j < ct;
                k.p[j] = ok ? a.pa[(long)n * ct + c + j] : 0.f;
                k.q[j] = ok ? a.pb[(long)n * ct + c + j] : 0.f;
            }
        }
    } else if (a.pro == SGD_PRO_LN_ROW) {
        k.p[0] = a.pa[row * 2];
        k.p[1] = a.pa[row * 2 + 1];
    }
    return k;
}

__device__ __forceinline__ f32x4 apply_pro(const sgd_igemm_args& a, f32x4 v, const Coef& k, int c, long row) {
    if (a.pro == SGD_PRO_AFFINE_NC) {
        v = v * k.p + k.q;
    } else if (a.pro == SGD_PRO_LN_ROW) {
        const int ct = a.c0 + a.c1;
        f32x4 g, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = c + j < ct;
            g[j] = ok ? a.pb[c + j] : 0.f;
            b[j] = (ok && a.pc) ? a.pc[c + j] : 0.f;
        }
        v = (v - k.p[0]) * k.p[1] * g + b;
    }
    if (a.pro_silu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sgd_silu(v[j]);
    }
    if (a.drop_p > 0.f) v = sgd_drop4(v, a.drop_p, a.drop_seed, row * (a.c0 + a.c1) + c);
    return v;
}

