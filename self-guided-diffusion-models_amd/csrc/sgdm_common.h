// Shared device helpers for the sgdm HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define SGD_WAVE 64

// The fp32 value about to be split into 16-bit hi + lo halves must be ONE rounded fp32 number.  If it is visibly a
// product, the backend may form hi with v_fma_mixlo_f16 (one rounding of the exact product) at one use and with
// v_cvt(_pk)_f16_f32 of the rounded product at another; the two differ by an fp16 ulp whenever the fp32 rounding
// crosses an fp16 tie (probability 2^-13 per element), and lo = v - hi is then paired with the wrong hi: a 2^-11
// relative error where the split promises 2^-22 (found in round 2: 1 attention row in 2,000 off by 3e-5).
// SGD_ROUNDED pins the value in a register without emitting an instruction.
#if defined(__HIP_DEVICE_COMPILE__)
#define SGD_ROUNDED(x) asm("" : "+v"(x))
#else
#define SGD_ROUNDED(x) ((void)0)
#endif

__device__ __forceinline__ float sgd_silu(float v) {
    // x * sigmoid(x); v_exp_f32 and v_rcp_f32 are 1 ulp each; the reference uses aten silu (fp32)
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

// counter-based dropout mask (include/sgdm_hip.h: sgd_igemm_args.drop_p).  One 32-bit hash serves TWO elements: element idx
// takes the (idx & 1)-th 16-bit half of the hash of pair idx >> 1, and is kept iff that half is >= p * 65536.  (Round 5: a hash
// per element was nine quarter-rate multiplies per channel quad in every kernel that recomputes the mask -- conv loaders,
// weight-gradient staging, both GroupNorm-backward passes; the HBM-bound reduce pass ran at 3.7 instead of 4.9 TB/s with it.)
__device__ __forceinline__ uint32_t sgd_drop_hash(uint32_t seed, long pair) {
    uint32_t h = seed ^ ((uint32_t)pair * 0x9E3779B1u) ^ ((uint32_t)((unsigned long)pair >> 32) * 0x632BE5ABu);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t sgd_drop_thr(float p) { return (uint32_t)(p * 65536.f); }
__device__ __forceinline__ bool sgd_drop_keep(uint32_t seed, long idx, uint32_t thr16) {
    const uint32_t h = sgd_drop_hash(seed, idx >> 1);
    return ((idx & 1) ? (h >> 16) : (h & 0xFFFFu)) >= thr16;
}
// four consecutive elements from index `base`: kept values times 1 / (1 - p), dropped ones zero
__device__ __forceinline__ f32x4 sgd_drop4(f32x4 v, float p, uint32_t seed, long base) {
    const uint32_t thr = sgd_drop_thr(p);
    const float inv = 1.0f / (1.0f - p);
    if ((base & 1) == 0) {                  // every 16-byte path (channel counts that are multiples of four): two hashes
        const uint32_t h0 = sgd_drop_hash(seed, base >> 1), h1 = sgd_drop_hash(seed, (base >> 1) + 1);
        v[0] = (h0 & 0xFFFFu) >= thr ? v[0] * inv : 0.f;
        v[1] = (h0 >> 16) >= thr ? v[1] * inv : 0.f;
        v[2] = (h1 & 0xFFFFu) >= thr ? v[2] * inv : 0.f;
        v[3] = (h1 >> 16) >= thr ? v[3] * inv : 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sgd_drop_keep(seed, base + j, thr) ? v[j] * inv : 0.f;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// error codes returned by every extern "C" launcher
#define SGD_OK 0
#define SGD_ERR_ARG 1
#define SGD_ERR_LAUNCH 2

// torch (or anyone) may leave a non-sticky error (e.g. hipErrorNotReady from an event query) in the
// thread's last-error slot: clear it before launching so sgd_check_launch reports OUR launch only.
#define SGD_CLEAR_ERR() (void)hipGetLastError()

static inline int sgd_check_launch() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return SGD_OK;
    fprintf(stderr, "sgdm_hip: kernel launch failed: %s (%d)\n", hipGetErrorString(e), (int)e);
    return SGD_ERR_LAUNCH;
}
