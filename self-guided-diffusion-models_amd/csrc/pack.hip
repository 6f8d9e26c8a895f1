// Weight packing for the fused implicit-GEMM conv / linear kernel (csrc/igemm.hip): OIHW parameters -> MFMA fragment order, per-tensor
// power-of-two scales of the split modes, and the batched re-pack of a whole program's weights (three launches per training step).
// Replaces (reference): nothing -- the reference hands OIHW tensors to ATen; this is the layout side of include/sgdm_hip.h's sgd_igemm.
#include "igemm_shared.h"
#include "prologue.h"

namespace {

// ---------------------------------------------------------------------------------------------
// weight packing:  OIHW [cout, cin, k, k]  ->  MFMA fragment order, so that the compute waves load their operands
// straight from global memory with fully coalesced 16-byte-per-lane loads (no LDS staging of weights).
//
//   unit(chunk, tap, nb) = 4 KiB holding the 32 output channels nb*32.. x 32 input channels chunk*32.. of one tap,
//   units ordered [chunk][tap][nb] (nb over ALL cout_p / 32 blocks: a K step of the stream is contiguous).
//   split modes: unit = [ks 0..1][hi | lo][lane 0..63][8 x 16-bit]   lane = lh*32 + li holds W[nb*32+li][chunk*32+ks*16+lh*8 .. +7]
//   f32        : unit = [ks 0..3][lane 0..63][4 x f32]               lane = lh*32 + li holds W[nb*32+li][chunk*32+ks*8+lh*4 .. +3]
//   (exactly the A-operand lane map of v_mfma_f32_32x32x16_f16 / four v_mfma_f32_32x32x2_f32 k-pairs).
// One thread produces one lane's 16 bytes (f32) or its hi AND lo 16 bytes (split) of one sub-step.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2_scale_of(uint32_t amax_bits) {
    // 2^k with max|w| * 2^k in [1, 2): k = -(exponent of amax); all-zero / non-finite tensors: 1
    const float amax = __uint_as_float(amax_bits);
    // subnormal maxima count as zero: 2^(1-e) would overflow to +inf for e <= -127 (1/inf = 0 -> NaN weights)
    if (!(amax >= 1.17549435e-38f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    frexpf(amax, &e);                                          // amax = m * 2^e, m in [0.5, 1)
    return ldexpf(1.f, 1 - e);                                 // e >= -125: at most 2^126
}

// max |w| of a tensor: every thread takes 16 elements as four independent 16-byte loads (the one-element grid-stride loop
// this replaces was a chain of dependent-latency iterations: 20 us per conv weight, 142 tensors per training step)
__device__ __forceinline__ void weight_amax_body(const float* __restrict__ w, long count, uint32_t* __restrict__ amax_bits, int vec,
                                                 int blk, int nblk) {
    float m = 0.f;
    if (vec) {
        const long nq = count >> 2;                                        // float4 quads
        const long q0 = (blk * (long)blockDim.x + threadIdx.x) * 4;
        const long step = (long)nblk * blockDim.x * 4;
        for (long q = q0; q < nq; q += step) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long qq = q + j < nq ? q + j : nq - 1;               // clamped: a duplicate does not change a maximum
                v[j] = *reinterpret_cast<const f32x4*>(w + qq * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[j][e]));
        }
        for (long i = (nq << 2) + blk * (long)blockDim.x + threadIdx.x; i < count; i += (long)nblk * blockDim.x)
            m = fmaxf(m, fabsf(w[i]));
    } else {
        for (long i = blk * (long)blockDim.x + threadIdx.x; i < count; i += (long)nblk * blockDim.x) m = fmaxf(m, fabsf(w[i]));
    }
    // one atomic per BLOCK: 2,300 same-address atomics (one per wave) took longer than reading the tensor
    m = wave_max(m);
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}
__global__ void weight_amax_kernel(const float* __restrict__ w, long count, uint32_t* __restrict__ amax_bits, int vec) {
    weight_amax_body(w, count, amax_bits, vec, blockIdx.x, gridDim.x);
}

template <int PREC>
__device__ __forceinline__ void pack_weight_body(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin,
                                                 int ks, int cout_p, int cin_p, int transpose, const uint32_t* __restrict__ amax_bits,
                                                 float* __restrict__ scale_inv_out, int blk, int grid_blocks) {
    const float wscale = amax_bits ? pow2_scale_of(*amax_bits) : 1.f;
    if (scale_inv_out && blk == 0 && threadIdx.x == 0) *scale_inv_out = 1.f / wscale;     // exact: power of two
    constexpr int NKS = PREC == SGD_PREC_F32 ? 4 : 2;         // sub-steps per unit
    constexpr int CPL = PREC == SGD_PREC_F32 ? 4 : 8;         // input channels per lane per sub-step
    const int kk = ks * ks, nblk = cout_p >> 5;
    const long total = (long)(cin_p >> 5) * kk * nblk * NKS * 64;
    for (long i = blk * (long)blockDim.x + threadIdx.x; i < total; i += (long)grid_blocks * blockDim.x) {
        const int lane = i & 63;
        long t = i >> 6;
        const int sub = t % NKS; t /= NKS;
        const long unit = t;
        const int nb = t % nblk; t /= nblk;
        const int tap = t % kk;
        const int chunk = t / kk;
        const int co = nb * 32 + (lane & 31);
        const int ci0 = chunk * 32 + sub * (2 * CPL) + (lane >> 5) * CPL;
        float v[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int ci = ci0 + j;
            float x = 0.f;
            if (ci < cin && co < cout) {
                // forward: W[co][ci][tap].  dgrad (adjoint conv): the packed "output" index co walks W's input channels,
                // "input" index ci walks W's output channels, taps are flipped.
                if (!transpose) x = src[((long)co * cin + ci) * kk + tap];
                else x = src[((long)ci * cout + co) * kk + (kk - 1 - tap)];
            }
            v[j] = x * wscale;
        }
        if constexpr (PREC == SGD_PREC_F32) {
            *reinterpret_cast<f32x4*>(dst + unit * 1024 + sub * 256 + lane * 4) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            typedef typename Split<PREC>::T T;
            typedef T T8 __attribute__((ext_vector_type(8)));
            T8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                T hj, lj;
                Split<PREC>::split(v[j], hj, lj);
                h[j] = hj;
                l[j] = lj;
            }
            T* up = reinterpret_cast<T*>(dst + unit * 1024) + sub * 1024 + lane * 8;
            *reinterpret_cast<T8*>(up) = h;
            *reinterpret_cast<T8*>(up + 512) = l;
        }
    }
}
template <int PREC>
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin,
                                   int ks, int cout_p, int cin_p, int transpose, const uint32_t* __restrict__ amax_bits,
                                   float* __restrict__ scale_inv_out) {
    pack_weight_body<PREC>(src, dst, cout, cin, ks, cout_p, cin_p, transpose, amax_bits, scale_inv_out, blockIdx.x, gridDim.x);
}

// ---- every weight of a training step in three launches (sgd_pack_weights_batched): the per-step re-pack was 142 pack +
// 74 amax launches of ~6 us each for 1.2 GB of traffic that takes 0.25 ms at HBM speed
__global__ void pack_zero_amax_kernel(const sgd_pack_job* __restrict__ jobs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && jobs[i].own_amax && jobs[i].amax_bits) *jobs[i].amax_bits = 0u;
}
__global__ void weight_amax_batched_kernel(const sgd_pack_job* __restrict__ jobs, const int32_t* __restrict__ block_job,
                                           const int32_t* __restrict__ first) {
    const int j = block_job[blockIdx.x];
    const sgd_pack_job jb = jobs[j];
    const long count = (long)jb.cout * jb.cin * jb.ksize * jb.ksize;
    weight_amax_body(jb.src, count, jb.amax_bits, (((uintptr_t)jb.src) & 15) == 0 && count >= 4, blockIdx.x - first[j], first[j + 1] - first[j]);
}
template <int PREC>
__global__ void pack_weight_batched_kernel(const sgd_pack_job* __restrict__ jobs, const int32_t* __restrict__ block_job,
                                           const int32_t* __restrict__ first) {
    const int j = block_job[blockIdx.x];
    const sgd_pack_job jb = jobs[j];
    // (the adjoint operator's dims are the transposed ones, as in sgd_pack_weight_scaled)
    const int co = jb.transpose ? jb.cin : jb.cout, ci = jb.transpose ? jb.cout : jb.cin;
    const int bn = (co % 128 == 0) ? 128 : 32;
    const int cout_p = ((co + bn - 1) / bn) * bn, cin_p = ((ci + KC - 1) / KC) * KC;
    pack_weight_body<PREC>(jb.src, reinterpret_cast<float*>(jb.dst), co, ci, jb.ksize, cout_p, cin_p, jb.transpose, jb.amax_bits,
                           jb.scale_inv, blockIdx.x - first[j], first[j + 1] - first[j]);
}

}  // namespace

static inline int pick_bn(int cout) { return (cout % 128 == 0) ? 128 : 32; }

extern "C" int64_t sgd_packed_weight_bytes(int32_t cout, int32_t cin, int32_t ksize, int32_t prec) {
    (void)prec;
    const int bn = pick_bn(cout);
    const int64_t cout_p = (int64_t)((cout + bn - 1) / bn) * bn;
    const int64_t cin_p = (int64_t)((cin + KC - 1) / KC) * KC;
    return (int64_t)ksize * ksize * cout_p * cin_p * 4;
}

static int pack_weight_impl(const float* w_src, void* w_dst, int32_t cout, int32_t cin, int32_t ksize, int32_t prec,
                            int32_t* cin_p_out, int32_t* cout_p_out, int transpose, void* stream,
                            const uint32_t* amax_bits = nullptr, float* scale_inv_out = nullptr) {
    SGD_CLEAR_ERR();
    if (!w_src || !w_dst || cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return SGD_ERR_ARG;
    const int bn = pick_bn(cout);
    const int cout_p = ((cout + bn - 1) / bn) * bn;
    const int cin_p = ((cin + KC - 1) / KC) * KC;
    if (cin_p_out) *cin_p_out = cin_p;
    if (cout_p_out) *cout_p_out = cout_p;
    const long total = (long)ksize * ksize * cout_p * cin_p / (prec == SGD_PREC_F32 ? 4 : 8);     // 16-byte vectors
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipStream_t st = (hipStream_t)stream;
    float* dst = reinterpret_cast<float*>(w_dst);
    if (prec == SGD_PREC_F32) hipLaunchKernelGGL((pack_weight_kernel<SGD_PREC_F32>), dim3(grid), dim3(256), 0, st, w_src, dst, cout, cin, ksize, cout_p, cin_p, transpose, amax_bits, scale_inv_out);
    else if (prec == SGD_PREC_F16X3) hipLaunchKernelGGL((pack_weight_kernel<SGD_PREC_F16X3>), dim3(grid), dim3(256), 0, st, w_src, dst, cout, cin, ksize, cout_p, cin_p, transpose, amax_bits, scale_inv_out);
    else if (prec == SGD_PREC_BF16X3) hipLaunchKernelGGL((pack_weight_kernel<SGD_PREC_BF16X3>), dim3(grid), dim3(256), 0, st, w_src, dst, cout, cin, ksize, cout_p, cin_p, transpose, amax_bits, scale_inv_out);
    else return SGD_ERR_ARG;
    return sgd_check_launch();
}

extern "C" int sgd_pack_weight(const float* w_src, void* w_dst, int32_t cout, int32_t cin, int32_t ksize,
                               int32_t prec, int32_t* cin_p_out, int32_t* cout_p_out, void* stream) {
    return pack_weight_impl(w_src, w_dst, cout, cin, ksize, prec, cin_p_out, cout_p_out, 0, stream);
}

// weights of the adjoint convolution (dgrad): w_src is the FORWARD weight [cout_fwd, cin_fwd, k, k]; the packed
// operator maps cout_fwd input channels to cin_fwd output channels with flipped taps.
extern "C" int sgd_pack_weight_dgrad(const float* w_src, void* w_dst, int32_t cout_fwd, int32_t cin_fwd, int32_t ksize,
                                     int32_t prec, int32_t* cin_p_out, int32_t* cout_p_out, void* stream) {
    return pack_weight_impl(w_src, w_dst, cin_fwd, cout_fwd, ksize, prec, cin_p_out, cout_p_out, 1, stream);
}

extern "C" int sgd_weight_amax(const float* w, int64_t count, uint32_t* amax_bits, void* stream) {
    SGD_CLEAR_ERR();
    if (!w || !amax_bits || count <= 0) return SGD_ERR_ARG;
    const int vec = (((uintptr_t)w) & 15) == 0 && count >= 4;
    long grid = vec ? (count + 4095) / 4096 : (count + 1023) / 1024;       // 16 elements per thread on the vector path
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(weight_amax_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w, (long)count, amax_bits, vec);
    return sgd_check_launch();
}

extern "C" int sgd_pack_weight_scaled(const float* w_src, void* w_dst, int32_t cout, int32_t cin, int32_t ksize, int32_t prec,
                                      int32_t transpose, const uint32_t* amax_bits, float* scale_inv_out, int32_t* cin_p_out,
                                      int32_t* cout_p_out, void* stream) {
    if (!amax_bits || !scale_inv_out) return SGD_ERR_ARG;
    if (transpose) return pack_weight_impl(w_src, w_dst, cin, cout, ksize, prec, cin_p_out, cout_p_out, 1, stream, amax_bits, scale_inv_out);
    return pack_weight_impl(w_src, w_dst, cout, cin, ksize, prec, cin_p_out, cout_p_out, 0, stream, amax_bits, scale_inv_out);
}

extern "C" int sgd_pack_job_blocks(int32_t cout, int32_t cin, int32_t ksize, int32_t prec, int32_t transpose, int32_t* amax_blocks,
                                   int32_t* pack_blocks, int32_t* cin_p_out, int32_t* cout_p_out) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return SGD_ERR_ARG;
    const int co = transpose ? cin : cout, ci = transpose ? cout : cin;
    const int bn = pick_bn(co);
    const int cout_p = ((co + bn - 1) / bn) * bn, cin_p = ((ci + KC - 1) / KC) * KC;
    const long count = (long)cout * cin * ksize * ksize;
    long ab = (count + 4095) / 4096;
    if (ab > 256) ab = 256;
    const long total = (long)ksize * ksize * cout_p * cin_p / (prec == SGD_PREC_F32 ? 4 : 8);
    long pb = (total + 255) / 256;
    if (pb > 4096) pb = 4096;
    if (amax_blocks) *amax_blocks = (int32_t)ab;
    if (pack_blocks) *pack_blocks = (int32_t)pb;
    if (cin_p_out) *cin_p_out = cin_p;
    if (cout_p_out) *cout_p_out = cout_p;
    return SGD_OK;
}

extern "C" int sgd_pack_weights_batched(const sgd_pack_job* jobs, int32_t n_jobs, const int32_t* amax_block_job,
                                        const int32_t* amax_first, int32_t n_amax_blocks, const int32_t* pack_block_job,
                                        const int32_t* pack_first, int32_t n_pack_blocks, int32_t prec, void* stream) {
    SGD_CLEAR_ERR();
    if (!jobs || n_jobs <= 0 || !pack_block_job || !pack_first || n_pack_blocks <= 0) return SGD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (prec != SGD_PREC_F32 && n_amax_blocks > 0) {
        if (!amax_block_job || !amax_first) return SGD_ERR_ARG;
        hipLaunchKernelGGL(pack_zero_amax_kernel, dim3((n_jobs + 255) / 256), dim3(256), 0, st, jobs, n_jobs);
        hipLaunchKernelGGL(weight_amax_batched_kernel, dim3(n_amax_blocks), dim3(256), 0, st, jobs, amax_block_job, amax_first);
    }
    if (prec == SGD_PREC_F32) hipLaunchKernelGGL((pack_weight_batched_kernel<SGD_PREC_F32>), dim3(n_pack_blocks), dim3(256), 0, st, jobs, pack_block_job, pack_first);
    else if (prec == SGD_PREC_F16X3) hipLaunchKernelGGL((pack_weight_batched_kernel<SGD_PREC_F16X3>), dim3(n_pack_blocks), dim3(256), 0, st, jobs, pack_block_job, pack_first);
    else if (prec == SGD_PREC_BF16X3) hipLaunchKernelGGL((pack_weight_batched_kernel<SGD_PREC_BF16X3>), dim3(n_pack_blocks), dim3(256), 0, st, jobs, pack_block_job, pack_first);
    else return SGD_ERR_ARG;
    return sgd_check_launch();
}

