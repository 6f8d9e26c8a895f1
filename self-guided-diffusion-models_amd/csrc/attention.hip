// Fused softmax(scale * q k^T) v for the two attention flavours of the reference (gfx950):
//   * legacy QKV self-attention   QKVAttentionLegacy.forward   openaimodel.py:403-420
//   * multi-query attention over [context | null | self] keys   Attention_LR.forward  crossattetion_lr.py:90-139
//
// One block = 4 waves = 128 queries of one (batch, head); K/V are staged through LDS in tiles of
// 64 keys (works for any key count; 256 and 273 in the shipped configs).  Exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32): the cores are ~1 % of the UNet FLOPs.
//
// Trick: compute S^T = K Q^T so that an accumulator lane owns ONE query column and its registers
// walk the keys.  Then (a) the softmax row statistics are register-local + one xor-32 shuffle,
// (b) P^T is already in B-operand position for O^T = V^T P^T -- no LDS round trip, no transposes --
// and (c) the online-softmax rescale of O^T is one scalar per lane.
#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"

namespace {

constexpr int KT = 64;   // keys per LDS tile

template <int D>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, int q_ld, int q_hs,
                                                        const float* __restrict__ k, const float* __restrict__ v,
                                                        int kv_ld, int kv_hs, int tq, int tk, float scale,
                                                        float* __restrict__ out, int out_ld) {
    constexpr int LD = D + 4;
    constexpr int DT = (D + 31) / 32;             // 32-row tiles of the O^T accumulator
    __shared__ __attribute__((aligned(16))) float Ks[KT * LD];
    __shared__ __attribute__((aligned(16))) float Vs[KT * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 128 + wave * 32 + li;      // this lane's query
    const int qc = qi < tq ? qi : tq - 1;

    // Q fragment: for k-step group s (8 channels) lane half lh holds channels 8s+4lh .. +3
    f32x4 qreg[D / 8];
    {
        const float* qp = q + ((long)b * tq + qc) * q_ld + head * q_hs;
#pragma unroll
        for (int s = 0; s < D / 8; ++s) {
            f32x4 t = *reinterpret_cast<const f32x4*>(qp + s * 8 + lh * 4);
            qreg[s] = t * scale;
        }
    }

    f32x16 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const float* kb = k + (long)b * tk * kv_ld + head * kv_hs;
    const float* vb = v + (long)b * tk * kv_ld + head * kv_hs;

    for (int kt0 = 0; kt0 < tk; kt0 += KT) {
        __syncthreads();
        // stage K and V tile rows [kt0, kt0+KT)
        constexpr int VPR = D / 4;                         // float4 per row
        for (int idx = tid; idx < KT * VPR; idx += 256) {
            const int row = idx / VPR, c4 = idx % VPR;
            f32x4 kv4 = {0.f, 0.f, 0.f, 0.f}, vv4 = {0.f, 0.f, 0.f, 0.f};
            if (kt0 + row < tk) {
                kv4 = *reinterpret_cast<const f32x4*>(kb + (long)(kt0 + row) * kv_ld + c4 * 4);
                vv4 = *reinterpret_cast<const f32x4*>(vb + (long)(kt0 + row) * kv_ld + c4 * 4);
            }
            *reinterpret_cast<f32x4*>(Ks + row * LD + c4 * 4) = kv4;
            *reinterpret_cast<f32x4*>(Vs + row * LD + c4 * 4) = vv4;
        }
        __syncthreads();

#pragma unroll
        for (int st = 0; st < KT / 32; ++st) {
            if (kt0 + st * 32 >= tk) break;
            // S^T tile [32 keys x 32 queries]
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < D / 8; ++s) {
                f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (st * 32 + li) * LD + s * 8 + lh * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qreg[s][j], sacc, 0, 0, 0);
            }
            // mask keys beyond tk, tile max
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt0 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (key >= tk) sacc[r] = -INFINITY;
                mx = fmaxf(mx, sacc[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = __expf(sacc[r] - m_new);
                psum += sacc[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                // O^T[dt] += V^T P^T : MFMA r contracts keys {k0, k0+4}, k0 = (r&3) + 8(r>>2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = st * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float vf = 0.f;
                    if (D >= 32 || li < D) vf = Vs[key * LD + dt * 32 + li];
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sacc[r], oacc[dt], 0, 0, 0);
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qi < tq) {
        float* op = out + ((long)b * tq + qi) * out_ld + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dd = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (dd < D) op[dd] = oacc[dt][r] * inv;
            }
    }
}

}  // namespace

extern "C" int sgd_attention(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                             int32_t kv_ld, int32_t kv_hs, int32_t batch, int32_t heads, int32_t tq, int32_t tk,
                             int32_t d, float scale, float* out, int32_t out_ld, void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !out || batch <= 0 || heads <= 0 || tq <= 0 || tk <= 0) return SGD_ERR_ARG;
    if ((q_ld & 3) || (q_hs & 3) || (kv_ld & 3) || (kv_hs & 3)) return SGD_ERR_ARG;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return SGD_ERR_ARG;
    dim3 grid((tq + 127) / 128, heads, batch);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 16: hipLaunchKernelGGL((attention_kernel<16>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld); break;
        case 32: hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld); break;
        case 64: hipLaunchKernelGGL((attention_kernel<64>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld); break;
        default: return SGD_ERR_ARG;
    }
    return sgd_check_launch();
}
