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
                                                        float* __restrict__ out, int out_ld,
                                                        float* __restrict__ lse,
                                                        const uint8_t* __restrict__ kmask = nullptr) {
    // kmask [batch, tk], 1 = attend (sgd_attention_masked): a masked key gets -inf like the keys past tk, i.e. weight 0
    // -- the result of masked_fill(~mask, -FLT_MAX) before the softmax (attention_ldm.py:246-249) whenever at least one
    // key of the row is valid.  Key 0 must be valid (the callers' null key is: the mask is padded with True in front).
    constexpr int LD = D + 4;
    constexpr int DT = (D + 31) / 32;             // 32-row tiles of the O^T accumulator
    constexpr int KT = D > 64 ? 32 : 64;          // keys per LDS tile (head dim 128 of the *_s64 widths: 2 x 32 x 132 floats)
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
            // mask keys beyond tk (and the caller's masked keys), tile max.  The mask bytes are gathered into a bit set
            // first, from clamped addresses: written as one short-circuit condition around the element assignment
            // (key >= tk || (kmask && !kmask[..])) hipcc 7.2 dropped the assignment for the masked lanes altogether.
            unsigned dead = 0;
            if (kmask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt0 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const uint8_t keep = kmask[(long)b * tk + (key < tk ? key : tk - 1)];
                    dead |= (keep ? 0u : 1u) << r;
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt0 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const bool drop = key >= tk || ((dead >> r) & 1u);
                sacc[r] = drop ? -INFINITY : sacc[r];
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
    if (lse && qi < tq && lh == 0) lse[((long)b * gridDim.y + head) * tq + qi] = m_run + __logf(l_tot);
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


// =============================================================================================
// split-precision forward (prec f16x3): the same S^T = K Q^T / O^T = V^T P^T orientation on
// v_mfma_f32_32x32x16_f16 with every fp32 operand carried as hi + lo 16-bit halves and three products
// (hi.hi + hi.lo + lo.hi, fp32 accumulate) -- 24 MFMAs of 32 cycles per 32-key tile at D = 64 instead of 64 MFMAs
// of 64 cycles.  K is staged row-major as two f16 planes (16-byte fragment reads); V is staged TRANSPOSED,
// [d][key], in the k order the accumulator-as-operand idiom imposes: the P^T fragment of k-step s is registers
// 8s..8s+7 of the score accumulator, whose element j on lane half h is key 16s + 8(j>>2) + 4h + (j&3); V^T's
// fragment must present the same key at the same (h, j), so key kk = 8a + 4h + c of a 16-key group is stored at
// position 8h + 4a + c.  P is scaled by 2^14 before the split (its lo half would otherwise sit in fp16's
// subnormal range); the scale cancels in O / l.
// =============================================================================================
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    SGD_ROUNDED(v);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

template <int D>
__global__ __launch_bounds__(256) void attention_split_kernel(const float* __restrict__ q, int q_ld, int q_hs,
                                                              const float* __restrict__ k, const float* __restrict__ v,
                                                              int kv_ld, int kv_hs, int tq, int tk, float scale,
                                                              float* __restrict__ out, int out_ld,
                                                              float* __restrict__ lse) {
    constexpr int DT = (D + 31) / 32;             // 32-row tiles of the O^T accumulator
    constexpr int DP = DT * 32;                   // rows of the V^T image (zero beyond D)
    constexpr int KS = (D + 15) / 16;             // 16-channel k steps of the score product
    constexpr int DK = KS * 16;
    constexpr int KT = D > 64 ? 32 : 64;          // keys per LDS tile
    constexpr int LDK = DK + 8;                   // f16 elements per K row (16-byte aligned, 4-bank skew)
    constexpr int LDV = KT + 8;                   // f16 elements per V^T row
    __shared__ __attribute__((aligned(16))) _Float16 Kh[KT * LDK];
    __shared__ __attribute__((aligned(16))) _Float16 Kl[KT * LDK];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[DP * LDV];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[DP * LDV];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 128 + wave * 32 + li;      // this lane's query
    const int qc = qi < tq ? qi : tq - 1;

    // Q^T fragments (B operand): k step s, lane half lh: channels 16s + 8lh .. +7 of query li
    f16x8 qh[KS], ql[KS];
    {
        const float* qp = q + ((long)b * tq + qc) * q_ld + head * q_hs;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                const int c = s * 16 + lh * 8 + j4 * 4;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (c < D) t = *reinterpret_cast<const f32x4*>(qp + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 h, l;
                    split_f16(t[j] * scale, h, l);
                    qh[s][j4 * 4 + j] = h;
                    ql[s][j4 * 4 + j] = l;
                }
            }
    }

    f32x16 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr float PSHIFT = 14.0f;               // P is carried times 2^14

    const float* kb = k + (long)b * tk * kv_ld + head * kv_hs;
    const float* vb = v + (long)b * tk * kv_ld + head * kv_hs;

    if (DP > D) {                                 // rows of the V^T image no channel writes (D = 16)
        for (int idx = tid; idx < (DP - D) * LDV; idx += 256) {
            Vh[D * LDV + idx] = (_Float16)0.f;
            Vl[D * LDV + idx] = (_Float16)0.f;
        }
    }

    // Every K / V quad of a tile is requested (clamped addresses, no branch around a load) before the first one is split
    // and stored: with the bounds check around the load each item was load -> s_waitcnt vmcnt(0) -> store, six memory
    // latencies in a row per 64-key tile.  Round 4: the requests of tile t + 1 go out BEFORE the MFMA phase of tile t and
    // are split into LDS after it (8 row quads per thread in registers): the staging latency was exposed once per tile.
    constexpr int VPR = D / 4;
    constexpr int NKI = (KT * VPR + 255) / 256;                // K quads per thread
    constexpr int NVI = ((KT / 2) * VPR + 255) / 256;          // V key-pair quads per thread
    f32x4 kq[NKI], vq0[NVI], vq1[NVI];
    const int klast = tk - 1;
    auto request = [&](int kt0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NKI; ++i) {
            int idx = tid + i * 256;
            idx = idx < KT * VPR ? idx : KT * VPR - 1;
            const int row = idx / VPR, c4 = idx % VPR;
            const int key = kt0 + row < tk ? kt0 + row : klast;
            kq[i] = *reinterpret_cast<const f32x4*>(kb + (long)key * kv_ld + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NVI; ++i) {
            int idx = tid + i * 256;
            idx = idx < (KT / 2) * VPR ? idx : (KT / 2) * VPR - 1;
            const int key = kt0 + (idx / VPR) * 2, c4 = idx % VPR;
            vq0[i] = *reinterpret_cast<const f32x4*>(vb + (long)(key < tk ? key : klast) * kv_ld + c4 * 4);
            vq1[i] = *reinterpret_cast<const f32x4*>(vb + (long)(key + 1 < tk ? key + 1 : klast) * kv_ld + c4 * 4);
        }
    };
    request(0);
    for (int kt0 = 0; kt0 < tk; kt0 += KT) {
        __syncthreads();
        // K rows: one float4 -> 4 hi + 4 lo (8-byte stores)
#pragma unroll
        for (int i = 0; i < NKI; ++i) {
            const int idx = tid + i * 256;
            if (idx >= KT * VPR) break;
            const int row = idx / VPR, c4 = idx % VPR;
            f32x4 t = kq[i];
            if (kt0 + row >= tk) t = f32x4{0.f, 0.f, 0.f, 0.f};
            f16x4 h4, l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 h, l;
                split_f16(t[j], h, l);
                h4[j] = h;
                l4[j] = l;
            }
            *reinterpret_cast<f16x4*>(Kh + row * LDK + c4 * 4) = h4;
            *reinterpret_cast<f16x4*>(Kl + row * LDK + c4 * 4) = l4;
        }
        // V^T: a thread takes two neighbouring keys (neighbours in the permuted order too) x 4 channels
#pragma unroll
        for (int i = 0; i < NVI; ++i) {
            const int idx = tid + i * 256;
            if (idx >= (KT / 2) * VPR) break;
            const int kp = idx / VPR, c4 = idx % VPR;
            const int key = kp * 2;
            f32x4 t0 = vq0[i], t1 = vq1[i];
            if (kt0 + key >= tk) t0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt0 + key + 1 >= tk) t1 = f32x4{0.f, 0.f, 0.f, 0.f};
            const int kk = key & 15;
            const int pos = (key & ~15) + (((kk >> 2) & 1) << 3) + ((kk >> 3) << 2) + (kk & 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 h0, l0, h1, l1;
                split_f16(t0[j], h0, l0);
                split_f16(t1[j], h1, l1);
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                f16x2 hh = {h0, h1}, ll = {l0, l1};
                *reinterpret_cast<f16x2*>(Vh + (c4 * 4 + j) * LDV + pos) = hh;
                *reinterpret_cast<f16x2*>(Vl + (c4 * 4 + j) * LDV + pos) = ll;
            }
        }
        if (kt0 + KT < tk) request(kt0 + KT);                 // in flight under this tile's MFMA phase
        __syncthreads();

#pragma unroll
        for (int st = 0; st < KT / 32; ++st) {
            if (kt0 + st * 32 >= tk) break;
            // S^T tile [32 keys x 32 queries]
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(Kh + (st * 32 + li) * LDK + s * 16 + lh * 8);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(Kl + (st * 32 + li) * LDK + s * 16 + lh * 8);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sacc, 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt0 + st * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (key >= tk) sacc[r] = -INFINITY;
                mx = fmaxf(mx, sacc[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            float psum = 0.f;
            const float mb = PSHIFT - m_new * LOG2E;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], LOG2E, mb));
                psum += sacc[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
            // P^T fragments of the two 16-key steps: registers 8s .. 8s+7
            f16x8 ph[2], pl[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    _Float16 h, l;
                    split_f16(sacc[s2 * 8 + j], h, l);
                    ph[s2][j] = h;
                    pl[s2][j] = l;
                }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(Vh + (dt * 32 + li) * LDV + st * 32 + s2 * 16 + lh * 8);
                    const f16x8 vl = *reinterpret_cast<const f16x8*>(Vl + (dt * 32 + li) * LDV + st * 32 + s2 * 16 + lh * 8);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s2], oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s2], oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s2], oacc[dt], 0, 0, 0);
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (lse && qi < tq && lh == 0) lse[((long)b * gridDim.y + head) * tq + qi] = m_run + __logf(l_tot) - PSHIFT * 0.6931471805599453f;
    if (qi < tq) {
        float* op = out + ((long)b * tq + qi) * out_ld + head * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = dt * 32 + 8 * g + 4 * lh;
                if (dd >= D) continue;
                f32x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = oacc[dt][g * 4 + j] * inv;
                *reinterpret_cast<f32x4*>(op + dd) = o4;
            }
    }
}


// =============================================================================================
// backward.  With LSE (from the forward) and D[q] = dO[q].O[q] both softmax statistics are known, so
// every (key tile, query tile) pair is independent:
//     S = scale q k^T, P = exp(S - LSE), dP = dO v^T, dS = P (dP - D)
//     dV += P^T dO,  dK += scale dS^T q,  dQ += scale dS k
// Two sweeps of ONE kernel shape (owner rows in registers as MFMA B fragments, the other side streamed
// through LDS), so no gradient is ever summed across waves / workgroups (no atomics, deterministic):
//   SWEEP 0 "kv": a wave owns 32 keys, loops over queries:  acc cols = key -> dK^T, dV^T accumulate over q
//   SWEEP 1 "q" : a wave owns 32 queries, loops over keys:  acc cols = q   -> dQ^T accumulates over keys
// =============================================================================================
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const float* __restrict__ o, int o_ld,
                                                            const float* __restrict__ dout, int dout_ld, int heads,
                                                            int tq, float* __restrict__ dvec) {
    // dvec[b, head, q] = sum_d dout[b,q,head*D+d] * o[b,q,head*D+d]
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;      // over b*heads*tq
    const long total = (long)gridDim.y * heads * tq;
    (void)total;
    const int b = blockIdx.y;
    if (i >= (long)heads * tq) return;
    const int head = i / tq, qi = i % tq;
    const float* op = o + ((long)b * tq + qi) * o_ld + head * D;
    const float* gp = dout + ((long)b * tq + qi) * dout_ld + head * D;
    float s = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4) {
        f32x4 a = *reinterpret_cast<const f32x4*>(op + d4 * 4), g = *reinterpret_cast<const f32x4*>(gp + d4 * 4);
        s += a[0] * g[0] + a[1] * g[1] + a[2] * g[2] + a[3] * g[3];
    }
    dvec[((long)b * heads + head) * tq + qi] = s;
}

// MASKED (sgd_attention_masked_bwd): kmask [batch, tk] bytes, 1 = attend -- a masked key had weight exactly 0 in the forward
// (and is absent from its log-sum-exp), so its P and dS are 0: no gradient into its key / value rows, none through it into q.
template <int D, int SWEEP, bool MASKED = false>
__global__ __launch_bounds__(256) void attention_bwd_kernel(
    const float* __restrict__ q, int q_ld, int q_hs, const float* __restrict__ k, const float* __restrict__ v,
    int kv_ld, int kv_hs, const float* __restrict__ dout, int dout_ld, const float* __restrict__ lse,
    const float* __restrict__ dvec, int tq, int tk, float scale, float* __restrict__ dq, float* __restrict__ dk,
    float* __restrict__ dv, int heads, int mq, const uint8_t* __restrict__ kmask = nullptr) {
    constexpr int LD = D + 4;
    constexpr int DT = (D + 31) / 32;
    constexpr int TT = D > 64 ? 32 : 64;                 // streamed rows per LDS tile (head dim 128: 2 x 32 x 132 floats)
    __shared__ __attribute__((aligned(16))) float Us[TT * LD];     // kv sweep: Q rows   | q sweep: K rows
    __shared__ __attribute__((aligned(16))) float Ws[TT * LD];     // kv sweep: dO rows  | q sweep: V rows
    __shared__ float Ls[TT], Ds[TT];                                // kv sweep: LSE / D of the streamed queries
    __shared__ float Ms[MASKED ? TT : 1];                           // q sweep, masked: 1.0 / 0.0 per streamed key

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // multi-query (kv_hs == 0): the kv sweep walks all heads inside one block so dK/dV of the shared keys are
    // summed in registers (gridDim.y == 1, `heads` passed explicitly); otherwise one head per block.
    const int b = blockIdx.z;
    const int head0 = mq ? 0 : blockIdx.y, head1 = mq ? heads : blockIdx.y + 1;
    const int own_n = SWEEP == 0 ? tk : tq;              // rows on the owner side
    const int str_n = SWEEP == 0 ? tq : tk;              // rows on the streamed side
    const int oi = blockIdx.x * 128 + wave * 32 + li;    // this lane's owned row (key or query)
    const int oc = oi < own_n ? oi : own_n - 1;
    float own_keep = 1.f;             // kv sweep, masked: validity of this lane's key
    if (MASKED && SWEEP == 0) own_keep = kmask[(long)b * tk + oc] ? 1.f : 0.f;

    f32x16 accA[DT], accB[DT];        // kv: dK^T, dV^T   q: dQ^T (accB unused)
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[i][r] = 0.f; accB[i][r] = 0.f; }
    int head = head0;
    for (; head < head1; ++head) {
    const float* qb = q + (long)b * tq * q_ld + head * q_hs;
    const float* kb = k + (long)b * tk * kv_ld + head * kv_hs;
    const float* vb = v + (long)b * tk * kv_ld + head * kv_hs;
    const float* gb = dout + (long)b * tq * dout_ld + head * D;
    const float* lseb = lse + ((long)b * heads + head) * tq;
    const float* dvb = dvec + ((long)b * heads + head) * tq;

    // owner fragments (MFMA B operands): X pairs with U (scores), Y pairs with W (dP)
    f32x4 xf[D / 8], yf[D / 8];
    {
        const float* xp = SWEEP == 0 ? kb + (long)oc * kv_ld : qb + (long)oc * q_ld;
        const float* yp = SWEEP == 0 ? vb + (long)oc * kv_ld : gb + (long)oc * dout_ld;
#pragma unroll
        for (int s = 0; s < D / 8; ++s) {
            xf[s] = *reinterpret_cast<const f32x4*>(xp + s * 8 + lh * 4);
            yf[s] = *reinterpret_cast<const f32x4*>(yp + s * 8 + lh * 4);
        }
    }
    float own_lse = 0.f, own_d = 0.f;
    if (SWEEP == 1) { own_lse = lseb[oc]; own_d = dvb[oc]; }

    for (int t0 = 0; t0 < str_n; t0 += TT) {
        __syncthreads();
        // all row quads of the tile requested (clamped rows, no branch around a load) before the first is stored
        constexpr int VPR = D / 4;
        constexpr int NI = (TT * VPR + 255) / 256;
        f32x4 ureg[NI], wreg[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int idx = tid + i * 256;
            idx = idx < TT * VPR ? idx : TT * VPR - 1;
            const int row = idx / VPR, c4 = idx % VPR;
            const long rr = t0 + row < str_n ? t0 + row : str_n - 1;
            if (SWEEP == 0) {
                ureg[i] = *reinterpret_cast<const f32x4*>(qb + rr * q_ld + c4 * 4);
                wreg[i] = *reinterpret_cast<const f32x4*>(gb + rr * dout_ld + c4 * 4);
            } else {
                ureg[i] = *reinterpret_cast<const f32x4*>(kb + rr * kv_ld + c4 * 4);
                wreg[i] = *reinterpret_cast<const f32x4*>(vb + rr * kv_ld + c4 * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + i * 256;
            if (idx >= TT * VPR) break;
            const int row = idx / VPR, c4 = idx % VPR;
            f32x4 u = ureg[i], w = wreg[i];
            if (t0 + row >= str_n) { u = f32x4{0.f, 0.f, 0.f, 0.f}; w = u; }
            *reinterpret_cast<f32x4*>(Us + row * LD + c4 * 4) = u;
            *reinterpret_cast<f32x4*>(Ws + row * LD + c4 * 4) = w;
        }
        if (SWEEP == 0 && tid < TT) {
            const int r = t0 + tid;
            Ls[tid] = r < str_n ? lseb[r] : 0.f;
            Ds[tid] = r < str_n ? dvb[r] : 0.f;
        }
        if (MASKED && SWEEP == 1 && tid < TT) {
            const int r = t0 + tid;
            Ms[tid] = (r < str_n && kmask[(long)b * tk + r]) ? 1.f : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < TT / 32; ++st) {
            if (t0 + st * 32 >= str_n) break;
            // scores / dP tiles: rows = streamed rows (registers), cols = owned rows (lanes)
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < D / 8; ++s) {
                const f32x4 uf = *reinterpret_cast<const f32x4*>(Us + (st * 32 + li) * LD + s * 8 + lh * 4);
                const f32x4 wf = *reinterpret_cast<const f32x4*>(Ws + (st * 32 + li) * LD + s * 8 + lh * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[j], xf[s][j], sacc, 0, 0, 0);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j], yf[s][j], pacc, 0, 0, 0);
                }
            }
            // P and dS (in place: sacc <- P, pacc <- dS)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = st * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;      // streamed row of this register
                const bool ok = (t0 + srow < str_n) && (oi < own_n);
                const float l = SWEEP == 0 ? Ls[srow] : own_lse;
                const float dd = SWEEP == 0 ? Ds[srow] : own_d;
                float p;
                if (MASKED) {
                    // a masked key is absent from the row's log-sum-exp, so its exponent has no bound: clamp it (valid keys
                    // have exponents <= 0 up to rounding) and multiply by 0 / 1 -- not a second condition on `ok`, see the
                    // note on masked lanes in attention_kernel
                    p = ok ? __expf(fminf(sacc[r] * scale - l, 0.f)) : 0.f;
                    p *= SWEEP == 0 ? own_keep : Ms[srow];
                } else {
                    p = ok ? __expf(sacc[r] * scale - l) : 0.f;
                }
                sacc[r] = p;
                pacc[r] = p * (pacc[r] - dd);
            }
            // gradient accumulation: contract the streamed rows
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int srow = st * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float uf = 0.f, wf = 0.f;
                    if (D >= 32 || li < D) {
                        uf = Us[srow * LD + dt * 32 + li];
                        if (SWEEP == 0) wf = Ws[srow * LD + dt * 32 + li];
                    }
                    accA[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf, pacc[r], accA[dt], 0, 0, 0);      // dK^T / dQ^T
                    if (SWEEP == 0) accB[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, sacc[r], accB[dt], 0, 0, 0);   // dV^T
                }
        }
    }
    }
    head = head0;
    if (oi < own_n) {
        // acc rows = d (4 consecutive per register quad), cols = owned row (lane)
        float* pa = SWEEP == 0 ? dk + ((long)b * tk + oi) * kv_ld + head * kv_hs : dq + ((long)b * tq + oi) * q_ld + head * q_hs;
        float* pb = SWEEP == 0 ? dv + ((long)b * tk + oi) * kv_ld + head * kv_hs : nullptr;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int dd = dt * 32 + gq * 8 + 4 * lh;
                if (dd >= D) continue;
                f32x4 va, vb2;
#pragma unroll
                for (int j = 0; j < 4; ++j) { va[j] = accA[dt][gq * 4 + j] * scale; vb2[j] = accB[dt][gq * 4 + j]; }
                *reinterpret_cast<f32x4*>(pa + dd) = va;
                if (SWEEP == 0) *reinterpret_cast<f32x4*>(pb + dd) = vb2;
            }
    }
}



// =============================================================================================
// split-precision backward (prec f16x3, round 4): the two sweeps of attention_bwd_kernel on v_mfma_f32_32x32x16_f16 with
// every fp32 operand as hi + lo f16 halves and three products, fp32 accumulate -- per 32-row tile at D = 64, 24 MFMAs
// for S^T and dP^T and 24 (q sweep) / 48 (kv sweep) for the gradient contractions, against 512 + 512 / 1024
// v_mfma_f32_32x32x2_f32 with one scalar LDS read each in the exact kernel.
//   * streamed rows U, W are staged twice: row-major planes (A operand of the score products: 8 consecutive channels of a
//     row per lane) and TRANSPOSED planes [d][row] in the row order the accumulator-as-operand idiom imposes (A operand
//     of the gradient products), exactly as attention_split_kernel stages K and V^T;
//   * P is carried times 2^14 (its lo half would sit in fp16's subnormals), undone in dV;
//   * dS = P (dP - D) has no a-priori range (it scales with the loss gradient): it is carried times a per-wave power of
//     two chosen from the running maximum of |dS| -- when a tile raises the maximum the accumulator is rescaled by the
//     (exact) ratio, as the online softmax does with its running maximum -- so the largest element sits in [2^13, 2^14)
//     and neither half overflows or underflows for elements within 2^-13 of it.
// =============================================================================================
template <int D, int SWEEP>
__global__ __launch_bounds__(256) void attention_bwd_split_kernel(
    const float* __restrict__ q, int q_ld, int q_hs, const float* __restrict__ k, const float* __restrict__ v,
    int kv_ld, int kv_hs, const float* __restrict__ dout, int dout_ld, const float* __restrict__ lse,
    const float* __restrict__ dvec, int tq, int tk, float scale, float* __restrict__ dq, float* __restrict__ dk,
    float* __restrict__ dv, int heads, int mq) {
    constexpr int DT = (D + 31) / 32;
    constexpr int DP = DT * 32;
    constexpr int KS = (D + 15) / 16;
    constexpr int TT = 32;                               // streamed rows per LDS tile
    constexpr int LDK = KS * 16 + 8;                     // f16 per row-major row
    constexpr int LDT = TT + 8;                          // f16 per transposed row
    constexpr bool KV = SWEEP == 0;
    __shared__ __attribute__((aligned(16))) _Float16 Uh[TT * LDK], Ul[TT * LDK], Wh[TT * LDK], Wl[TT * LDK];
    __shared__ __attribute__((aligned(16))) _Float16 UTh[DP * LDT], UTl[DP * LDT];
    __shared__ __attribute__((aligned(16))) _Float16 WTh[KV ? DP * LDT : 8], WTl[KV ? DP * LDT : 8];
    __shared__ float Ls[TT], Ds[TT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z;
    const int head0 = mq ? 0 : blockIdx.y, head1 = mq ? heads : blockIdx.y + 1;
    const int own_n = KV ? tk : tq, str_n = KV ? tq : tk;
    const int oi = blockIdx.x * 128 + wave * 32 + li;    // this lane's owned row (key or query)
    const int oc = oi < own_n ? oi : own_n - 1;
    constexpr float PS = 16384.0f;                       // 2^14

    f32x16 accA[DT], accB[DT];        // kv: dK^T (x sc_run), dV^T (x 2^14)   q: dQ^T (x sc_run)
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[i][r] = 0.f; accB[i][r] = 0.f; }
    float sc_run = 1.329227996e36f;                      // 2^120: any tile lowers it
    if (DP > D) {                                        // rows of the transposed images no channel writes (D = 16)
        for (int idx = tid; idx < (DP - D) * LDT; idx += 256) {
            UTh[D * LDT + idx] = (_Float16)0.f; UTl[D * LDT + idx] = (_Float16)0.f;
            if (KV) { WTh[D * LDT + idx] = (_Float16)0.f; WTl[D * LDT + idx] = (_Float16)0.f; }
        }
    }
    for (int head = head0; head < head1; ++head) {
        const float* qb = q + (long)b * tq * q_ld + head * q_hs;
        const float* kb = k + (long)b * tk * kv_ld + head * kv_hs;
        const float* vb = v + (long)b * tk * kv_ld + head * kv_hs;
        const float* gb = dout + (long)b * tq * dout_ld + head * D;
        const float* lseb = lse + ((long)b * heads + head) * tq;
        const float* dvb = dvec + ((long)b * heads + head) * tq;
        // owner fragments (B operands): k step s, lane half lh: channels 16 s + 8 lh .. + 7 of owned row li
        f16x8 xh[KS], xl[KS], yh[KS], yl[KS];
        {
            const float* xp = KV ? kb + (long)oc * kv_ld : qb + (long)oc * q_ld;
            const float* yp = KV ? vb + (long)oc * kv_ld : gb + (long)oc * dout_ld;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j4 = 0; j4 < 2; ++j4) {
                    const int c = s * 16 + lh * 8 + j4 * 4;
                    f32x4 tx = {0.f, 0.f, 0.f, 0.f}, ty = tx;
                    if (c < D) { tx = *reinterpret_cast<const f32x4*>(xp + c); ty = *reinterpret_cast<const f32x4*>(yp + c); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        _Float16 h, l;
                        split_f16(tx[j], h, l); xh[s][j4 * 4 + j] = h; xl[s][j4 * 4 + j] = l;
                        split_f16(ty[j], h, l); yh[s][j4 * 4 + j] = h; yl[s][j4 * 4 + j] = l;
                    }
                }
        }
        float own_lse = 0.f, own_d = 0.f;
        if (!KV) { own_lse = lseb[oc]; own_d = dvb[oc]; }

        // a thread stages two neighbouring rows (neighbours in the permuted order too) x 4 channels of U and of W; the rows
        // of tile t + 1 are requested before the MFMA phase of tile t and split into LDS after it
        constexpr int VPR = D / 4;
        constexpr int NI = ((TT / 2) * VPR + 255) / 256;
        f32x4 u0[NI], u1[NI], w0[NI], w1[NI];
        const long last = str_n - 1;
        auto request = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int idx = tid + i * 256;
                idx = idx < (TT / 2) * VPR ? idx : (TT / 2) * VPR - 1;
                const int row = (idx / VPR) * 2, c4 = idx % VPR;
                const long r0 = t0 + row < str_n ? t0 + row : last, r1 = t0 + row + 1 < str_n ? t0 + row + 1 : last;
                if (KV) {
                    u0[i] = *reinterpret_cast<const f32x4*>(qb + r0 * q_ld + c4 * 4);
                    u1[i] = *reinterpret_cast<const f32x4*>(qb + r1 * q_ld + c4 * 4);
                    w0[i] = *reinterpret_cast<const f32x4*>(gb + r0 * dout_ld + c4 * 4);
                    w1[i] = *reinterpret_cast<const f32x4*>(gb + r1 * dout_ld + c4 * 4);
                } else {
                    u0[i] = *reinterpret_cast<const f32x4*>(kb + r0 * kv_ld + c4 * 4);
                    u1[i] = *reinterpret_cast<const f32x4*>(kb + r1 * kv_ld + c4 * 4);
                    w0[i] = *reinterpret_cast<const f32x4*>(vb + r0 * kv_ld + c4 * 4);
                    w1[i] = *reinterpret_cast<const f32x4*>(vb + r1 * kv_ld + c4 * 4);
                }
            }
        };
        request(0);
        for (int t0 = 0; t0 < str_n; t0 += TT) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int idx = tid + i * 256;
                if (idx >= (TT / 2) * VPR) break;
                const int row = (idx / VPR) * 2, c4 = idx % VPR;
                f32x4 a0 = u0[i], a1 = u1[i], b0 = w0[i], b1 = w1[i];
                if (t0 + row >= str_n) { a0 = f32x4{0.f, 0.f, 0.f, 0.f}; b0 = a0; }
                if (t0 + row + 1 >= str_n) { a1 = f32x4{0.f, 0.f, 0.f, 0.f}; b1 = a1; }
                const int kk = row & 15;
                const int pos = (row & ~15) + (((kk >> 2) & 1) << 3) + ((kk >> 3) << 2) + (kk & 3);
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                f16x4 ah0, al0, ah1, al1, bh0, bl0, bh1, bl1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 h, l;
                    split_f16(a0[j], h, l); ah0[j] = h; al0[j] = l;
                    split_f16(a1[j], h, l); ah1[j] = h; al1[j] = l;
                    split_f16(b0[j], h, l); bh0[j] = h; bl0[j] = l;
                    split_f16(b1[j], h, l); bh1[j] = h; bl1[j] = l;
                    // (row is even: pos is even, the pair is one aligned 4-byte store)
                    *reinterpret_cast<f16x2*>(UTh + (c4 * 4 + j) * LDT + pos) = f16x2{ah0[j], ah1[j]};
                    *reinterpret_cast<f16x2*>(UTl + (c4 * 4 + j) * LDT + pos) = f16x2{al0[j], al1[j]};
                    if (KV) {
                        *reinterpret_cast<f16x2*>(WTh + (c4 * 4 + j) * LDT + pos) = f16x2{bh0[j], bh1[j]};
                        *reinterpret_cast<f16x2*>(WTl + (c4 * 4 + j) * LDT + pos) = f16x2{bl0[j], bl1[j]};
                    }
                }
                *reinterpret_cast<f16x4*>(Uh + row * LDK + c4 * 4) = ah0;
                *reinterpret_cast<f16x4*>(Ul + row * LDK + c4 * 4) = al0;
                *reinterpret_cast<f16x4*>(Uh + (row + 1) * LDK + c4 * 4) = ah1;
                *reinterpret_cast<f16x4*>(Ul + (row + 1) * LDK + c4 * 4) = al1;
                *reinterpret_cast<f16x4*>(Wh + row * LDK + c4 * 4) = bh0;
                *reinterpret_cast<f16x4*>(Wl + row * LDK + c4 * 4) = bl0;
                *reinterpret_cast<f16x4*>(Wh + (row + 1) * LDK + c4 * 4) = bh1;
                *reinterpret_cast<f16x4*>(Wl + (row + 1) * LDK + c4 * 4) = bl1;
            }
            if (KV && tid < TT) {
                const int r = t0 + tid;
                Ls[tid] = r < str_n ? lseb[r] : 0.f;
                Ds[tid] = r < str_n ? dvb[r] : 0.f;
            }
            if (t0 + TT < str_n) request(t0 + TT);               // in flight under this tile's MFMA phase
            __syncthreads();

            // scores / dP tiles: rows = streamed rows (registers), cols = owned rows (lanes)
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f16x8 uh = *reinterpret_cast<const f16x8*>(Uh + li * LDK + s * 16 + lh * 8);
                const f16x8 ul = *reinterpret_cast<const f16x8*>(Ul + li * LDK + s * 16 + lh * 8);
                const f16x8 wh = *reinterpret_cast<const f16x8*>(Wh + li * LDK + s * 16 + lh * 8);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(Wl + li * LDK + s * 16 + lh * 8);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, xh[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, xl[s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, xh[s], sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, yh[s], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yl[s], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, yh[s], pacc, 0, 0, 0);
            }
            // P (x 2^14) and dS (sacc <- P', pacc <- dS), and the wave's largest |dS|
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const bool ok = (t0 + srow < str_n) && (oi < own_n);
                const float l = KV ? Ls[srow] : own_lse;
                const float dd = KV ? Ds[srow] : own_d;
                const float p = ok ? __expf(sacc[r] * scale - l) : 0.f;
                sacc[r] = p * PS;
                pacc[r] = p * (pacc[r] - dd);
                mx = fmaxf(mx, fabsf(pacc[r]));
            }
            mx = wave_max(mx);
            // power of two that puts the running maximum into [2^13, 2^14); exponent clamped so tiny gradients stay finite
            {
                int e = (int)((__float_as_uint(mx) >> 23) & 255u) - 126;          // mx = f 2^e, f in [0.5, 1)
                int se = 14 - e;
                se = se > 100 ? 100 : (se < -100 ? -100 : se);
                const float sc_new = __uint_as_float((uint32_t)(se + 127) << 23);
                if (mx > 0.f && sc_new < sc_run) {                                // wave-uniform
                    const float ratio = sc_new / sc_run;                          // exact: powers of two (0 on the first tile)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accA[dt][r] *= ratio;
                    sc_run = sc_new;
                }
            }
            const float sc = sc_run < 1.0e36f ? sc_run : 0.f;                     // no non-zero dS seen yet: dS is all zero
            f16x8 ph[2], pl[2], dh[2], dl[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    _Float16 h, l;
                    split_f16(sacc[s2 * 8 + j], h, l); ph[s2][j] = h; pl[s2][j] = l;
                    split_f16(pacc[s2 * 8 + j] * sc, h, l); dh[s2][j] = h; dl[s2][j] = l;
                }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const f16x8 th = *reinterpret_cast<const f16x8*>(UTh + (dt * 32 + li) * LDT + s2 * 16 + lh * 8);
                    const f16x8 tl = *reinterpret_cast<const f16x8*>(UTl + (dt * 32 + li) * LDT + s2 * 16 + lh * 8);
                    accA[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl, dh[s2], accA[dt], 0, 0, 0);
                    accA[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, dl[s2], accA[dt], 0, 0, 0);
                    accA[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, dh[s2], accA[dt], 0, 0, 0);
                    if (KV) {
                        const f16x8 gh = *reinterpret_cast<const f16x8*>(WTh + (dt * 32 + li) * LDT + s2 * 16 + lh * 8);
                        const f16x8 gl = *reinterpret_cast<const f16x8*>(WTl + (dt * 32 + li) * LDT + s2 * 16 + lh * 8);
                        accB[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, ph[s2], accB[dt], 0, 0, 0);
                        accB[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, pl[s2], accB[dt], 0, 0, 0);
                        accB[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, ph[s2], accB[dt], 0, 0, 0);
                    }
                }
        }
    }
    if (oi < own_n) {
        const int head = head0;
        const float fa = sc_run < 1.0e36f ? scale / sc_run : 0.f, fb = 1.0f / PS;
        float* pa = KV ? dk + ((long)b * tk + oi) * kv_ld + head * kv_hs : dq + ((long)b * tq + oi) * q_ld + head * q_hs;
        float* pb = KV ? dv + ((long)b * tk + oi) * kv_ld + head * kv_hs : nullptr;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int dd = dt * 32 + gq * 8 + 4 * lh;
                if (dd >= D) continue;
                f32x4 va, vb2;
#pragma unroll
                for (int j = 0; j < 4; ++j) { va[j] = accA[dt][gq * 4 + j] * fa; vb2[j] = accB[dt][gq * 4 + j] * fb; }
                *reinterpret_cast<f32x4*>(pa + dd) = va;
                if (KV) *reinterpret_cast<f32x4*>(pb + dd) = vb2;
            }
    }
}


// ---------------------------------------------------------------------------------------------
// Linear attention core of attention_ldm.LinearCrossAttention (dynamic/attention_ldm.py:261-298): per (batch, head)
//   q~ = softmax_d(q) * scale        k~ = softmax over the KEYS of k (column-wise)        out = q~ (k~^T v)
// with masked keys taking k = -FLT_MAX (weight exactly 0 after the softmax) and v = 0.  A handful of context tokens x a
// 16..128-wide head: one block per (batch, head) keeps k~^T v (d x d) in LDS; pure fp32 FMA (the reference runs these
// einsums in fp32), nothing here is MFMA-shaped.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_attention_kernel(const float* __restrict__ q, int q_ld, int q_hs,
                                                               const float* __restrict__ k, const float* __restrict__ v,
                                                               int kv_ld, int kv_hs, int tq, int tk, int d, float scale,
                                                               const uint8_t* __restrict__ kmask,
                                                               float* __restrict__ out, int out_ld) {
    extern __shared__ float sh[];                    // ctx[d][d + 1] | cmax[d] | csum[d]
    float* ctx = sh;
    float* cmax = sh + d * (d + 1);
    float* csum = cmax + d;
    const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* kb = k + (long)b * tk * kv_ld + head * kv_hs;
    const float* vb = v + (long)b * tk * kv_ld + head * kv_hs;
    const uint8_t* mb = kmask ? kmask + (long)b * tk : nullptr;
    // column softmax statistics of k over the keys
    for (int c = tid; c < d; c += 256) {
        float m = -INFINITY;
        for (int j = 0; j < tk; ++j) m = fmaxf(m, (mb && !mb[j]) ? -3.402823466e38f : kb[(long)j * kv_ld + c]);
        float sum = 0.f;
        for (int j = 0; j < tk; ++j) sum += __expf(((mb && !mb[j]) ? -3.402823466e38f : kb[(long)j * kv_ld + c]) - m);
        cmax[c] = m;
        csum[c] = sum;
    }
    __syncthreads();
    // ctx[c][e] = sum_j k~[j][c] v[j][e]
    for (int i = tid; i < d * d; i += 256) {
        const int c = i / d, e = i % d;
        float acc = 0.f;
        for (int j = 0; j < tk; ++j) {
            const bool ok = !mb || mb[j];
            const float kk = ok ? kb[(long)j * kv_ld + c] : -3.402823466e38f;
            const float vv = ok ? vb[(long)j * kv_ld + e] : 0.f;
            acc += __expf(kk - cmax[c]) / csum[c] * vv;
        }
        ctx[c * (d + 1) + e] = acc;
    }
    __syncthreads();
    // one query row per thread: feature softmax, then the d x d product
    for (int i = tid; i < tq; i += 256) {
        const float* qp = q + ((long)b * tq + i) * q_ld + head * q_hs;
        float m = -INFINITY;
        for (int c = 0; c < d; ++c) m = fmaxf(m, qp[c]);
        float sum = 0.f;
        for (int c = 0; c < d; ++c) sum += __expf(qp[c] - m);
        const float inv = scale / sum;
        float* op = out + ((long)b * tq + i) * out_ld + head * q_hs;
        for (int e = 0; e < d; ++e) {
            float acc = 0.f;
            for (int c = 0; c < d; ++c) acc += __expf(qp[c] - m) * inv * ctx[c * (d + 1) + e];
            op[e] = acc;
        }
    }
}

// Backward of linear_attention_kernel (autograd of attention_ldm.py:283-296).  With s = softmax_d(q), K~ = the key softmax of k (0 at
// masked keys), v' = v (0 at masked keys), ctx = K~^T v' and out = scale s ctx:
//   dctx[c][e] = scale sum_i s[i][c] g[i][e]                                  t[i][c] = scale sum_e g[i][e] ctx[c][e]
//   dq[i][c]   = s[i][c] (t[i][c] - sum_c' s[i][c'] t[i][c'])                 dv[j][e] = sum_c K~[j][c] dctx[c][e]
//   dk[j][c]   = K~[j][c] (sum_e dctx[c][e] v'[j][e] - sum_e dctx[c][e] ctx[c][e])
// (the column term is sum_j K~[j][c] dK~[j][c] with ctx = K~^T v' substituted); masked keys receive no gradient, as
// masked_fill gives them.  One block per (batch, head), the query rows in tiles of `tr`; ctx and dctx (d x d) stay in LDS.
// Fixed summation order, no atomics.
__global__ __launch_bounds__(256) void linear_attention_bwd_kernel(const float* __restrict__ q, int q_ld, int q_hs,
                                                                   const float* __restrict__ k, const float* __restrict__ v,
                                                                   int kv_ld, int kv_hs, int tq, int tk, int d, float scale,
                                                                   const uint8_t* __restrict__ kmask,
                                                                   const float* __restrict__ dout, int dout_ld,
                                                                   float* __restrict__ dq, float* __restrict__ dk,
                                                                   float* __restrict__ dv, int tr) {
    extern __shared__ float sh[];
    const int P = d + 1;
    float* ctx = sh;                   // [d][d + 1]
    float* dctx = ctx + d * P;         // [d][d + 1]
    float* cmax = dctx + d * P;        // [d] key-softmax statistics of the columns of k
    float* csum = cmax + d;
    float* colt = csum + d;            // [d] sum_e dctx[c][e] ctx[c][e]
    float* S = colt + d;               // [tr][d] softmax_d(q) of the tile's rows
    float* G = S + tr * d;             // [tr][d] their output gradients
    float* T = G + tr * d;             // [tr][d] scale g ctx^T
    float* rmax = T + tr * d;          // [tr]
    float* rinv = rmax + tr;           // [tr]
    float* rdot = rinv + tr;           // [tr]
    const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* kb = k + (long)b * tk * kv_ld + head * kv_hs;
    const float* vb = v + (long)b * tk * kv_ld + head * kv_hs;
    const uint8_t* mb = kmask ? kmask + (long)b * tk : nullptr;
    for (int c = tid; c < d; c += 256) {
        float m = -INFINITY;
        for (int j = 0; j < tk; ++j) m = fmaxf(m, (mb && !mb[j]) ? -3.402823466e38f : kb[(long)j * kv_ld + c]);
        float sum = 0.f;
        for (int j = 0; j < tk; ++j) sum += __expf(((mb && !mb[j]) ? -3.402823466e38f : kb[(long)j * kv_ld + c]) - m);
        cmax[c] = m;
        csum[c] = sum;
    }
    __syncthreads();
    for (int i = tid; i < d * d; i += 256) {
        const int c = i / d, e = i % d;
        float acc = 0.f;
        for (int j = 0; j < tk; ++j) {
            const bool ok = !mb || mb[j];
            const float kk = ok ? kb[(long)j * kv_ld + c] : -3.402823466e38f;
            const float vv = ok ? vb[(long)j * kv_ld + e] : 0.f;
            acc += __expf(kk - cmax[c]) / csum[c] * vv;
        }
        ctx[c * P + e] = acc;
        dctx[c * P + e] = 0.f;
    }
    __syncthreads();
    for (int i0 = 0; i0 < tq; i0 += tr) {
        const int rows = tq - i0 < tr ? tq - i0 : tr;
        if (tid < rows) {
            const float* qp = q + ((long)b * tq + i0 + tid) * q_ld + head * q_hs;
            float m = -INFINITY;
            for (int c = 0; c < d; ++c) m = fmaxf(m, qp[c]);
            float sum = 0.f;
            for (int c = 0; c < d; ++c) sum += __expf(qp[c] - m);
            rmax[tid] = m;
            rinv[tid] = 1.f / sum;
        }
        __syncthreads();
        for (int idx = tid; idx < rows * d; idx += 256) {
            const int r = idx / d, c = idx % d;
            const long row = (long)b * tq + i0 + r;
            S[idx] = __expf(q[row * q_ld + head * q_hs + c] - rmax[r]) * rinv[r];
            G[idx] = dout[row * dout_ld + head * q_hs + c];
        }
        __syncthreads();
        for (int i = tid; i < d * d; i += 256) {
            const int c = i / d, e = i % d;
            float acc = 0.f;
            for (int r = 0; r < rows; ++r) acc += S[r * d + c] * G[r * d + e];
            dctx[c * P + e] += scale * acc;
        }
        for (int idx = tid; idx < rows * d; idx += 256) {
            const int r = idx / d, c = idx % d;
            float acc = 0.f;
            for (int e = 0; e < d; ++e) acc += G[r * d + e] * ctx[c * P + e];
            T[idx] = scale * acc;
        }
        __syncthreads();
        if (tid < rows) {
            float acc = 0.f;
            for (int c = 0; c < d; ++c) acc += S[tid * d + c] * T[tid * d + c];
            rdot[tid] = acc;
        }
        __syncthreads();
        for (int idx = tid; idx < rows * d; idx += 256) {
            const int r = idx / d, c = idx % d;
            dq[((long)b * tq + i0 + r) * q_ld + head * q_hs + c] = S[idx] * (T[idx] - rdot[r]);
        }
        __syncthreads();
    }
    for (int c = tid; c < d; c += 256) {
        float acc = 0.f;
        for (int e = 0; e < d; ++e) acc += dctx[c * P + e] * ctx[c * P + e];
        colt[c] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < tk * d; idx += 256) {
        const int j = idx / d, c = idx % d;
        const bool ok = !mb || mb[j];
        float gk = 0.f, gv = 0.f;
        if (ok) {
            const float kt = __expf(kb[(long)j * kv_ld + c] - cmax[c]) / csum[c];
            float acc = 0.f;
            for (int e = 0; e < d; ++e) acc += dctx[c * P + e] * vb[(long)j * kv_ld + e];
            gk = kt * (acc - colt[c]);
            for (int c2 = 0; c2 < d; ++c2) gv += __expf(kb[(long)j * kv_ld + c2] - cmax[c2]) / csum[c2] * dctx[c2 * P + c];
        }
        dk[((long)b * tk + j) * kv_ld + head * kv_hs + c] = gk;
        dv[((long)b * tk + j) * kv_ld + head * kv_hs + c] = gv;
    }
}


}  // namespace

extern "C" int sgd_attention(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                             int32_t kv_ld, int32_t kv_hs, int32_t batch, int32_t heads, int32_t tq, int32_t tk,
                             int32_t d, float scale, float* out, int32_t out_ld, float* lse, void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !out || batch <= 0 || heads <= 0 || tq <= 0 || tk <= 0) return SGD_ERR_ARG;
    if ((q_ld & 3) || (q_hs & 3) || (kv_ld & 3) || (kv_hs & 3)) return SGD_ERR_ARG;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return SGD_ERR_ARG;
    dim3 grid((tq + 127) / 128, heads, batch);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 16: hipLaunchKernelGGL((attention_kernel<16>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse); break;
        case 32: hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse); break;
        case 64: hipLaunchKernelGGL((attention_kernel<64>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse); break;
        case 128: hipLaunchKernelGGL((attention_kernel<128>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse); break;
        default: return SGD_ERR_ARG;
    }
    return sgd_check_launch();
}

extern "C" int sgd_attention_split(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                                   int32_t kv_ld, int32_t kv_hs, int32_t batch, int32_t heads, int32_t tq, int32_t tk,
                                   int32_t d, float scale, float* out, int32_t out_ld, float* lse, void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !out || batch <= 0 || heads <= 0 || tq <= 0 || tk <= 0) return SGD_ERR_ARG;
    if ((q_ld & 3) || (q_hs & 3) || (kv_ld & 3) || (kv_hs & 3) || (out_ld & 3)) return SGD_ERR_ARG;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) return SGD_ERR_ARG;
    dim3 grid((tq + 127) / 128, heads, batch);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 16: hipLaunchKernelGGL((attention_split_kernel<16>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse); break;
        case 32: hipLaunchKernelGGL((attention_split_kernel<32>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse); break;
        case 64: hipLaunchKernelGGL((attention_split_kernel<64>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse); break;
        case 128: hipLaunchKernelGGL((attention_split_kernel<128>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse); break;
        default: return SGD_ERR_ARG;
    }
    return sgd_check_launch();
}

extern "C" int sgd_attention_bwd(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                                 int32_t kv_ld, int32_t kv_hs, const float* o, int32_t o_ld, const float* dout,
                                 int32_t dout_ld, const float* lse, float* dvec, int32_t batch, int32_t heads,
                                 int32_t tq, int32_t tk, int32_t d, float scale, float* dq, float* dk, float* dv,
                                 void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !o || !dout || !lse || !dvec || !dq || !dk || !dv || batch <= 0 || heads <= 0 || tq <= 0 ||
        tk <= 0)
        return SGD_ERR_ARG;
    if ((q_ld & 3) || (q_hs & 3) || (kv_ld & 3) || (kv_hs & 3) || (o_ld & 3) || (dout_ld & 3)) return SGD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int mq = (kv_hs == 0 && heads > 1) ? 1 : 0;
    dim3 gp((heads * tq + 255) / 256, batch), gkv((tk + 127) / 128, mq ? 1 : heads, batch), gq((tq + 127) / 128, heads, batch);
#define SGD_ATTN_BWD(DD)                                                                                              \
    hipLaunchKernelGGL((attn_bwd_prep_kernel<DD>), gp, dim3(256), 0, st, o, o_ld, dout, dout_ld, heads, tq, dvec);      \
    hipLaunchKernelGGL((attention_bwd_kernel<DD, 0>), gkv, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, dout,   \
                       dout_ld, lse, dvec, tq, tk, scale, dq, dk, dv, heads, mq);                                      \
    hipLaunchKernelGGL((attention_bwd_kernel<DD, 1>), gq, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, dout,    \
                       dout_ld, lse, dvec, tq, tk, scale, dq, dk, dv, heads, 0);
    switch (d) {
        case 16: SGD_ATTN_BWD(16) break;
        case 32: SGD_ATTN_BWD(32) break;
        case 64: SGD_ATTN_BWD(64) break;
        case 128: SGD_ATTN_BWD(128) break;
        default: return SGD_ERR_ARG;
    }
#undef SGD_ATTN_BWD
    return sgd_check_launch();
}

// the same contract as sgd_attention_bwd in split precision (the f16x3 engine; head dims 16 / 32 / 64)
extern "C" int sgd_attention_bwd_split(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                                       int32_t kv_ld, int32_t kv_hs, const float* o, int32_t o_ld, const float* dout,
                                       int32_t dout_ld, const float* lse, float* dvec, int32_t batch, int32_t heads,
                                       int32_t tq, int32_t tk, int32_t d, float scale, float* dq, float* dk, float* dv,
                                       void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !o || !dout || !lse || !dvec || !dq || !dk || !dv || batch <= 0 || heads <= 0 || tq <= 0 ||
        tk <= 0)
        return SGD_ERR_ARG;
    if ((q_ld & 3) || (q_hs & 3) || (kv_ld & 3) || (kv_hs & 3) || (o_ld & 3) || (dout_ld & 3)) return SGD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int mq = (kv_hs == 0 && heads > 1) ? 1 : 0;
    dim3 gp((heads * tq + 255) / 256, batch), gkv((tk + 127) / 128, mq ? 1 : heads, batch), gq((tq + 127) / 128, heads, batch);
#define SGD_ATTN_BWD_S(DD)                                                                                            \
    hipLaunchKernelGGL((attn_bwd_prep_kernel<DD>), gp, dim3(256), 0, st, o, o_ld, dout, dout_ld, heads, tq, dvec);      \
    hipLaunchKernelGGL((attention_bwd_split_kernel<DD, 0>), gkv, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs,   \
                       dout, dout_ld, lse, dvec, tq, tk, scale, dq, dk, dv, heads, mq);                                \
    hipLaunchKernelGGL((attention_bwd_split_kernel<DD, 1>), gq, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs,    \
                       dout, dout_ld, lse, dvec, tq, tk, scale, dq, dk, dv, heads, 0);
    switch (d) {
        case 16: SGD_ATTN_BWD_S(16) break;
        case 32: SGD_ATTN_BWD_S(32) break;
        case 64: SGD_ATTN_BWD_S(64) break;
        default: return SGD_ERR_ARG;
    }
#undef SGD_ATTN_BWD_S
    return sgd_check_launch();
}

// attention with a per-key validity mask (exact fp32 core): kmask [batch, tk] bytes, 1 = attend; key 0 must be valid
extern "C" int sgd_attention_masked(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                                    int32_t kv_ld, int32_t kv_hs, const uint8_t* kmask, int32_t batch, int32_t heads,
                                    int32_t tq, int32_t tk, int32_t d, float scale, float* out, int32_t out_ld, float* lse,
                                    void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !out || batch <= 0 || heads <= 0 || tq <= 0 || tk <= 0) return SGD_ERR_ARG;
    if ((q_ld & 3) || (q_hs & 3) || (kv_ld & 3) || (kv_hs & 3)) return SGD_ERR_ARG;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return SGD_ERR_ARG;
    dim3 grid((tq + 127) / 128, heads, batch);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 16: hipLaunchKernelGGL((attention_kernel<16>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse, kmask); break;
        case 32: hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse, kmask); break;
        case 64: hipLaunchKernelGGL((attention_kernel<64>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse, kmask); break;
        case 128: hipLaunchKernelGGL((attention_kernel<128>), grid, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs, tq, tk, scale, out, out_ld, lse, kmask); break;
        default: return SGD_ERR_ARG;
    }
    return sgd_check_launch();
}

extern "C" int sgd_linear_attention(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                                    int32_t kv_ld, int32_t kv_hs, const uint8_t* kmask, int32_t batch, int32_t heads,
                                    int32_t tq, int32_t tk, int32_t d, float scale, float* out, int32_t out_ld,
                                    void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !out || batch <= 0 || heads <= 0 || tq <= 0 || tk <= 0 || d <= 0 || d > 128) return SGD_ERR_ARG;
    const size_t smem = ((size_t)d * (d + 1) + 2 * d) * sizeof(float);
    // d = 127 / 128 need 66-67 KB of dynamic LDS, above the 64 KB a kernel gets without asking
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)linear_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(linear_attention_kernel, dim3(heads, batch), dim3(256), smem, (hipStream_t)stream, q, q_ld, q_hs, k,
                       v, kv_ld, kv_hs, tq, tk, d, scale, kmask, out, out_ld);
    return sgd_check_launch();
}

// sgd_attention_bwd's exact-fp32 kernels with the forward's key mask (sgd_attention_masked): head dims 16 / 32 / 64, one head per
// block (kv_hs may be 0 only with heads == 1)
extern "C" int sgd_attention_masked_bwd(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                                        int32_t kv_ld, int32_t kv_hs, const uint8_t* kmask, const float* o, int32_t o_ld,
                                        const float* dout, int32_t dout_ld, const float* lse, float* dvec, int32_t batch,
                                        int32_t heads, int32_t tq, int32_t tk, int32_t d, float scale, float* dq, float* dk,
                                        float* dv, void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !kmask || !o || !dout || !lse || !dvec || !dq || !dk || !dv || batch <= 0 || heads <= 0 || tq <= 0 ||
        tk <= 0)
        return SGD_ERR_ARG;
    if ((q_ld & 3) || (q_hs & 3) || (kv_ld & 3) || (kv_hs & 3) || (o_ld & 3) || (dout_ld & 3)) return SGD_ERR_ARG;
    if (kv_hs == 0 && heads > 1) return SGD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    dim3 gp((heads * tq + 255) / 256, batch), gkv((tk + 127) / 128, heads, batch), gq((tq + 127) / 128, heads, batch);
#define SGD_ATTN_BWD_M(DD)                                                                                            \
    hipLaunchKernelGGL((attn_bwd_prep_kernel<DD>), gp, dim3(256), 0, st, o, o_ld, dout, dout_ld, heads, tq, dvec);      \
    hipLaunchKernelGGL((attention_bwd_kernel<DD, 0, true>), gkv, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs,   \
                       dout, dout_ld, lse, dvec, tq, tk, scale, dq, dk, dv, heads, 0, kmask);                          \
    hipLaunchKernelGGL((attention_bwd_kernel<DD, 1, true>), gq, dim3(256), 0, st, q, q_ld, q_hs, k, v, kv_ld, kv_hs,    \
                       dout, dout_ld, lse, dvec, tq, tk, scale, dq, dk, dv, heads, 0, kmask);
    switch (d) {
        case 16: SGD_ATTN_BWD_M(16) break;
        case 32: SGD_ATTN_BWD_M(32) break;
        case 64: SGD_ATTN_BWD_M(64) break;
        case 128: SGD_ATTN_BWD_M(128) break;
        default: return SGD_ERR_ARG;
    }
#undef SGD_ATTN_BWD_M
    return sgd_check_launch();
}

extern "C" int sgd_linear_attention_bwd(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v,
                                        int32_t kv_ld, int32_t kv_hs, const uint8_t* kmask, const float* dout,
                                        int32_t dout_ld, int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d,
                                        float scale, float* dq, float* dk, float* dv, void* stream) {
    SGD_CLEAR_ERR();
    if (!q || !k || !v || !dout || !dq || !dk || !dv || batch <= 0 || heads <= 0 || tq <= 0 || tk <= 0 || d <= 0 || d > 128)
        return SGD_ERR_ARG;
    // query rows per tile: ctx + dctx take 2 d (d + 1) floats (129 KB at d = 128) of the 160 KB
    const int tr = d > 64 ? 8 : 32;
    const size_t smem = ((size_t)2 * d * (d + 1) + 3 * d + (size_t)3 * tr * d + 3 * tr) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)linear_attention_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  152 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(linear_attention_bwd_kernel, dim3(heads, batch), dim3(256), smem, (hipStream_t)stream, q, q_ld, q_hs,
                       k, v, kv_ld, kv_hs, tq, tk, d, scale, kmask, dout, dout_ld, dq, dk, dv, tr);
    return sgd_check_launch();
}
