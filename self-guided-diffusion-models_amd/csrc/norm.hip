// GroupNorm / LayerNorm statistics kernels (gfx950).  HBM-bound: each reads its input once with
// 16-byte lane loads; reductions are wave-shuffle + LDS, deterministic (no atomics).
//
// GroupNorm(32, C) (util.py:199-216) is split into
//   sgd_chan_stats : per-(n, channel) sum / sum-of-squares over the HW rows (NHWC => coalesced),
//   sgd_gn_coef    : folds group statistics, gamma/beta and the ResBlock FiLM scale/shift
//                    (openaimodel.py:312-316) into per-(n, c) coefficients a, b with
//                    GN(x)*(1+scale)+shift == a*x + b, consumed by sgd_igemm's prologue.
// Per-channel sums make the grouping independent of where the channels live, so a GroupNorm over a
// virtual concat (openaimodel.py:950 -> :246) is two sgd_chan_stats calls into one sums buffer.
#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"

namespace {

// block: 256 threads = 8 channel-quads x 32 row lanes; one block per (n, 32-channel slab)
__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, int hw, int c,
                                                         float* __restrict__ sums, int c_total, int c_off) {
    const int slabs = (c + 31) / 32;
    const int n = blockIdx.x / slabs, slab = blockIdx.x % slabs;
    const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int ch = slab * 32 + q * 4;
    float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    const float* base = x + (long)n * hw * c;
    if (ch < c) {
        if ((c & 3) == 0) {
            for (int r = rl; r < hw; r += 32) {
                f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)r * c + ch);
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[j] += v[j]; ss[j] += v[j] * v[j]; }
            }
        } else {
            for (int r = rl; r < hw; r += 32)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ch + j < c) { float v = base[(long)r * c + ch + j]; s[j] += v; ss[j] += v * v; }
        }
    }
    __shared__ double red[32][8][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[rl][q][j] = s[j]; red[rl][q][4 + j] = ss[j]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int qq = threadIdx.x >> 3, jj = threadIdx.x & 7;
        double t = 0;
        for (int r = 0; r < 32; ++r) t += red[r][qq][jj];
        const int cc = slab * 32 + qq * 4 + (jj & 3);
        if (cc < c) sums[((long)n * c_total + c_off + cc) * 2 + (jj >> 2)] = (float)t;
    }
}

// partial[n][parts][2][c] -> sums[n][c_total][2] at channel offset c_off.  8 part lanes x 32 (k, channel) columns per
// block keep the loads independent (a single thread walking the parts pays one memory latency per part);
// fixed-order double accumulation.
__global__ __launch_bounds__(256) void stats_reduce_kernel(const float* __restrict__ partial, int parts, int c,
                                                           float* __restrict__ sums, int c_total, int c_off) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;      // i = (k, channel)
    double t = 0;
    if (i < 2 * c) {
        const float* p = partial + (long)n * parts * 2 * c + i;
#pragma unroll 4
        for (int q = pl; q < parts; q += 8) t += p[(long)q * 2 * c];
    }
    __shared__ double red[8][32];
    red[pl][threadIdx.x & 31] = t;
    __syncthreads();
    if (threadIdx.x < 32 && i < 2 * c) {
        double u = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) u += red[q][threadIdx.x];
        const int k = i / c, ch = i - k * c;
        sums[((long)n * c_total + c_off + ch) * 2 + k] = (float)u;
    }
}

__global__ void gn_coef_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ film, int film_ld,
                               int n, int c, int groups, int hw, float eps, float* __restrict__ a,
                               float* __restrict__ b) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)n * c) return;
    const int nn = i / c, cc = i % c;
    const int cpg = c / groups, g0 = (cc / cpg) * cpg;
    double s = 0, ss = 0;
    for (int k = 0; k < cpg; ++k) {
        s += sums[((long)nn * c + g0 + k) * 2];
        ss += sums[((long)nn * c + g0 + k) * 2 + 1];
    }
    const double cnt = (double)cpg * hw;
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    double ga = gamma[cc] * rstd, be = beta[cc] - mean * ga;
    if (film) {
        const double sc = 1.0 + film[(long)nn * film_ld + cc], sh = film[(long)nn * film_ld + c + cc];
        ga *= sc;
        be = be * sc + sh;
    }
    a[i] = (float)ga;
    b[i] = (float)be;
}

// sgd_stats_reduce (for up to two concatenated sources) + sgd_gn_coef in ONE launch: one block per image.
// Phase 1 folds the producers' partial statistics into sums[n, c, 2] (kept: the training backward reads them; a source
// with parts == 0 already has its sums there, written by sgd_chan_stats); phase 2 is gn_coef_kernel's arithmetic on the
// float-rounded sums.  The partials are folded in DOUBLE in strided groups, sgd_stats_reduce folds them in index order: the
// two routes agree wherever the double sums round to the same float -- in practice always, but the contract is the
// tolerance of tests/test_hip_kernels.py::test_groupnorm_coefficients_from_partial_statistics (1e-6), not bit identity.
__global__ __launch_bounds__(512) void gn_coef_parts_kernel(const float* __restrict__ p0, int parts0, int c0,
                                                            const float* __restrict__ p1, int parts1, int c1,
                                                            float* __restrict__ sums, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ film,
                                                            int film_ld, int groups, int hw, float eps,
                                                            float* __restrict__ a, float* __restrict__ b) {
    extern __shared__ double shd[];                   // [G][c][2] partial sums of the part groups, then float sh[c][2]
    const int nn = blockIdx.x, c = c0 + c1, nthr = blockDim.x;
    // G threads per channel, each folding every G-th partial pair (round 4: with one thread per channel half of the block
    // idled at 128 channels and a 64x64 map's 32 partials were four dependent round trips -- 6 us for a launch that runs
    // 49 times per UNet evaluation); fixed assignment and fixed fold order: deterministic
    const int G = c < nthr ? nthr / c : 1;
    float* sh = reinterpret_cast<float*>(shd + (size_t)G * c * 2);
    for (int it = threadIdx.x; it < c * G; it += nthr) {
        const int ch = it % c, grp = it / c;
        const bool first = ch < c0;
        const float* p = first ? p0 : p1;
        const int parts = first ? parts0 : parts1, cs = first ? c0 : c1, cl = first ? ch : ch - c0;
        double t0 = 0, t1 = 0;
        if (parts > 0) {
            const float* q = p + (long)nn * parts * 2 * cs + cl;
            // up to eight partial pairs in flight per thread, folded in index order
            int k = grp;
            for (; k + 7 * G < parts; k += 8 * G) {
                float v0[8], v1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { v0[j] = q[(long)(k + j * G) * 2 * cs]; v1[j] = q[(long)(k + j * G) * 2 * cs + cs]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) { t0 += v0[j]; t1 += v1[j]; }
            }
            if (k + 3 * G < parts) {
                float v0[4], v1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = q[(long)(k + j * G) * 2 * cs]; v1[j] = q[(long)(k + j * G) * 2 * cs + cs]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) { t0 += v0[j]; t1 += v1[j]; }
                k += 4 * G;
            }
            for (; k < parts; k += G) { t0 += q[(long)k * 2 * cs]; t1 += q[(long)k * 2 * cs + cs]; }
        }
        shd[((size_t)grp * c + ch) * 2] = t0;
        shd[((size_t)grp * c + ch) * 2 + 1] = t1;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += nthr) {
        const int parts = ch < c0 ? parts0 : parts1;
        float s, ss;
        if (parts > 0) {
            double t0 = 0, t1 = 0;
            for (int gi = 0; gi < G; ++gi) { t0 += shd[((size_t)gi * c + ch) * 2]; t1 += shd[((size_t)gi * c + ch) * 2 + 1]; }
            s = (float)t0;
            ss = (float)t1;
            sums[((long)nn * c + ch) * 2] = s;
            sums[((long)nn * c + ch) * 2 + 1] = ss;
        } else {
            s = sums[((long)nn * c + ch) * 2];
            ss = sums[((long)nn * c + ch) * 2 + 1];
        }
        sh[ch * 2] = s;                               // (sh sits behind the partials: no overlay)
        sh[ch * 2 + 1] = ss;
    }
    __syncthreads();
    const int cpg = c / groups;
    for (int cc = threadIdx.x; cc < c; cc += blockDim.x) {
        const int g0 = (cc / cpg) * cpg;
        double s = 0, ss = 0;
        for (int k = 0; k < cpg; ++k) { s += sh[(g0 + k) * 2]; ss += sh[(g0 + k) * 2 + 1]; }
        const double cnt = (double)cpg * hw;
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0) var = 0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        double ga = gamma[cc] * rstd, be = beta[cc] - mean * ga;
        if (film) {
            const double sc = 1.0 + film[(long)nn * film_ld + cc], shf = film[(long)nn * film_ld + c + cc];
            ga *= sc;
            be = be * sc + shf;
        }
        a[(long)nn * c + cc] = (float)ga;
        b[(long)nn * c + cc] = (float)be;
    }
}

// one wave per row, row held in registers (c <= 1024)
template <bool APPLY>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ res,
                                                 int rows, int c, float eps, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xp = x + (long)row * c;
    float v[16];
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int cc = lane + k * 64;
        v[k] = cc < c ? xp[cc] : 0.f;
        s += v[k];
    }
    const float mean = wave_sum(s) / c;
    float q = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int cc = lane + k * 64;
        float d = cc < c ? v[k] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / c + eps);
    if (!APPLY) {
        if (lane == 0) { out[(long)row * 2] = mean; out[(long)row * 2 + 1] = rstd; }
        return;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int cc = lane + k * 64;
        if (cc < c) {
            float y = (v[k] - mean) * rstd * gamma[cc] + (beta ? beta[cc] : 0.f);
            if (res) y += res[(long)row * c + cc];
            out[(long)row * c + cc] = y;
        }
    }
}


// LayerNorm backward, one wave per row (row in registers, c <= 1024):
//   y = xhat*gamma + beta ;  dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat))
// dx is written (or accumulated) into dst; gxhat[row, c] = g * xhat is emitted for the dgamma column sum.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ gamma, int rows, int c, float eps,
                                                     float* __restrict__ dst, int accumulate,
                                                     float* __restrict__ gxhat, const float* __restrict__ gres) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xp = x + (long)row * c;
    const float* gp = g + (long)row * c;
    float v[16], gg[16];
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int cc = lane + k * 64;
        v[k] = cc < c ? xp[cc] : 0.f;
        gg[k] = cc < c ? gp[cc] : 0.f;
        s += v[k];
    }
    const float mean = wave_sum(s) / c;
    float q = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int cc = lane + k * 64;
        float d = cc < c ? v[k] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / c + eps);
    float m1 = 0, m2 = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int cc = lane + k * 64;
        if (cc < c) {
            const float xh = (v[k] - mean) * rstd, gw = gg[k] * gamma[cc];
            v[k] = xh;
            m1 += gw;
            m2 += gw * xh;
        }
    }
    m1 = wave_sum(m1) / c;
    m2 = wave_sum(m2) / c;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int cc = lane + k * 64;
        if (cc < c) {
            float dx = rstd * (gg[k] * gamma[cc] - m1 - v[k] * m2);
            if (gres) dx += gres[(long)row * c + cc];             // gradient of a residual branch around the norm
            float* dp = dst + (long)row * c + cc;
            *dp = accumulate ? *dp + dx : dx;
            if (gxhat) gxhat[(long)row * c + cc] = gg[k] * v[k];
        }
    }
}

// ResBlock WITHOUT scale-shift norm (openaimodel.py:317-319: h = h + emb_out[..., None, None] in front of out_layers' GroupNorm):
// x[n, px, c] += e[n, c], in place, 16-byte accesses (c, e_ld multiples of 4).  The shipped plans all use the FiLM form, which
// folds the embedding into the GroupNorm coefficients instead (sgd_gn_coef); this form keeps the statistics pass of its own.
__global__ __launch_bounds__(256) void add_rows_nc_kernel(float* __restrict__ x, const float* __restrict__ e, int e_ld, long hw,
                                                          int c4, long total) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % c4);
        const long img = (i / c4) / hw;
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
        v += *reinterpret_cast<const f32x4*>(e + img * e_ld + q * 4);
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

}  // namespace

extern "C" int sgd_chan_stats(const float* x, int32_t n, int32_t hw, int32_t c, float* sums, int32_t c_total,
                              int32_t c_off, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !sums || n <= 0 || hw <= 0 || c <= 0 || c_off < 0 || c_off + c > c_total) return SGD_ERR_ARG;
    const int slabs = (c + 31) / 32;
    hipLaunchKernelGGL(chan_stats_kernel, dim3(n * slabs), dim3(256), 0, (hipStream_t)stream, x, hw, c, sums,
                       c_total, c_off);
    return sgd_check_launch();
}

extern "C" int sgd_add_rows_nc(float* x, const float* e, int32_t e_ld, int32_t n, int64_t hw, int32_t c, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !e || n <= 0 || hw <= 0 || c <= 0 || (c & 3) || (e_ld & 3) || e_ld < c) return SGD_ERR_ARG;
    if ((((uintptr_t)x) | ((uintptr_t)e)) & 15) return SGD_ERR_ARG;
    const long total = (long)n * hw * (c / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add_rows_nc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, e, e_ld, (long)hw, c / 4,
                       total);
    return sgd_check_launch();
}

extern "C" int sgd_stats_reduce(const float* partial, int32_t n, int32_t parts, int32_t c, float* sums,
                                int32_t c_total, int32_t c_off, void* stream) {
    SGD_CLEAR_ERR();
    if (!partial || !sums || n <= 0 || parts <= 0 || c <= 0 || c_off < 0 || c_off + c > c_total) return SGD_ERR_ARG;
    hipLaunchKernelGGL(stats_reduce_kernel, dim3((2 * c + 31) / 32, n), dim3(256), 0, (hipStream_t)stream, partial,
                       parts, c, sums, c_total, c_off);
    return sgd_check_launch();
}

extern "C" int sgd_gn_coef(const float* sums, const float* gamma, const float* beta, const float* film,
                           int32_t film_ld, int32_t n, int32_t c, int32_t groups, int32_t hw, float eps, float* a,
                           float* b, void* stream) {
    SGD_CLEAR_ERR();
    if (!sums || !gamma || !beta || !a || !b || n <= 0 || c <= 0 || groups <= 0 || c % groups != 0 || hw <= 0)
        return SGD_ERR_ARG;
    if (film && film_ld < 2 * c) return SGD_ERR_ARG;
    const long total = (long)n * c;
    hipLaunchKernelGGL(gn_coef_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       sums, gamma, beta, film, film_ld, n, c, groups, hw, eps, a, b);
    return sgd_check_launch();
}

extern "C" int sgd_gn_coef_parts(const float* p0, int32_t parts0, int32_t c0, const float* p1, int32_t parts1, int32_t c1,
                                 float* sums, const float* gamma, const float* beta, const float* film, int32_t film_ld,
                                 int32_t n, int32_t groups, int32_t hw, float eps, float* a, float* b, void* stream) {
    SGD_CLEAR_ERR();
    const int c = c0 + c1;
    if (!sums || !gamma || !beta || !a || !b || n <= 0 || c0 <= 0 || c1 < 0 || groups <= 0 || c % groups != 0 || hw <= 0 ||
        parts0 < 0 || parts1 < 0 || (parts0 > 0 && !p0) || (c1 > 0 && parts1 > 0 && !p1) || c > 8192)
        return SGD_ERR_ARG;
    if (film && film_ld < 2 * c) return SGD_ERR_ARG;
    const int nthr = 512, G = c < nthr ? nthr / c : 1;
    // dynamic LDS: G partial double pairs per channel + the folded float pair: 24 bytes per channel from 512 channels on.
    // Above the 64 KB default (c > 2730) the function attribute is raised once; what does not fit a CU's 160 KB is refused
    // here instead of failing at launch (the guard used to be the 8-bytes-per-channel figure of the round-3 kernel)
    const size_t lds = (size_t)G * c * 2 * sizeof(double) + (size_t)c * 2 * sizeof(float);
    if (lds > 160 * 1024) return SGD_ERR_ARG;
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            (void)hipFuncSetAttribute((const void*)gn_coef_parts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = true;
        }
    }
    hipLaunchKernelGGL(gn_coef_parts_kernel, dim3(n), dim3(nthr), lds,
                       (hipStream_t)stream, p0, parts0, c0, p1, parts1, c1, sums, gamma, beta, film, film_ld, groups, hw, eps, a, b);
    return sgd_check_launch();
}

extern "C" int sgd_ln_stats(const float* x, int32_t rows, int32_t c, float eps, float* stats, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !stats || rows <= 0 || c <= 0 || c > 1024) return SGD_ERR_ARG;
    hipLaunchKernelGGL((ln_kernel<false>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, rows, c, eps, stats);
    return sgd_check_launch();
}

extern "C" int sgd_ln_apply(const float* x, const float* gamma, const float* beta, const float* res, int32_t rows,
                            int32_t c, float eps, float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !gamma || !out || rows <= 0 || c <= 0 || c > 1024) return SGD_ERR_ARG;
    hipLaunchKernelGGL((ln_kernel<true>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                       res, rows, c, eps, out);
    return sgd_check_launch();
}

extern "C" int sgd_ln_bwd(const float* x, const float* g, const float* gamma, int32_t rows, int32_t c, float eps,
                          float* dst, int32_t accumulate, float* gxhat, const float* gres, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !g || !gamma || !dst || rows <= 0 || c <= 0 || c > 1024) return SGD_ERR_ARG;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, g, gamma, rows, c, eps,
                       dst, accumulate, gxhat, gres);
    return sgd_check_launch();
}
