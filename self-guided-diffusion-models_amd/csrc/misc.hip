// Small boundary / sampler kernels (gfx950).  All element-wise, HBM- or launch-bound.
#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"

namespace {

// timestep_embedding (util.py:151-171): cat(cos(t*f), sin(t*f)); the frequency table f_j =
// exp(-ln(1e4) * j / half) is deterministic host math (built once with the reference's own fp32 op
// order) so that t*f is bit-identical to the reference -- a 1-ulp change of f moves cos(999 f) by 1e-5.
__global__ void temb_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, int n_src, int n,
                            int dim, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    const int r = i / dim, j = i % dim;
    const int half = dim / 2;
    float val = 0.f;                       // odd dim: last column is zero (util.py:167-168)
    if (j < 2 * half) {
        const int jj = j < half ? j : j - half;
        const float arg = (float)t[r % n_src] * freqs[jj];
        val = j < half ? cosf(arg) : sinf(arg);
    }
    out[i] = val;
}

__global__ void cond_select_kernel(const void* __restrict__ cond, int is_i64, const uint8_t* __restrict__ mask,
                                   const float* __restrict__ null_row, int n_src, int n, int k,
                                   float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)n * k) return;
    const int r = i / k, j = i % k;
    float v;
    if (mask && mask[r]) v = null_row[j];
    else if (is_i64 == 2) {                                   // class / cluster id -> one-hot
        // an id outside [0, k) has no one-hot row (F.one_hot raises on it): the row becomes NaN, so the failure is loud
        const int64_t id = reinterpret_cast<const int64_t*>(cond)[r % n_src];
        v = (uint64_t)id < (uint64_t)k ? (id == j ? 1.f : 0.f) : __builtin_nanf("");
    }
    else if (is_i64) v = (float)reinterpret_cast<const int64_t*>(cond)[(long)(r % n_src) * k + j];
    else v = reinterpret_cast<const float*>(cond)[(long)(r % n_src) * k + j];
    out[i] = v;
}

// NCHW x (+ masked layout) -> NHWC; thread per output pixel-channel, output-coalesced
__global__ void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ layout,
                                  const uint8_t* __restrict__ mask, const float* __restrict__ null_layout,
                                  int n_src, int n, int cx, int cl, int hw, float* __restrict__ out) {
    const int ct = cx + cl;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)n * hw * ct) return;
    const int c = i % ct;
    const long t = i / ct;
    const int p = t % hw, r = t / hw;
    const int rs = r % n_src;
    float v;
    if (c < cx) v = x[((long)rs * cx + c) * hw + p];
    else if (mask && mask[r]) v = null_layout[p];          // null_layout_emb is [1,1,H,W]: broadcast over channels
    else v = layout[((long)rs * cl + (c - cx)) * hw + p];
    out[i] = v;
}

// the same with the layout given in its compact on-disk form and expanded here (bit-exact restatement of the
// reference's CPU-side expansion, dataset/transforms/complex_ds_common_util.py:118-123 stego_to_onehotmask and :151-162
// get_lostbboxmask): fmt 1 = uint8 label map [n_src, h, w] (255 -> class 0) one-hot over cl channels,
// fmt 2 = int32 boxes [n_src, 4] = (x0, y0, x1, y1), mask[y0:y1, x0:x1] = 1 (cl == 1)
__global__ void pack_input_compact_kernel(const float* __restrict__ x, const void* __restrict__ layout, int fmt,
                                          const uint8_t* __restrict__ mask, const float* __restrict__ null_layout,
                                          int n_src, int n, int cx, int cl, int h, int w, float* __restrict__ out) {
    const int ct = cx + cl, hw = h * w;
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)n * hw * ct) return;
    const int c = i % ct;
    const long t = i / ct;
    const int p = t % hw, r = t / hw;
    const int rs = r % n_src;
    float v;
    if (c < cx) v = x[((long)rs * cx + c) * hw + p];
    else if (mask && mask[r]) v = null_layout[p];
    else if (fmt == 1) {
        int lab = reinterpret_cast<const uint8_t*>(layout)[(long)rs * hw + p];
        if (lab == 255) lab = 0;
        v = lab == c - cx ? 1.f : 0.f;
    } else {
        const int* b = reinterpret_cast<const int*>(layout) + (long)rs * 4;
        const int px = p % w, py = p / w;
        v = (px >= b[0] && px < b[2] && py >= b[1] && py < b[3]) ? 1.f : 0.f;
    }
    out[i] = v;
}

// n-hot vector of the labels present in a label map (stegomask_to_attr_nhot, complex_ds_common_util.py:126-133):
// one block per image, presence flags in LDS
__global__ void labelmap_nhot_kernel(const uint8_t* __restrict__ labels, int hw, int k, float* __restrict__ out) {
    __shared__ int present[256];
    for (int j = threadIdx.x; j < 256; j += blockDim.x) present[j] = 0;
    __syncthreads();
    const uint8_t* lp = labels + (long)blockIdx.x * hw;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) present[lp[p]] = 1;
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += blockDim.x) out[(long)blockIdx.x * k + j] = present[j] ? 1.f : 0.f;
}

// first Linear of mlp_cond on a one-hot input = one weight column per row (openaimodel.py:597-607 at cluster k = 5000:
// 2*n*k*nout flops and a k-wide input row per sample replaced by a gather).  x . w[c, :] with x one-hot at id is
// exactly w[c, id] (adding exact zeros), so out = w[c, id] + bias[c] is bit-identical to the dense product; rows whose
// cond is dropped take the precomputed projection of the null embedding.
__global__ void linear_gather_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ mask,
                                     const float* __restrict__ w, const float* __restrict__ bias,
                                     const float* __restrict__ nullproj, int n_src, int n, int nout, int k,
                                     float* __restrict__ out, int ldo) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)n * nout) return;
    const int r = i / nout, c = i % nout;
    float v;
    if (mask && mask[r]) v = nullproj[c];
    else {
        const int64_t id = ids[r % n_src];
        // no out-of-bounds read for an id outside [0, k): NaN out (the reference's F.one_hot raises on such ids)
        v = (uint64_t)id < (uint64_t)k ? w[(long)c * k + id] + (bias ? bias[c] : 0.f) : __builtin_nanf("");
    }
    out[(long)r * ldo + c] = v;
}

// y[r, :] = bias + sum over the NON-ZERO entries of an int64 condition row of value * w[:, k]: the first Linear of mlp_cond on
// the reference's own input format (unsupervised_cluster.py:33-46 hands the UNet a one-hot int64 [B, K] row, K = 5000 at C2).
// A dense GEMM over that row multiplies 4,999 zeros per sample (sgd_linear_splitk: 0.095 ms per UNet evaluation, 5 MB of weight
// read for 160 columns of it); this kernel finds the row's non-zero entries (one block per row: count per thread chunk, prefix,
// write the (index, value) list in ascending k -- deterministic) and gathers just those weight columns.  Any integer row is
// handled (multi-hot, counts, negative entries): skipped terms are exact zeros; a one-hot row gives fmaf(1, w, 0) + bias, the
// dense kernel's bits.  Dropped rows (mask) take the projection of the null embedding, as in linear_gather_kernel.
__global__ __launch_bounds__(256) void linear_sparse_rows_kernel(const int64_t* __restrict__ cond, const uint8_t* __restrict__ mask,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 const float* __restrict__ nullproj, int n_src, int nout, int k,
                                                                 float* __restrict__ out, int ldo) {
    extern __shared__ int sparse_lds[];              // idx[k] | val[k]
    __shared__ int cnt[257];
    int* idx = sparse_lds;
    float* val = reinterpret_cast<float*>(sparse_lds + k);
    const int r = blockIdx.x, t = threadIdx.x;
    if (mask && mask[r]) {
        for (int c = t; c < nout; c += 256) out[(long)r * ldo + c] = nullproj[c];
        return;
    }
    const int64_t* row = cond + (long)(r % n_src) * k;
    const int per = (k + 255) / 256, k0 = t * per, k1 = (k0 + per < k) ? k0 + per : k;
    int mine = 0;
    for (int j = k0; j < k1; ++j) mine += row[j] != 0;
    cnt[t + 1] = mine;
    if (t == 0) cnt[0] = 0;
    __syncthreads();
    if (t < 64) {                                    // inclusive scan of the 256 counts by one wave: 4 per lane + wave scan
        int a0 = cnt[4 * t + 1], a1 = a0 + cnt[4 * t + 2], a2 = a1 + cnt[4 * t + 3], a3 = a2 + cnt[4 * t + 4];
        int run = a3;
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(run, d);
            if (t >= d) run += up;
        }
        const int base = run - a3;
        cnt[4 * t + 1] = base + a0; cnt[4 * t + 2] = base + a1; cnt[4 * t + 3] = base + a2; cnt[4 * t + 4] = base + a3;
    }
    __syncthreads();
    int at = cnt[t];
    for (int j = k0; j < k1; ++j) {
        const int64_t v = row[j];
        if (v != 0) { idx[at] = j; val[at] = (float)v; ++at; }
    }
    __syncthreads();
    const int total = cnt[256];
    for (int c = t; c < nout; c += 256) {
        const float* wr = w + (long)c * k;
        float acc = 0.f;
        for (int e = 0; e < total; ++e) acc = fmaf(val[e], wr[idx[e]], acc);
        out[(long)r * ldo + c] = acc + (bias ? bias[c] : 0.f);
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int n, int hw, int c, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)n * hw * c) return;
    const int p = i % hw;
    const long t = i / hw;
    const int cc = t % c, r = t / c;
    out[i] = x[((long)r * hw + p) * c + cc];
}

__global__ void fill_null_kv_kernel(const float* __restrict__ null_kv, int batch, int rows_per_b, int row, int d,
                                    float* __restrict__ kv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * 2 * d) return;
    const int b = i / (2 * d), j = i % (2 * d);
    kv[((long)b * rows_per_b + row) * 2 * d + j] = null_kv[j];      // [k(0:d) | v(d:2d)] == null_kv[0], null_kv[1]
}

__device__ __forceinline__ float guided(const float* __restrict__ eps, int cfg_mode, float w, int b, int n, int c,
                                        int hw, int cc, int p) {
    const float ec = eps[((long)n * hw + p) * c + cc];
    if (cfg_mode == 0) return ec;
    const float eu = eps[((long)(n + b) * hw + p) * c + cc];
    if (cfg_mode == 1) return (1.f - w) * eu + w * ec;        // imagen  (openaimodel.py:855)
    return (1.f + w) * ec - w * eu;                           // cfg     (openaimodel.py:857)
}

struct Coef5 { float v[5]; };

// x / x_out carry no __restrict__: the graph-replayed form updates x in place
__global__ void ddpm_step_kernel(const float* x, const float* __restrict__ eps,
                                 const float* __restrict__ z, int cfg_mode, float w, Coef5 k, const float* kdev, int clip,
                                 int b, int c, int hw, float* x_out, float* __restrict__ x0_out,
                                 const float* __restrict__ dyn_s) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)b * c * hw) return;
    if (kdev) {
#pragma unroll
        for (int j = 0; j < 5; ++j) k.v[j] = kdev[j];
    }
    const int p = i % hw;
    const long t = i / hw;
    const int cc = t % c, n = t / c;
    const float e = guided(eps, cfg_mode, w, b, n, c, hw, cc, p);
    const float xv = x[i];
    float x0 = k.v[0] * xv - k.v[1] * e;                      // predict_start_from_noise (ddpm_sampler.py:132-137)
    if (dyn_s) {                                              // dynamic thresholding, dtp < 1 (diffusion_utils/util.py:70-79)
        const float s = dyn_s[n];
        x0 = fminf(fmaxf(x0, -s), s) / s;
    } else if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);        // clip_x0_minus_one_to_one, dtp == 1
    const float mean = k.v[2] * x0 + k.v[3] * xv;             // q_posterior (ddpm_sampler.py:121-125)
    x_out[i] = mean + k.v[4] * z[i];                          // :190-191, k4 = nonzero*exp(.5 logvar)*temperature
    if (x0_out) x0_out[i] = x0;
}

__global__ void ddim_step_kernel(const float* x, const float* __restrict__ eps,
                                 const float* __restrict__ z, int cfg_mode, float w, Coef5 k, const float* kdev,
                                 float temperature, int clip, int b, int c, int hw, float* x_out,
                                 float* __restrict__ x0_out, const float* __restrict__ dyn_s) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)b * c * hw) return;
    if (kdev) {
#pragma unroll
        for (int j = 0; j < 4; ++j) k.v[j] = kdev[j];
    }
    const int p = i % hw;
    const long t = i / hw;
    const int cc = t % c, n = t / c;
    const float e = guided(eps, cfg_mode, w, b, n, c, hw, cc, p);
    const float s1m = k.v[0], a_t = k.v[1], a_prev = k.v[2], sigma = k.v[3];
    float x0 = (x[i] - s1m * e) / sqrtf(a_t);                 // ddim_plms_sampler.py:369-370
    if (dyn_s) {
        const float s = dyn_s[n];
        x0 = fminf(fmaxf(x0, -s), s) / s;
    } else if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    const float dir = sqrtf(1.0f - a_prev - sigma * sigma) * e;   // :381-382
    const float noise = sigma * z[i] * temperature;              // :383-387
    x_out[i] = sqrtf(a_prev) * x0 + dir + noise;                 // :390
    if (x0_out) x0_out[i] = x0;
}

__global__ void to_uint8_kernel(const float* __restrict__ x, long count, uint8_t* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= count) return;
    float v = (x[i] + 1.f) * 127.5f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    out[i] = (uint8_t)v;                                       // truncation like .to(torch.uint8)
}

__global__ void cfg_combine_kernel(const float* __restrict__ eps, int cfg_mode, float w, int b, int c, int hw,
                                   float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)b * c * hw) return;
    const int p = i % hw;
    const long t = i / hw;
    const int cc = t % c, n = t / c;
    out[i] = guided(eps, cfg_mode, w, b, n, c, hw, cc, p);
}

// GEGLU gate of the LDM transformer feed-forward (reference dynamic/attention.py:38-45): out = a * gelu(g), exact (erf) GELU
__global__ void geglu_kernel(const float* __restrict__ in, long rows, int inner, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;      // one thread = 4 consecutive channels
    const int q = inner >> 2;
    if (i >= rows * q) return;
    const long r = i / q;
    const int c = (int)(i - r * q) * 4;
    const float4 a = *reinterpret_cast<const float4*>(in + r * 2 * inner + c);
    const float4 g = *reinterpret_cast<const float4*>(in + r * 2 * inner + inner + c);
    float4 o;
    o.x = a.x * (0.5f * g.x * (1.0f + erff(g.x * 0.70710678118654752f)));
    o.y = a.y * (0.5f * g.y * (1.0f + erff(g.y * 0.70710678118654752f)));
    o.z = a.z * (0.5f * g.z * (1.0f + erff(g.z * 0.70710678118654752f)));
    o.w = a.w * (0.5f * g.w * (1.0f + erff(g.w * 0.70710678118654752f)));
    *reinterpret_cast<float4*>(out + r * inner + c) = o;
}

inline unsigned nblk(long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int sgd_timestep_embedding(const int64_t* t, const float* freqs, int32_t n_src, int32_t n, int32_t dim,
                                      float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!t || !freqs || !out || n_src <= 0 || n <= 0 || dim <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(temb_kernel, dim3(nblk((long)n * dim)), dim3(256), 0, (hipStream_t)stream, t, freqs, n_src, n,
                       dim, out);
    return sgd_check_launch();
}

extern "C" int sgd_cond_select(const void* cond, int32_t is_i64, const uint8_t* mask, const float* null_row,
                               int32_t n_src, int32_t n, int32_t k, float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!cond || !out || n_src <= 0 || n <= 0 || k <= 0 || (mask && !null_row)) return SGD_ERR_ARG;
    hipLaunchKernelGGL(cond_select_kernel, dim3(nblk((long)n * k)), dim3(256), 0, (hipStream_t)stream, cond, is_i64,
                       mask, null_row, n_src, n, k, out);
    return sgd_check_launch();
}

extern "C" int sgd_pack_input(const float* x, const float* layout, const uint8_t* mask, const float* null_layout,
                              int32_t n_src, int32_t n, int32_t cx, int32_t cl, int32_t h, int32_t w, float* out,
                              void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !out || n_src <= 0 || n <= 0 || cx <= 0 || cl < 0 || h <= 0 || w <= 0) return SGD_ERR_ARG;
    if (cl > 0 && (!layout || (mask && !null_layout))) return SGD_ERR_ARG;
    hipLaunchKernelGGL(pack_input_kernel, dim3(nblk((long)n * h * w * (cx + cl))), dim3(256), 0, (hipStream_t)stream,
                       x, layout, mask, null_layout, n_src, n, cx, cl, h * w, out);
    return sgd_check_launch();
}

extern "C" int sgd_pack_input_compact(const float* x, const void* layout, int32_t layout_fmt, const uint8_t* mask,
                                      const float* null_layout, int32_t n_src, int32_t n, int32_t cx, int32_t cl, int32_t h,
                                      int32_t w, float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !out || !layout || n_src <= 0 || n <= 0 || cx <= 0 || cl <= 0 || h <= 0 || w <= 0) return SGD_ERR_ARG;
    if ((layout_fmt != 1 && layout_fmt != 2) || (layout_fmt == 2 && cl != 1) || (layout_fmt == 1 && cl > 255)) return SGD_ERR_ARG;
    if (mask && !null_layout) return SGD_ERR_ARG;
    hipLaunchKernelGGL(pack_input_compact_kernel, dim3(nblk((long)n * h * w * (cx + cl))), dim3(256), 0, (hipStream_t)stream,
                       x, layout, layout_fmt, mask, null_layout, n_src, n, cx, cl, h, w, out);
    return sgd_check_launch();
}

extern "C" int sgd_labelmap_nhot(const uint8_t* labels, int32_t b, int32_t hw, int32_t k, float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!labels || !out || b <= 0 || hw <= 0 || k <= 0 || k > 256) return SGD_ERR_ARG;
    hipLaunchKernelGGL(labelmap_nhot_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, labels, hw, k, out);
    return sgd_check_launch();
}

extern "C" int sgd_linear_gather(const int64_t* ids, const uint8_t* mask, const float* w, const float* bias,
                                 const float* nullproj, int32_t n_src, int32_t n, int32_t nout, int32_t k, float* out,
                                 int32_t ldo, void* stream) {
    SGD_CLEAR_ERR();
    if (!ids || !w || !out || n_src <= 0 || n <= 0 || nout <= 0 || k <= 0 || ldo < nout || (mask && !nullproj)) return SGD_ERR_ARG;
    hipLaunchKernelGGL(linear_gather_kernel, dim3(nblk((long)n * nout)), dim3(256), 0, (hipStream_t)stream, ids, mask, w,
                       bias, nullproj, n_src, n, nout, k, out, ldo);
    return sgd_check_launch();
}

extern "C" int sgd_linear_sparse_rows(const int64_t* cond, const uint8_t* mask, const float* w, const float* bias,
                                      const float* nullproj, int32_t n_src, int32_t n, int32_t nout, int32_t k, float* out,
                                      int32_t ldo, void* stream) {
    SGD_CLEAR_ERR();
    if (!cond || !w || !out || n_src <= 0 || n <= 0 || nout <= 0 || k <= 0 || ldo < nout || (mask && !nullproj)) return SGD_ERR_ARG;
    const size_t lds = 8 * (size_t)k;
    if (lds > 64 * 1024 - 2048) return SGD_ERR_ARG;          // (the caller keeps longer rows on the dense kernel)
    hipLaunchKernelGGL(linear_sparse_rows_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, cond, mask, w, bias, nullproj,
                       n_src, nout, k, out, ldo);
    return sgd_check_launch();
}

extern "C" int sgd_nhwc_to_nchw(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, float* out,
                                void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblk((long)n * h * w * c)), dim3(256), 0, (hipStream_t)stream, x, n,
                       h * w, c, out);
    return sgd_check_launch();
}

extern "C" int sgd_fill_null_kv(const float* null_kv, int32_t batch, int32_t rows_per_b, int32_t row, int32_t d,
                                float* kv, void* stream) {
    SGD_CLEAR_ERR();
    if (!null_kv || !kv || batch <= 0 || rows_per_b <= 0 || row < 0 || row >= rows_per_b || d <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(fill_null_kv_kernel, dim3(nblk((long)batch * 2 * d)), dim3(256), 0, (hipStream_t)stream,
                       null_kv, batch, rows_per_b, row, d, kv);
    return sgd_check_launch();
}

static int ddpm_step_impl(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                          const float* coef, int32_t clip, int32_t b, int32_t c, int32_t hw, float* x_out,
                          float* x0_out, const float* dyn_s, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !eps_nhwc || !z || !coef || !x_out || b <= 0 || c <= 0 || hw <= 0 || cfg_mode < 0 || cfg_mode > 2)
        return SGD_ERR_ARG;
    Coef5 k;
    for (int i = 0; i < 5; ++i) k.v[i] = coef[i];
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(nblk((long)b * c * hw)), dim3(256), 0, (hipStream_t)stream, x, eps_nhwc,
                       z, cfg_mode, w, k, (const float*)nullptr, clip, b, c, hw, x_out, x0_out, dyn_s);
    return sgd_check_launch();
}

extern "C" int sgd_ddpm_step_dev(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                                 const float* coef_dev, int32_t clip, int32_t b, int32_t c, int32_t hw, float* x_out,
                                 float* x0_out, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !eps_nhwc || !z || !coef_dev || !x_out || b <= 0 || c <= 0 || hw <= 0 || cfg_mode < 0 || cfg_mode > 2)
        return SGD_ERR_ARG;
    Coef5 k = {};
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(nblk((long)b * c * hw)), dim3(256), 0, (hipStream_t)stream, x, eps_nhwc,
                       z, cfg_mode, w, k, coef_dev, clip, b, c, hw, x_out, x0_out, (const float*)nullptr);
    return sgd_check_launch();
}

static int ddim_step_impl(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                          const float* coef, float temperature, int32_t clip, int32_t b, int32_t c, int32_t hw,
                          float* x_out, float* x0_out, const float* dyn_s, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !eps_nhwc || !z || !coef || !x_out || b <= 0 || c <= 0 || hw <= 0 || cfg_mode < 0 || cfg_mode > 2)
        return SGD_ERR_ARG;
    Coef5 k;
    for (int i = 0; i < 4; ++i) k.v[i] = coef[i];
    k.v[4] = 0.f;
    hipLaunchKernelGGL(ddim_step_kernel, dim3(nblk((long)b * c * hw)), dim3(256), 0, (hipStream_t)stream, x, eps_nhwc,
                       z, cfg_mode, w, k, (const float*)nullptr, temperature, clip, b, c, hw, x_out, x0_out, dyn_s);
    return sgd_check_launch();
}

extern "C" int sgd_ddim_step_dev(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                                 const float* coef_dev, float temperature, int32_t clip, int32_t b, int32_t c, int32_t hw,
                                 float* x_out, float* x0_out, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !eps_nhwc || !z || !coef_dev || !x_out || b <= 0 || c <= 0 || hw <= 0 || cfg_mode < 0 || cfg_mode > 2)
        return SGD_ERR_ARG;
    Coef5 k = {};
    hipLaunchKernelGGL(ddim_step_kernel, dim3(nblk((long)b * c * hw)), dim3(256), 0, (hipStream_t)stream, x, eps_nhwc,
                       z, cfg_mode, w, k, coef_dev, temperature, clip, b, c, hw, x_out, x0_out, (const float*)nullptr);
    return sgd_check_launch();
}

extern "C" int sgd_ddpm_step(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                             const float* coef, int32_t clip, int32_t b, int32_t c, int32_t hw, float* x_out,
                             float* x0_out, void* stream) {
    return ddpm_step_impl(x, eps_nhwc, z, cfg_mode, w, coef, clip, b, c, hw, x_out, x0_out, nullptr, stream);
}
extern "C" int sgd_ddim_step(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                             const float* coef, float temperature, int32_t clip, int32_t b, int32_t c, int32_t hw,
                             float* x_out, float* x0_out, void* stream) {
    return ddim_step_impl(x, eps_nhwc, z, cfg_mode, w, coef, temperature, clip, b, c, hw, x_out, x0_out, nullptr, stream);
}
extern "C" int sgd_ddpm_step_dyn(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                                 const float* coef, const float* dyn_s, int32_t b, int32_t c, int32_t hw, float* x_out,
                                 float* x0_out, void* stream) {
    if (!dyn_s) return SGD_ERR_ARG;
    return ddpm_step_impl(x, eps_nhwc, z, cfg_mode, w, coef, 0, b, c, hw, x_out, x0_out, dyn_s, stream);
}
extern "C" int sgd_ddim_step_dyn(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                                 const float* coef, float temperature, const float* dyn_s, int32_t b, int32_t c, int32_t hw,
                                 float* x_out, float* x0_out, void* stream) {
    if (!dyn_s) return SGD_ERR_ARG;
    return ddim_step_impl(x, eps_nhwc, z, cfg_mode, w, coef, temperature, 0, b, c, hw, x_out, x0_out, dyn_s, stream);
}

// Dynamic thresholding scale (dtp < 1; clip_x0_minus_one_to_one, diffusion_utils/util.py:70-79): per sample
//   s = max(1, quantile(|x0|, dtp)) with torch.quantile's linear interpolation between the order statistics `lo` and `hi`
//   (rank = dtp * (n - 1) in fp32, formed by the caller exactly as torch forms it; frac = rank - lo).
// One block per sample; x0 is recomputed from (x, guided eps) like the step kernels do; the k-th smallest |x0| comes from
// a 4-pass byte-wise radix select over the float bit patterns (non-negative floats order like their bits) -- exact order
// statistics, no sort, no atomics on global memory.
template <int KIND>
__global__ __launch_bounds__(256) void x0_quantile_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                          int cfg_mode, float w, Coef5 k, int b, int c, int hw, int lo,
                                                          int hi, float frac, float* __restrict__ s_out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_prefix, sel_k;
    const int n = blockIdx.x, cnt = c * hw;
    auto key_of = [&](int j) -> unsigned {
        const int cc = j / hw, p = j - cc * hw;
        const float e = guided(eps, cfg_mode, w, b, n, c, hw, cc, p);
        const float xv = x[(long)n * cnt + j];
        const float x0 = KIND == 0 ? k.v[0] * xv - k.v[1] * e : (xv - k.v[0] * e) / sqrtf(k.v[1]);
        return __float_as_uint(fabsf(x0));
    };
    float vals[2];
    for (int which = 0; which < 2; ++which) {
        if (threadIdx.x == 0) { sel_prefix = 0; sel_k = which == 0 ? lo : hi; }
        unsigned mask = 0;
        for (int pass = 3; pass >= 0; --pass) {
            for (int j = threadIdx.x; j < 256; j += blockDim.x) hist[j] = 0;
            __syncthreads();
            const unsigned prefix = sel_prefix;
            for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
                const unsigned key = key_of(j);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned kk = sel_k, cum = 0;
                int bkt = 0;
                for (; bkt < 256; ++bkt) {
                    if (cum + hist[bkt] > kk) break;
                    cum += hist[bkt];
                }
                sel_k = kk - cum;
                sel_prefix = prefix | ((unsigned)bkt << (8 * pass));
            }
            mask |= 0xFFu << (8 * pass);
            __syncthreads();
        }
        vals[which] = __uint_as_float(sel_prefix);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // torch.lerp(v_lo, v_hi, frac)
        const float d = vals[1] - vals[0];
        const float q = frac < 0.5f ? vals[0] + frac * d : vals[1] - d * (1.0f - frac);
        s_out[n] = fmaxf(q, 1.0f);                            // s.clamp_(min=1.0): only takes effect if s > 1
    }
}

extern "C" int sgd_x0_quantile(int32_t kind, const float* x, const float* eps_nhwc, int32_t cfg_mode, float w,
                               const float* coef, int32_t b, int32_t c, int32_t hw, int32_t lo, int32_t hi, float frac,
                               float* s_out, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !eps_nhwc || !coef || !s_out || b <= 0 || c <= 0 || hw <= 0 || cfg_mode < 0 || cfg_mode > 2) return SGD_ERR_ARG;
    if (lo < 0 || hi < lo || hi >= c * hw || (kind != 0 && kind != 1)) return SGD_ERR_ARG;
    Coef5 k = {};
    for (int i = 0; i < (kind == 0 ? 5 : 4); ++i) k.v[i] = coef[i];
    if (kind == 0) hipLaunchKernelGGL((x0_quantile_kernel<0>), dim3(b), dim3(256), 0, (hipStream_t)stream, x, eps_nhwc, cfg_mode, w, k, b, c, hw, lo, hi, frac, s_out);
    else hipLaunchKernelGGL((x0_quantile_kernel<1>), dim3(b), dim3(256), 0, (hipStream_t)stream, x, eps_nhwc, cfg_mode, w, k, b, c, hw, lo, hi, frac, s_out);
    return sgd_check_launch();
}

// pooled guidance token (openaimodel_ca.py:999-1004): cond [n, T, c] -> out [n, c]; cls = 1: token 0, else the mean over T
__global__ void token_pool_kernel(const float* __restrict__ cond, int n, int T, int c, int cls, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    const int r = i / c, j = i % c;
    const float* p = cond + (long)r * T * c + j;
    float v = p[0];
    if (!cls) {
        for (int t = 1; t < T; ++t) v += p[(long)t * c];
        v /= (float)T;
    }
    out[i] = v;
}

extern "C" int sgd_token_pool(const float* cond, int32_t n, int32_t tokens, int32_t c, int32_t cls, float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!cond || !out || n <= 0 || tokens <= 0 || c <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(token_pool_kernel, dim3(nblk((long)n * c)), dim3(256), 0, (hipStream_t)stream, cond, n, tokens, c, cls, out);
    return sgd_check_launch();
}

extern "C" int sgd_geglu(const float* in, int64_t rows, int32_t inner, float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!in || !out || rows <= 0 || inner <= 0 || (inner & 3)) return SGD_ERR_ARG;
    hipLaunchKernelGGL(geglu_kernel, dim3(nblk(rows * (inner >> 2))), dim3(256), 0, (hipStream_t)stream, in, (long)rows,
                       inner, out);
    return sgd_check_launch();
}

extern "C" int sgd_to_uint8(const float* x, int64_t count, uint8_t* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !out || count <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(to_uint8_kernel, dim3(nblk(count)), dim3(256), 0, (hipStream_t)stream, x, (long)count, out);
    return sgd_check_launch();
}

extern "C" int sgd_cfg_combine(const float* eps_nhwc, int32_t cfg_mode, float w, int32_t b, int32_t c, int32_t hw,
                               float* out_nchw, void* stream) {
    SGD_CLEAR_ERR();
    if (!eps_nhwc || !out_nchw || b <= 0 || c <= 0 || hw <= 0 || cfg_mode < 0 || cfg_mode > 2) return SGD_ERR_ARG;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(nblk((long)b * c * hw)), dim3(256), 0, (hipStream_t)stream, eps_nhwc,
                       cfg_mode, w, b, c, hw, out_nchw);
    return sgd_check_launch();
}

// ---------------------------------------------------------------------------------------------
// fused AdamW + LitEma step over a table of tensors (multi-tensor apply): HBM-bound, 36 bytes per element
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int OPT_CHUNK = 4096;

__global__ __launch_bounds__(256) void adamw_ema_kernel(const sgd_opt_tensor* __restrict__ table,
                                                        const int32_t* __restrict__ chunk_start, int count, float lr,
                                                        float omb1, float b2, float omb2, float eps, float wd,
                                                        float bc1, float bc2, float omd, const float* __restrict__ skip) {
    // the step is void when the gradients are: a balanced-tail time-out of the backward program poisons its tiles with NaN
    // and raises the health flag the caller hands in (all-reduced over the ranks of a data-parallel job) -- parameters,
    // moments and EMA shadows stay as they are and the host raises at its next look at the flag (unet._Engine.poll_health)
    if (skip && *skip != 0.f) return;
    // the tensor that owns this chunk: last t with chunk_start[t] <= blockIdx.x
    int lo = 0, hi = count - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (chunk_start[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const sgd_opt_tensor t = table[lo];
    const long base = (long)(b - chunk_start[lo]) * OPT_CHUNK;
    const float step_size = lr / bc1, rs2 = 1.0f / sqrtf(bc2), decay_f = 1.0f - lr * wd;
    for (int i = threadIdx.x; i < OPT_CHUNK; i += 256) {
        const long e = base + i;
        if (e >= t.n) break;
        float p = t.p[e];
        if (t.g) {
            const float g = t.g[e];
            float m = t.m[e], v = t.v[e];
            p *= decay_f;
            m = m + omb1 * (g - m);
            v = b2 * v + omb2 * g * g;
            p -= step_size * (m / (sqrtf(v) * rs2 + eps));
            t.m[e] = m;
            t.v[e] = v;
            t.p[e] = p;
        }
        if (omd >= 0.f && t.ema) {
            const float s = t.ema[e];
            t.ema[e] = s - omd * (s - p);
        }
    }
}
}  // namespace

extern "C" int sgd_adamw_ema_step(const sgd_opt_tensor* table, const int32_t* chunk_start, int32_t count,
                                  int32_t total_chunks, float lr, float one_minus_beta1, float beta2,
                                  float one_minus_beta2, float eps, float weight_decay, float bias_correction1,
                                  float bias_correction2, float ema_one_minus_decay, const float* skip_if_nonzero,
                                  void* stream) {
    SGD_CLEAR_ERR();
    if (!table || !chunk_start || count <= 0 || total_chunks <= 0 || bias_correction1 <= 0.f || bias_correction2 <= 0.f)
        return SGD_ERR_ARG;
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_start,
                       count, lr, one_minus_beta1, beta2, one_minus_beta2, eps, weight_decay, bias_correction1, bias_correction2,
                       ema_one_minus_decay, skip_if_nonzero);
    return sgd_check_launch();
}

// ---------------------------------------------------------------------------------------------
// skinny split-K linear (few rows, thousands of input features): the tiled conv kernel would walk the whole K range
// with 4 blocks; here (n / 64) x ksplit blocks each own a K slice
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int LS_MMAX = 256;     // rows handled by one launch (8 per thread x 4 thread rows x up to 8 row tiles)

__global__ __launch_bounds__(256) void linear_splitk_kernel(const float* __restrict__ x, int x_ld,
                                                            const float* __restrict__ w, int m, int n, int k,
                                                            int ksplit, float* __restrict__ work) {
    __shared__ float ws[64][33];
    __shared__ float xs[32][33];
    const int tn = threadIdx.x & 63, tm = threadIdx.x >> 6;       // column of the tile, row group (8 rows of a 32-row tile)
    const int n0 = blockIdx.x * 64, ks = blockIdx.y;
    const int kper = ((k + ksplit - 1) / ksplit + 31) / 32 * 32;
    const int kb = ks * kper, ke = (kb + kper < k) ? kb + kper : k;
    float acc[LS_MMAX / 32][8];
#pragma unroll
    for (int i = 0; i < LS_MMAX / 32; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int k0 = kb; k0 < ke; k0 += 32) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 32; i += 256) {
            const int r = i >> 5, c = i & 31;
            ws[r][c] = (n0 + r < n && k0 + c < ke) ? w[(long)(n0 + r) * k + k0 + c] : 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < LS_MMAX / 32; ++mt) {
            if (mt * 32 >= m) break;
            __syncthreads();
            for (int i = threadIdx.x; i < 32 * 32; i += 256) {
                const int r = i >> 5, c = i & 31;
                xs[r][c] = (mt * 32 + r < m && k0 + c < ke) ? x[(long)(mt * 32 + r) * x_ld + k0 + c] : 0.f;
            }
            __syncthreads();
#pragma unroll 8
            for (int c = 0; c < 32; ++c) {
                const float wv = ws[tn][c];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[mt][j] = fmaf(xs[tm * 8 + j][c], wv, acc[mt][j]);
            }
        }
    }
    if (n0 + tn < n) {
#pragma unroll
        for (int mt = 0; mt < LS_MMAX / 32; ++mt)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = mt * 32 + tm * 8 + j;
                if (row < m) work[((long)ks * m + row) * n + n0 + tn] = acc[mt][j];
            }
    }
}

// the same with the weight indexed [k][n] (row stride w_ld): y[m, n] = sum_k x[m, k] w[k][n] -- the input gradient of a wide
// linear layer seen from its few rows, e.g. all ResBlocks' emb_layers as one GEMM (openaimodel.py:262-268): 80 rows x
// 13,824 output features back to 768 embedding channels ran as 4 tiles of the conv kernel walking 432 K steps each
// (0.75 ms per training step for 1.7 GFLOP)
__global__ __launch_bounds__(256) void linear_splitk_t_kernel(const float* __restrict__ x, int x_ld,
                                                              const float* __restrict__ w, int w_ld, int m, int n, int k,
                                                              int ksplit, float* __restrict__ work) {
    __shared__ float ws[64][33];
    __shared__ float xs[32][33];
    const int tn = threadIdx.x & 63, tm = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 64, ks = blockIdx.y;
    const int kper = ((k + ksplit - 1) / ksplit + 31) / 32 * 32;
    const int kb = ks * kper, ke = (kb + kper < k) ? kb + kper : k;
    float acc[LS_MMAX / 32][8];
#pragma unroll
    for (int i = 0; i < LS_MMAX / 32; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int k0 = kb; k0 < ke; k0 += 32) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 32; i += 256) {
            const int r = i & 63, c = i >> 6;                     // consecutive threads walk a weight row: coalesced
            ws[r][c] = (n0 + r < n && k0 + c < ke) ? w[(long)(k0 + c) * w_ld + n0 + r] : 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < LS_MMAX / 32; ++mt) {
            if (mt * 32 >= m) break;
            __syncthreads();
            for (int i = threadIdx.x; i < 32 * 32; i += 256) {
                const int r = i >> 5, c = i & 31;
                xs[r][c] = (mt * 32 + r < m && k0 + c < ke) ? x[(long)(mt * 32 + r) * x_ld + k0 + c] : 0.f;
            }
            __syncthreads();
#pragma unroll 8
            for (int c = 0; c < 32; ++c) {
                const float wv = ws[tn][c];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[mt][j] = fmaf(xs[tm * 8 + j][c], wv, acc[mt][j]);
            }
        }
    }
    if (n0 + tn < n) {
#pragma unroll
        for (int mt = 0; mt < LS_MMAX / 32; ++mt)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = mt * 32 + tm * 8 + j;
                if (row < m) work[((long)ks * m + row) * n + n0 + tn] = acc[mt][j];
            }
    }
}

__global__ void linear_splitk_fold_kernel(const float* __restrict__ work, const float* __restrict__ bias, int m, int n,
                                          int ksplit, float* __restrict__ y, int y_ld) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)m * n) return;
    const int row = (int)(i / n), col = (int)(i - (long)row * n);
    float s = 0.f;
    for (int q = 0; q < ksplit; ++q) s += work[(long)q * m * n + i];
    y[(long)row * y_ld + col] = s + (bias ? bias[col] : 0.f);
}
}  // namespace

extern "C" int sgd_linear_splitk(const float* x, int32_t x_ld, const float* w, const float* bias, int32_t m, int32_t n,
                                 int32_t k, float* work, int32_t ksplit, float* y, int32_t y_ld, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !w || !work || !y || m <= 0 || m > LS_MMAX || n <= 0 || k <= 0 || ksplit <= 0 || x_ld < k || y_ld < n)
        return SGD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(linear_splitk_kernel, dim3((n + 63) / 64, ksplit), dim3(256), 0, st, x, x_ld, w, m, n, k, ksplit, work);
    hipLaunchKernelGGL(linear_splitk_fold_kernel, dim3((unsigned)(((long)m * n + 255) / 256)), dim3(256), 0, st, work, bias,
                       m, n, ksplit, y, y_ld);
    return sgd_check_launch();
}

extern "C" int sgd_linear_splitk_t(const float* x, int32_t x_ld, const float* w, int32_t w_ld, int32_t m, int32_t n, int32_t k,
                                   float* work, int32_t ksplit, float* y, int32_t y_ld, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !w || !work || !y || m <= 0 || m > LS_MMAX || n <= 0 || k <= 0 || ksplit <= 0 || x_ld < k || y_ld < n || w_ld < n)
        return SGD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(linear_splitk_t_kernel, dim3((n + 63) / 64, ksplit), dim3(256), 0, st, x, x_ld, w, w_ld, m, n, k, ksplit, work);
    hipLaunchKernelGGL(linear_splitk_fold_kernel, dim3((unsigned)(((long)m * n + 255) / 256)), dim3(256), 0, st, work,
                       (const float*)nullptr, m, n, ksplit, y, y_ld);
    return sgd_check_launch();
}
