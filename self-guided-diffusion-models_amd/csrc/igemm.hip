// Fused implicit-GEMM convolution / linear kernel for gfx950 (MI355X).
//
//   M = output rows (pixels of NHWC maps, or flat rows), N = Cout, K = taps * Cin.
//   Block: 256 threads = 4 waves, tile 128 rows x BN cols, K chunk = 32 input channels.
//   CONV3: the (activated) input halo tile of the 128 output pixels is staged ONCE per channel
//          chunk in LDS and re-read by the 9 taps; the tap's weight slice [BN x 32] is staged per
//          tap, register-prefetched one tap ahead (double-buffered LDS).
//   Prologue (GroupNorm apply + SiLU, LayerNorm, avg-pool / nearest-upsample, channel concat)
//   is applied in the global->LDS loader; epilogue adds bias and the residual.
//
//   Arithmetic: PREC_F32 uses v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
//               PREC_F16X3 / BF16X3 split every operand x = hi + lo into two 16-bit floats and
//               accumulate hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_{f16,bf16} in fp32
//               (BASELINE.md section 2 "precision headroom": 4.3e-6 / 2.6e-5 rel. error per UNet eval).
//
// Replaces (reference): nn.Conv2d/Conv1d/Linear call sites listed in include/sgdm_hip.h.
#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"

namespace {

constexpr int KC = 32;        // input channels per K chunk
constexpr int LDA = KC + 4;   // LDS row stride in floats (144 B): conflict-free b128 fragment reads
constexpr int BM = 128;

struct Geo {
    int tw_l2, th_l2;         // log2 of the spatial tile (CONV3)
    int hh, hw;               // halo tile dims (rows, cols) in conv-input pixels
    int nb;                   // images per M tile
    int tiles_x, tiles_y;     // spatial tiles per image
    int pix;                  // nb*hh*hw (CONV3) or 128 (FLAT)
    int mt, nt;               // number of M / N tiles
    int hc, wc;               // conv-input dims (after resample)
};

struct KArgs {
    sgd_igemm_args a;
    Geo g;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// one activated input vector (4 consecutive channels starting at c) of conv-input pixel (n, y, x)
// or flat row `row`; all masking (bounds / padding) is done by the caller.
template <bool VEC>
__device__ __forceinline__ f32x4 load_raw(const sgd_igemm_args& a, long row_idx, int c) {
    // row_idx indexes rows of the source tensors (both have the same row count)
    f32x4 v;
    if (VEC) {
        if (c < a.c0) v = ld4(a.x0 + row_idx * a.c0 + c);
        else v = ld4(a.x1 + row_idx * a.c1 + (c - a.c0));
    } else {
        const int ct = a.c0 + a.c1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int cc = c + j;
            float s = 0.f;
            if (cc < ct) s = (cc < a.c0) ? a.x0[row_idx * a.c0 + cc] : a.x1[row_idx * a.c1 + (cc - a.c0)];
            v[j] = s;
        }
    }
    return v;
}

template <bool VEC>
__device__ __forceinline__ f32x4 prologue(const sgd_igemm_args& a, f32x4 v, int n, long row, int c) {
    const int ct = a.c0 + a.c1;
    if (a.pro == SGD_PRO_AFFINE_NC) {
        f32x4 pa, pb;
        if (VEC) {
            pa = ld4(a.pa + (long)n * ct + c);
            pb = ld4(a.pb + (long)n * ct + c);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bool ok = c + j < ct;
                pa[j] = ok ? a.pa[(long)n * ct + c + j] : 0.f;
                pb[j] = ok ? a.pb[(long)n * ct + c + j] : 0.f;
            }
        }
        v = v * pa + pb;
    } else if (a.pro == SGD_PRO_LN_ROW) {
        const float mean = a.pa[row * 2], rstd = a.pa[row * 2 + 1];
        f32x4 g, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = c + j < ct;
            g[j] = ok ? a.pb[c + j] : 0.f;
            b[j] = (ok && a.pc) ? a.pc[c + j] : 0.f;
        }
        v = (v - mean) * rstd * g + b;
    }
    if (a.pro_silu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sgd_silu(v[j]);
    }
    return v;
}

// activated conv-input value at (n, y, x) [conv-input resolution], channels c..c+3
template <bool VEC>
__device__ __forceinline__ f32x4 load_act_conv(const sgd_igemm_args& a, int n, int y, int x, int c) {
    if (a.resample == SGD_RS_AVGPOOL2) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                long r = ((long)n * a.hi + (2 * y + dy)) * a.wi + (2 * x + dx);
                acc += prologue<VEC>(a, load_raw<VEC>(a, r, c), n, r, c);
            }
        return acc * 0.25f;
    }
    long r;
    if (a.resample == SGD_RS_UP2) r = ((long)n * a.hi + (y >> 1)) * a.wi + (x >> 1);
    else r = ((long)n * a.hi + y) * a.wi + x;
    return prologue<VEC>(a, load_raw<VEC>(a, r, c), n, r, c);
}

// ---------------------------------------------------------------------------------------------
// LDS element packing per precision
// ---------------------------------------------------------------------------------------------
// F32   : row = 32 floats (channel order) + 4 pad.
// split : row = 4 groups of 8 channels; group = hi[8] (16 B) | lo[8] (16 B); + 16 B pad. Same 144 B.
template <int PREC> struct Split;
template <> struct Split<SGD_PREC_F16X3> {
    typedef _Float16 T;
    static __device__ __forceinline__ T hi(float v) { return (T)v; }
    static __device__ __forceinline__ float back(T h) { return (float)h; }
};
template <> struct Split<SGD_PREC_BF16X3> {
    typedef __bf16 T;
    static __device__ __forceinline__ T hi(float v) { return (T)v; }
    static __device__ __forceinline__ float back(T h) { return (float)h; }
};

template <int PREC>
__device__ __forceinline__ void lds_store_act(float* rowp, int c4, f32x4 v) {
    if constexpr (PREC == SGD_PREC_F32) {
        *reinterpret_cast<f32x4*>(rowp + c4 * 4) = v;
    } else {
        typedef typename Split<PREC>::T T;
        // channels c4*4 .. +3 -> group g = c4 >> 1, position (c4 & 1) * 4 inside the 8-group
        T* base = reinterpret_cast<T*>(rowp) + (c4 >> 1) * 16 + (c4 & 1) * 4;
        T h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = Split<PREC>::hi(v[j]);
            l[j] = Split<PREC>::hi(v[j] - Split<PREC>::back(h[j]));
        }
        typedef T T4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<T4*>(base) = T4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<T4*>(base + 8) = T4{l[0], l[1], l[2], l[3]};
    }
}

// ---------------------------------------------------------------------------------------------
// the kernel
//   BN = 128: waves 2(M) x 2(N), each 64 x 64 (2 x 2 MFMA tiles of 32x32)
//   BN = 32 : waves 4(M) x 1(N), each 32 x 32
// ---------------------------------------------------------------------------------------------
template <int BN, int PREC, bool VEC>
__global__ __launch_bounds__(256) void igemm_kernel(const KArgs ka) {
    const sgd_igemm_args& a = ka.a;
    const Geo& g = ka.g;
    constexpr int WM = (BN == 128) ? 64 : 32;     // wave tile rows
    constexpr int WN = (BN == 128) ? 64 : 32;     // wave tile cols
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int WAVES_N = BN / WN;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [pix][LDA]
    float* Bs = smem + (size_t)g.pix * LDA;        // [2][BN][LDA]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a
    // contiguous run of tiles with the N tiles of one M tile adjacent (they share the input tile).
    const int total = g.mt * g.nt;
    const int chunk = (total + 7) >> 3;
    const int lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (lin >= total) return;
    const int mtile = lin / g.nt, ntile = lin - mtile * g.nt;
    const int n0c = ntile * BN;

    const bool conv = a.mode == SGD_MODE_CONV3;
    const int taps = conv ? 9 : 1;
    const int cin = a.c0 + a.c1;
    const int nchunks = (cin + KC - 1) / KC;
    const int s = conv ? a.stride : 1;

    // tile origin
    int img0 = 0, ty0 = 0, tx0 = 0;      // CONV3
    long m0 = 0;                          // FLAT
    const int TW = 1 << g.tw_l2, TH = 1 << g.th_l2;
    if (conv) {
        int per_img = g.tiles_x * g.tiles_y;
        int it = mtile / per_img, rem = mtile - it * per_img;
        img0 = it * g.nb;
        ty0 = (rem / g.tiles_x) * TH;
        tx0 = (rem % g.tiles_x) * TW;
    } else {
        m0 = (long)mtile * BM;
    }
    const int M = conv ? a.n * a.ho * a.wo : a.m;

    // per-lane LDS pixel bases of this wave's MFMA row tiles (row -> halo pixel for tap (0,0))
    int pixbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int r = wm * WM + mt * 32 + li;
        if (conv) {
            int tx = r & (TW - 1), ty = (r >> g.tw_l2) & (TH - 1), nb = r >> (g.tw_l2 + g.th_l2);
            pixbase[mt] = (nb * g.hh + ty * s) * g.hw + tx * s;
        } else {
            pixbase[mt] = r;
        }
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- stage helpers -------------------------------------------------------------------
    auto stage_A = [&](int kc0) {
        const int items = g.pix * 8;
        for (int idx = tid; idx < items; idx += 256) {
            const int pix = idx >> 3, c4 = idx & 7;
            const int c = kc0 + c4 * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < cin) {
                if (conv) {
                    int hx = pix % g.hw, t = pix / g.hw;
                    int hy = t % g.hh, nb = t / g.hh;
                    int n = img0 + nb, y = ty0 * s - 1 + hy, x = tx0 * s - 1 + hx;
                    if (n < a.n && y >= 0 && y < g.hc && x >= 0 && x < g.wc) v = load_act_conv<VEC>(a, n, y, x, c);
                } else {
                    long row = m0 + pix;
                    if (row < M) {
                        int n = (a.pro == SGD_PRO_AFFINE_NC) ? (int)(row / a.rows_per_n) : 0;
                        v = prologue<VEC>(a, load_raw<VEC>(a, row, c), n, row, c);
                    }
                }
            }
            lds_store_act<PREC>(As + (size_t)pix * LDA, c4, v);
        }
    };
    // weights: packed rows of KC-chunk granularity: [tap][cout_p][cin_p] with 4-byte elements
    // (f32, or a (hi,lo) 16-bit pair pre-arranged in the same 8-group layout as the LDS rows).
    constexpr int BITEMS = BN * 8 / 256;           // float4 per thread per tap slice (4 or 1)
    f32x4 breg[BITEMS];
    auto load_B = [&](int tap, int kc0) {
        const float* wp = reinterpret_cast<const float*>(a.w) + ((size_t)tap * a.cout_p + n0c) * a.cin_p + kc0;
#pragma unroll
        for (int it = 0; it < BITEMS; ++it) {
            int idx = tid + it * 256;
            int row = idx >> 3, c4 = idx & 7;
            breg[it] = ld4(wp + (size_t)row * a.cin_p + c4 * 4);
        }
    };
    auto store_B = [&](int buf) {
        float* bp = Bs + (size_t)buf * BN * LDA;
#pragma unroll
        for (int it = 0; it < BITEMS; ++it) {
            int idx = tid + it * 256;
            int row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<f32x4*>(bp + row * LDA + c4 * 4) = breg[it];
        }
    };
    auto compute = [&](int tap, int buf) {
        const int tapoff = conv ? ((tap / 3) * g.hw + (tap % 3)) : 0;
        const float* bp = Bs + (size_t)buf * BN * LDA + (wn * WN + li) * LDA;
        if constexpr (PREC == SGD_PREC_F32) {
#pragma unroll
            for (int ks = 0; ks < KC / 8; ++ks) {
                f32x4 av[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    av[mt] = *reinterpret_cast<const f32x4*>(As + (size_t)(pixbase[mt] + tapoff) * LDA + ks * 8 + lh * 4);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bv[nt] = *reinterpret_cast<const f32x4*>(bp + nt * 32 * LDA + ks * 8 + lh * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][j], bv[nt][j], acc[mt][nt], 0, 0, 0);
            }
        } else {
            typedef typename Split<PREC>::T T;
            typedef T T8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                // 16 channels per MFMA: lane half lh supplies channels 8*lh .. 8*lh+7 of the step
                T8 ah[MT], al[MT], bh[NT], bl[NT];
                const int goff = (ks * 2 + lh) * 8;      // float offset of the 8-group (32 B)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float* p = As + (size_t)(pixbase[mt] + tapoff) * LDA + goff;
                    ah[mt] = *reinterpret_cast<const T8*>(p);
                    al[mt] = *reinterpret_cast<const T8*>(p + 4);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float* p = bp + nt * 32 * LDA + goff;
                    bh[nt] = *reinterpret_cast<const T8*>(p);
                    bl[nt] = *reinterpret_cast<const T8*>(p + 4);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (PREC == SGD_PREC_F16X3) {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        } else {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        }
                    }
            }
        }
    };

    // ---- main loop ------------------------------------------------------------------------
    load_B(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int kc0 = ch * KC;
        // (the barrier closing the previous chunk's last tap already ordered all As reads)
        stage_A(kc0);
        if (ch == 0) store_B(0);
        __syncthreads();
        for (int tap = 0; tap < taps; ++tap) {
            const int step = ch * taps + tap;
            const bool more = (tap + 1 < taps) || (ch + 1 < nchunks);
            if (more) {
                if (tap + 1 < taps) load_B(tap + 1, kc0);
                else load_B(0, kc0 + KC);
            }
            compute(tap, step & 1);
            if (more) store_B((step + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue --------------------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = n0c + wn * WN + nt * 32 + li;
        if (col >= a.cout) continue;
        const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                long orow;        // output row index
                int n = 0, oy = 0, ox = 0;
                if (conv) {
                    int tx = row & (TW - 1), ty = (row >> g.tw_l2) & (TH - 1), nb = row >> (g.tw_l2 + g.th_l2);
                    n = img0 + nb; oy = ty0 + ty; ox = tx0 + tx;
                    if (n >= a.n) continue;
                    orow = ((long)n * a.ho + oy) * a.wo + ox;
                } else {
                    orow = m0 + row;
                    if (orow >= M) continue;
                }
                float v = acc[mt][nt][r] + bias;
                if (a.res) {
                    if (a.res_mode == SGD_RS_NONE) {
                        v += a.res[orow * a.cout + col];
                    } else if (a.res_mode == SGD_RS_AVGPOOL2) {
                        const int rw = a.wo * 2;
                        const float* rp = a.res + (((long)n * a.ho * 2 + 2 * oy) * rw + 2 * ox) * a.cout + col;
                        v += 0.25f * (rp[0] + rp[a.cout] + rp[(long)rw * a.cout] + rp[(long)(rw + 1) * a.cout]);
                    } else {
                        const int rw = a.wo >> 1;
                        v += a.res[(((long)n * (a.ho >> 1) + (oy >> 1)) * rw + (ox >> 1)) * a.cout + col];
                    }
                }
                if (a.orows_in > 0) orow = (orow / a.orows_in) * a.orows_out + a.orow_off + orow % a.orows_in;
                a.y[orow * a.y_ld + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing:  OIHW [cout, cin, k, k]  ->  [tap][cout_p][cin_p] 4-byte elements
// ---------------------------------------------------------------------------------------------
template <int PREC>
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin,
                                   int ks, int cout_p, int cin_p) {
    const long total = (long)ks * ks * cout_p * cin_p;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int ci = i % cin_p;
        long t = i / cin_p;
        int co = t % cout_p;
        int tap = t / cout_p;
        float v = 0.f;
        if (ci < cin && co < cout) v = src[((long)co * cin + ci) * ks * ks + tap];
        if constexpr (PREC == SGD_PREC_F32) {
            dst[i] = v;
        } else {
            typedef typename Split<PREC>::T T;
            T h = Split<PREC>::hi(v);
            T l = Split<PREC>::hi(v - Split<PREC>::back(h));
            // same layout as the LDS rows: 8-channel groups, hi[8] | lo[8]
            T* row = reinterpret_cast<T*>(dst + (i - ci));
            int gidx = ci >> 3, p = ci & 7;
            row[gidx * 16 + p] = h;
            row[gidx * 16 + 8 + p] = l;
        }
    }
}

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

template <int BN, int PREC>
int launch(const KArgs& ka, bool vec, size_t smem, hipStream_t st) {
    const int total = ka.g.mt * ka.g.nt;
    const int grid = ((total + 7) / 8) * 8;
    if (vec) {
        static bool attr_v = false;
        if (!attr_v) { (void)hipFuncSetAttribute((const void*)igemm_kernel<BN, PREC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_v = true; }
        hipLaunchKernelGGL((igemm_kernel<BN, PREC, true>), dim3(grid), dim3(256), smem, st, ka);
    } else {
        static bool attr_s = false;
        if (!attr_s) { (void)hipFuncSetAttribute((const void*)igemm_kernel<BN, PREC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_s = true; }
        hipLaunchKernelGGL((igemm_kernel<BN, PREC, false>), dim3(grid), dim3(256), smem, st, ka);
    }
    return sgd_check_launch();
}

}  // namespace

extern "C" int sgd_abi_version(void) { return SGD_ABI_VERSION; }

static inline int pick_bn(int cout) { return (cout % 128 == 0) ? 128 : 32; }

extern "C" int64_t sgd_packed_weight_bytes(int32_t cout, int32_t cin, int32_t ksize, int32_t prec) {
    (void)prec;
    const int bn = pick_bn(cout);
    const int64_t cout_p = (int64_t)((cout + bn - 1) / bn) * bn;
    const int64_t cin_p = (int64_t)((cin + KC - 1) / KC) * KC;
    return (int64_t)ksize * ksize * cout_p * cin_p * 4;
}

extern "C" int sgd_pack_weight(const float* w_src, void* w_dst, int32_t cout, int32_t cin, int32_t ksize,
                               int32_t prec, int32_t* cin_p_out, int32_t* cout_p_out, void* stream) {
    SGD_CLEAR_ERR();
    if (!w_src || !w_dst || cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return SGD_ERR_ARG;
    const int bn = pick_bn(cout);
    const int cout_p = ((cout + bn - 1) / bn) * bn;
    const int cin_p = ((cin + KC - 1) / KC) * KC;
    if (cin_p_out) *cin_p_out = cin_p;
    if (cout_p_out) *cout_p_out = cout_p;
    const long total = (long)ksize * ksize * cout_p * cin_p;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipStream_t st = (hipStream_t)stream;
    float* dst = reinterpret_cast<float*>(w_dst);
    if (prec == SGD_PREC_F32) hipLaunchKernelGGL((pack_weight_kernel<SGD_PREC_F32>), dim3(grid), dim3(256), 0, st, w_src, dst, cout, cin, ksize, cout_p, cin_p);
    else if (prec == SGD_PREC_F16X3) hipLaunchKernelGGL((pack_weight_kernel<SGD_PREC_F16X3>), dim3(grid), dim3(256), 0, st, w_src, dst, cout, cin, ksize, cout_p, cin_p);
    else if (prec == SGD_PREC_BF16X3) hipLaunchKernelGGL((pack_weight_kernel<SGD_PREC_BF16X3>), dim3(grid), dim3(256), 0, st, w_src, dst, cout, cin, ksize, cout_p, cin_p);
    else return SGD_ERR_ARG;
    return sgd_check_launch();
}

extern "C" int sgd_igemm(const sgd_igemm_args* args, void* stream) {
    SGD_CLEAR_ERR();
    if (!args) return SGD_ERR_ARG;
    KArgs ka;
    ka.a = *args;
    sgd_igemm_args& a = ka.a;
    Geo& g = ka.g;
    if (!a.x0 || !a.w || !a.y || a.c0 <= 0 || a.c1 < 0 || a.cout <= 0) return SGD_ERR_ARG;
    if (a.c1 > 0 && (!a.x1 || a.c0 % KC != 0)) return SGD_ERR_ARG;
    if (a.y_ld < a.cout) return SGD_ERR_ARG;
    if (a.pro != SGD_PRO_NONE && (!a.pa || !a.pb)) return SGD_ERR_ARG;
    const int cin = a.c0 + a.c1;
    const int bn = pick_bn(a.cout);
    if (a.cout_p % bn != 0 || a.cout_p < a.cout || a.cin_p % KC != 0 || a.cin_p < cin) return SGD_ERR_ARG;
    const bool vec = (a.c0 % 4 == 0) && (a.c1 % 4 == 0);
    if (a.mode == SGD_MODE_CONV3) {
        if (a.n <= 0 || a.hi <= 0 || a.wi <= 0 || (a.stride != 1 && a.stride != 2)) return SGD_ERR_ARG;
        if (a.stride == 2 && a.resample != SGD_RS_NONE) return SGD_ERR_ARG;
        g.hc = a.resample == SGD_RS_AVGPOOL2 ? a.hi / 2 : (a.resample == SGD_RS_UP2 ? a.hi * 2 : a.hi);
        g.wc = a.resample == SGD_RS_AVGPOOL2 ? a.wi / 2 : (a.resample == SGD_RS_UP2 ? a.wi * 2 : a.wi);
        if (a.resample == SGD_RS_AVGPOOL2 && ((a.hi | a.wi) & 1)) return SGD_ERR_ARG;
        const int ho = a.stride == 2 ? (g.hc + 1) / 2 : g.hc, wo = a.stride == 2 ? (g.wc + 1) / 2 : g.wc;
        if (a.ho != ho || a.wo != wo) return SGD_ERR_ARG;
        if (!is_pow2(a.ho) || !is_pow2(a.wo) || a.ho < 2 || a.wo < 2) return SGD_ERR_ARG;
        if (a.res && a.res_mode == SGD_RS_UP2 && ((a.ho | a.wo) & 1)) return SGD_ERR_ARG;
        int tw = a.wo < 16 ? a.wo : 16;
        int th = BM / tw; if (th > a.ho) th = a.ho; if (th > 8 && tw == 16) th = 8;
        // keep th*tw <= 128 and a power of two
        while (th * tw > BM) th >>= 1;
        int nb = BM / (th * tw);
        g.tw_l2 = ilog2(tw); g.th_l2 = ilog2(th); g.nb = nb;
        g.tiles_x = a.wo / tw; g.tiles_y = a.ho / th;
        g.hh = a.stride == 2 ? 2 * th + 1 : th + 2;
        g.hw = a.stride == 2 ? 2 * tw + 1 : tw + 2;
        g.pix = nb * g.hh * g.hw;
        g.mt = ((a.n + nb - 1) / nb) * g.tiles_x * g.tiles_y;
    } else if (a.mode == SGD_MODE_FLAT) {
        if (a.m <= 0) return SGD_ERR_ARG;
        if (a.pro == SGD_PRO_AFFINE_NC && a.rows_per_n <= 0) return SGD_ERR_ARG;
        if (a.res && a.res_mode != SGD_RS_NONE) return SGD_ERR_ARG;
        g.tw_l2 = g.th_l2 = 0; g.nb = 1; g.tiles_x = g.tiles_y = 1; g.hh = g.hw = 1; g.hc = g.wc = 1;
        g.pix = BM;
        g.mt = (a.m + BM - 1) / BM;
    } else {
        return SGD_ERR_ARG;
    }
    g.nt = a.cout_p / bn;
    const size_t smem = ((size_t)g.pix * LDA + 2 * (size_t)bn * LDA) * sizeof(float);
    if (smem > 160 * 1024) return SGD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
#define SGD_DISPATCH(P)                                                            \
    (bn == 128 ? launch<128, P>(ka, vec, smem, st) : launch<32, P>(ka, vec, smem, st))
    switch (a.prec) {
        case SGD_PREC_F32: return SGD_DISPATCH(SGD_PREC_F32);
        case SGD_PREC_F16X3: return SGD_DISPATCH(SGD_PREC_F16X3);
        case SGD_PREC_BF16X3: return SGD_DISPATCH(SGD_PREC_BF16X3);
        default: return SGD_ERR_ARG;
    }
#undef SGD_DISPATCH
}
