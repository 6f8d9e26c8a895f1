// Backward kernels of the training step (gfx950): weight gradients, GroupNorm/SiLU backward, bias
// gradients, q_sample and the MSE loss.  Input gradients of convolutions / linears are the FORWARD
// kernel (igemm.hip) run on adjoint-packed weights (sgd_pack_weight_dgrad).
#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"
#include "prologue.h"

namespace {

__device__ __forceinline__ float dsilu(float z) {
    const float s = 1.0f / (1.0f + __expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

// =============================================================================================
// weight gradient.  GEMM view: D[co, ci] = sum_k GY[k, co] * U[k(+tap), ci], k = output rows.
// Block = 256 threads (4 waves 2x2, each 64 co x 64 ci) owns one (tap, co tile, ci tile, k slice);
// K tiles of 64 rows are staged in LDS as [row][128 ch] (NHWC rows as they are: no transpose needed,
// the f32 MFMA takes one scalar per lane and consecutive lanes read consecutive channels).
// =============================================================================================
constexpr int WK = 64;          // rows per K tile
constexpr int WT = 128;         // co / ci tile
constexpr int WLD = WT + 4;

struct WArgs {
    sgd_igemm_args a;           // forward descriptor (input side)
    const float* gy;
    int gy_ld, cout, ksplit, taps, rows, co_tiles, ci_tiles, ktiles, hc, wc, wo_l2, ho_l2, gvec;
    float* slabs;
    float* bslab;               // [ksplit][cout] partial column sums of gy (bias gradient), or NULL
    // wave-specialised kernel with pre-split operands (split_rows_kernel / act_split_kernel): 16-bit hi / lo planes of the
    // gradient [rows][cout] and of the activated input [source rows][cin]
    const void* gh; const void* gl; const void* uh; const void* ul;
    int no_flat_pipe;           // SGD_TUNE_WGRAD_NO_PIPE (A/B runs): the synchronous staging of the 1x1 / linear kernel
    int tap9;                   // 1: a strided 3x3 conv on the 1x1 / linear split kernel, one tap per block (round 5, see wgrad_impl)
};
__device__ __forceinline__ bool flat_pipe_off(const WArgs& w) { return w.no_flat_pipe != 0; }

template <bool VEC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WArgs w) {
    const sgd_igemm_args& a = w.a;
    __shared__ __attribute__((aligned(16))) float Gs[WK * WLD];
    __shared__ __attribute__((aligned(16))) float Us[WK * WLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    int bid = blockIdx.x;
    const int ks = bid % w.ksplit; bid /= w.ksplit;
    const int cit = bid % w.ci_tiles; bid /= w.ci_tiles;
    const int cot = bid % w.co_tiles; bid /= w.co_tiles;
    const int tap = bid;
    const int co0 = cot * WT, ci0 = cit * WT;
    const int cin = a.c0 + a.c1;
    const bool conv = a.mode == SGD_MODE_CONV3;
    const int dy = conv ? tap / 3 - 1 : 0, dx = conv ? tap % 3 - 1 : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this block's K tiles: ks, ks + ksplit, ...
    for (int kt = ks; kt < w.ktiles; kt += w.ksplit) {
        const int row0 = kt * WK;
        __syncthreads();
        // ---- stage GY tile [64 rows][128 co] and the activated input tile [64 rows (shifted)][128 ci]
        for (int idx = tid; idx < WK * (WT / 4); idx += 256) {
            const int r = idx >> 5, q = idx & 31;
            const int row = row0 + r;
            f32x4 gv = {0.f, 0.f, 0.f, 0.f}, uv = {0.f, 0.f, 0.f, 0.f};
            if (row < w.rows) {
                const int co = co0 + q * 4;
                if (co + 3 < w.cout) gv = ld4(w.gy + (long)row * w.gy_ld + co);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < w.cout) gv[j] = w.gy[(long)row * w.gy_ld + co + j];
                }
                const int c = ci0 + q * 4;
                if (c < cin) {
                    if (conv) {
                        // output pixel (n, oy, ox) -> conv-input pixel (oy*s + dy, ox*s + dx)
                        // ho, wo are powers of two (sgd_igemm's CONV3 contract): shifts, not divisions
                        const int ox = row & (a.wo - 1), t = row >> w.wo_l2;
                        const int oy = t & (a.ho - 1), n = t >> w.ho_l2;
                        const int y = oy * a.stride + dy, x = ox * a.stride + dx;
                        if (y >= 0 && y < w.hc && x >= 0 && x < w.wc) {
                            if (a.resample == SGD_RS_AVGPOOL2) {
#pragma unroll
                                for (int sy = 0; sy < 2; ++sy)
#pragma unroll
                                    for (int sx = 0; sx < 2; ++sx) {
                                        long rr = ((long)n * a.hi + 2 * y + sy) * a.wi + 2 * x + sx;
                                        uv += apply_pro(a, load_raw<VEC>(a, rr, c), load_coef<VEC>(a, n, rr, c), c, rr);
                                    }
                                uv = uv * 0.25f;
                            } else {
                                long rr = a.resample == SGD_RS_UP2 ? ((long)n * a.hi + (y >> 1)) * a.wi + (x >> 1)
                                                                   : ((long)n * a.hi + y) * a.wi + x;
                                uv = apply_pro(a, load_raw<VEC>(a, rr, c), load_coef<VEC>(a, n, rr, c), c, rr);
                            }
                        }
                    } else {
                        const int n = a.pro == SGD_PRO_AFFINE_NC ? row / a.rows_per_n : 0;
                        uv = apply_pro(a, load_raw<VEC>(a, row, c), load_coef<VEC>(a, n, row, c), c, row);
                    }
                }
            }
            *reinterpret_cast<f32x4*>(Gs + r * WLD + q * 4) = gv;
            *reinterpret_cast<f32x4*>(Us + r * WLD + q * 4) = uv;
        }
        __syncthreads();
        // ---- 32 k-pairs: A[i = co][k] = Gs[k][co], B[k][j = ci] = Us[k][ci]
#pragma unroll 4
        for (int s = 0; s < WK / 2; ++s) {
            const float* gp = Gs + (2 * s + lh) * WLD + wm * 64 + li;
            const float* up = Us + (2 * s + lh) * WLD + wn * 64 + li;
            const float a0 = gp[0], a1 = gp[32], b0 = up[0], b1 = up[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // ---- slab store: D rows = co (registers), cols = ci (lanes)
    float* slab = w.slabs + ((long)ks * w.taps + tap) * w.cout * cin;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int ci = ci0 + wn * 64 + nt * 32 + li;
            if (ci >= cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < w.cout) slab[(long)co * cin + ci] = acc[mt][nt][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// weight gradient of a 3x3 stride-1 conv in split precision (f16x3 / bf16x3), all nine taps per block.
// K tile = an 8x8 patch of output pixels of one image: the gradient tile GY[64 px][128 co] and the activated
// input halo tile U[10x10 px][32 ci] are staged ONCE (16-bit hi / lo planes) and feed 9 taps x 4 k-steps x 3
// products = 108 MFMAs per wave (wave = 32 co x 32 ci x 9 taps, 144 accumulator registers).  Both operands are
// consumed K-major (a lane needs 8 consecutive pixels of ONE channel) from row-major [pixel][channel] LDS
// planes through ds_read_b64_tr_b16, the hardware transposing read: no transpose pass, NHWC rows staged as is.
// Row pitches: GY 320 B (4 consecutive pixels -> bank offsets 0/64/128/192), U 64 B (4 consecutive halo pixels
// = 256 contiguous bytes): conflict-free for the 32-lane halves of the transposing read.
// ---------------------------------------------------------------------------------------------
constexpr int FGP = 160;        // GY plane row pitch (16-bit elements)

// TAPS == 9: 3x3 stride-1 conv as described above (U = 10x10 halo x 32 ci, pitch 64 B, accumulators = taps).
// TAPS == 1: 1x1 conv / linear: K tile = 64 consecutive rows, U = [64 rows][128 ci] at the GY pitch, the four
//            accumulators are the four 32-channel ci sub-blocks (same transposing reads, no halo).
// FK (1x1 / linear only): 0 = every staging form behind run-time flags (what a launch with a partial 128-channel gradient block,
// dropout, scalar rows ... takes); 1 / 2 / 3 = the register-pipelined staging alone, for no / GroupNorm-affine prologue, the per-tap
// form of a strided conv, the LayerNorm-row prologue: the launcher (wgrad_flat_kind) proves the pipelined form's conditions for
// the WHOLE launch, and the instance holds nothing else.  Round 5: with all forms in one body the kernel needed 256 registers and
// spilled 46 -- ~70 scratch reloads of loop-invariant 64-bit row pointers inside the pipelined loop, each of them a vector memory
// operation the in-order vmcnt counts behind the row requests it was supposed to overlap.
template <int PREC, bool VEC, int TAPS, int FK = 0>
__global__ __launch_bounds__(256, 2) void wgrad_conv_kernel(const WArgs w) {
    typedef typename Split<PREC>::T T;
    typedef T T4 __attribute__((ext_vector_type(4)));
    typedef T T8 __attribute__((ext_vector_type(8)));
    typedef short s4 __attribute__((ext_vector_type(4)));
    constexpr bool CONV = TAPS == 9;
    constexpr int NACC = CONV ? 9 : 4;
    constexpr int UROWS = CONV ? 100 : 64;
    constexpr int UPITCH = CONV ? 32 : FGP;
    constexpr int CIT = CONV ? 32 : 128;              // ci tile of a block
    const sgd_igemm_args& a = w.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    T* Gh = reinterpret_cast<T*>(wsm);
    T* Gl = Gh + 64 * FGP;
    T* Uh = Gl + 64 * FGP;
    T* Ul = Uh + UROWS * UPITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    int bid = blockIdx.x;
    const int ks = bid % w.ksplit; bid /= w.ksplit;
    const int cit = bid % w.ci_tiles; bid /= w.ci_tiles;
    const int cot = bid % w.co_tiles;
    const int tap = bid / w.co_tiles;                          // > 0 only in the per-tap form (w.tap9)
    const int co0 = cot * WT, ci0 = cit * CIT;
    const int cin = a.c0 + a.c1;
    const int pw = CONV ? a.wo >> 3 : 1, ppi = CONV ? pw * (a.ho >> 3) : 1;   // patches per row / per image
    // per-tap form: the gradient rows are the conv's output pixels, the input row of output pixel (n, oy, ox) under this
    // block's tap is conv-input pixel (oy * stride + tap / 3 - 1, ox * stride + tap % 3 - 1), or nothing (zero padding)
    const int tdy = tap / 3 - 1, tdx = tap - (tap / 3) * 3 - 1;
    auto tap_row = [&](long row, int& n) -> long {
        const int ox = (int)row & (a.wo - 1);
        const int t = (int)(row >> w.wo_l2);
        const int oy = t & (a.ho - 1);
        n = t >> w.ho_l2;
        const int y = oy * a.stride + tdy, x = ox * a.stride + tdx;
        return (y >= 0 && y < w.hc && x >= 0 && x < w.wc) ? ((long)n * a.hi + y) * a.wi + x : -1;
    };

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposing-read addresses of this lane: group gidx = lane >> 4 -> (k half, 16-channel half); lane 4q+p of the group
    // addresses pixel q of the 4-pixel block, channels 4p .. 4p+3
    const int gidx = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int kbase = 8 * (gidx >> 1) + q;                    // + 16 s + 4 rd
    const int chl = 16 * (gidx & 1) + 4 * pp;                 // channel inside a 32-channel block
    auto trd = [&](const T* p) -> T4 {
        s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
        return __builtin_bit_cast(T4, v);
    };
    auto split_store = [&](T* hp, T* lp, f32x4 v) {
        if constexpr (PREC == SGD_PREC_F16X3) {
            u32x2 h, l;
            split4_f16(v, h, l);                             // 8 vector instructions (prologue.h)
            *reinterpret_cast<u32x2*>(hp) = h;
            *reinterpret_cast<u32x2*>(lp) = l;
        } else {
            T4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T hj, lj;
                Split<PREC>::split(v[j], hj, lj);
                h[j] = hj;
                l[j] = lj;
            }
            *reinterpret_cast<T4*>(hp) = h;
            *reinterpret_cast<T4*>(lp) = l;
        }
    };

    // halo U[10 x 10 px][32 ci] fast path: the (up to) four items of a thread share their channel quad (idx & 7 is tid & 7 for
    // all of them) and their image, so the GroupNorm coefficients are loaded once per stage; raw rows come from clamped
    // addresses and the padding / tail items are masked after the transform.  Requests and transform are separate so that
    // the requests can go out BEFORE the GY rows are transformed: one memory latency per stage instead of two.
    f32x4 ur[4];
    long rrs[4];
    Coef ukq;
    int un = 0, uy0 = 0, ux0 = 0;
    auto u_request = [&]() {
        const int c = ci0 + (tid & 7) * 4;
        const int cc = c < cin ? c : 0;
        ukq = load_coef<VEC>(a, un, 0, cc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hp = (tid >> 3) + i * 32;
            const int hy = hp / 10, hx = hp - hy * 10;
            int y = uy0 + hy - 1, x = ux0 + hx - 1;
            y = y < 0 ? 0 : (y >= w.hc ? w.hc - 1 : y);
            x = x < 0 ? 0 : (x >= w.wc ? w.wc - 1 : x);
            rrs[i] = a.resample == SGD_RS_UP2 ? ((long)un * a.hi + (y >> 1)) * a.wi + (x >> 1) : ((long)un * a.hi + y) * a.wi + x;
            ur[i] = load_raw<VEC>(a, rrs[i], cc);
        }
    };
    auto u_finish = [&]() {
        const int qd = tid & 7, c = ci0 + qd * 4;
        const int cc = c < cin ? c : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hp = (tid >> 3) + i * 32;
            if (hp >= 100) break;                                 // i == 3: only the threads of the first 4 pixel rows
            const int hy = hp / 10, hx = hp - hy * 10;
            const int y = uy0 + hy - 1, x = ux0 + hx - 1;
            f32x4 uv = apply_pro(a, ur[i], ukq, cc, rrs[i]);
            if (!(c < cin && y >= 0 && y < w.hc && x >= 0 && x < w.wc)) uv = f32x4{0.f, 0.f, 0.f, 0.f};
            split_store(Uh + hp * UPITCH + qd * 4, Ul + hp * UPITCH + qd * 4, uv);
        }
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                   // bias gradient: column sums of the staged gy rows
    const bool gy_fast = w.gvec && co0 + WT <= w.cout && w.rows > 0;
    const bool u_fast = CONV && VEC && (a.resample == SGD_RS_NONE || a.resample == SGD_RS_UP2)
                        && (a.pro == SGD_PRO_NONE || a.pro == SGD_PRO_AFFINE_NC);
    // 1x1 / linear, the common case (16-byte rows on both sides, whole 128-channel gradient block, GroupNorm affine of
    // ONE image per K tile or no prologue, no dropout): the rows of K tile kt + ksplit are requested into registers before
    // the MFMA phase of tile kt and transformed into LDS after it -- round 4: the staging below is synchronous (request ->
    // wait -> split -> barrier -> 48 MFMAs), two exposed memory latencies per 1.5k-cycle MFMA phase, 150 TF
    static_assert(FK == 0 || (!CONV && VEC), "specialised pipelined instances: 1x1 / linear, 16-byte rows");
    const bool tap9 = FK ? FK == 2 : w.tap9 != 0;
    bool flat_pipe = FK != 0;
    if constexpr (!CONV && FK == 0)
        flat_pipe = VEC && gy_fast && a.drop_p == 0.f && !flat_pipe_off(w)
                    && (a.pro == SGD_PRO_NONE || (a.pro == SGD_PRO_LN_ROW && !w.tap9)
                        || (a.pro == SGD_PRO_AFFINE_NC && (w.tap9 ? (a.ho * a.wo) % 64 == 0 : a.rows_per_n % 64 == 0)));
    if (flat_pipe) {
        const int qd = tid & 31, r0 = tid >> 5;
        const int c = ci0 + qd * 4;
        const int cc = c < cin ? c : 0;
        const float* gcol = w.gy + co0 + qd * 4;
        // pipelined at HALF-tile granularity (32 rows = two k-steps): while the MFMAs of one half run, the rows of the other
        // half are in flight -- 4 + 4 row quads per thread in registers (a whole tile's 16 would spill next to the 64
        // accumulators at two blocks per CU).  The two halves are disjoint LDS rows, so a half is overwritten while the
        // other is being read; two barriers per tile, as before.
        f32x4 gv[4], uv[4];
        Coef kq;
        // LayerNorm-row prologue (to_q / to_kv): (mean, rstd) of each requested row next to it, gamma / beta of this thread's
        // channel quad once per block
        const bool lnp = FK ? FK == 3 : a.pro == SGD_PRO_LN_ROW;
        float2 rst[4];
        f32x4 lng = {1.f, 1.f, 1.f, 1.f}, lnb = {0.f, 0.f, 0.f, 0.f};
        if (lnp) {
            lng = ld4(a.pb + cc);
            if (a.pc) lnb = ld4(a.pc + cc);
        }
        unsigned okx = 0xFu;                                     // per-tap form: which of the four requested input rows exist
        auto request = [&](int kt, int half) __attribute__((always_inline)) {
            const long base = (long)kt * 64;
            const long rlast = w.rows - 1;
            const long rfirst = base < w.rows ? base : rlast;
            // image of the K tile (64 rows of ONE image in both forms)
            const int nimg = a.pro != SGD_PRO_AFFINE_NC ? 0
                             : (tap9 ? (int)(rfirst >> (w.wo_l2 + w.ho_l2)) : (int)(rfirst / a.rows_per_n));
            if (!lnp) kq = load_coef<true>(a, nimg, rfirst, cc);
            okx = 0xFu;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                long row = base + r0 + (half * 4 + i) * 8;
                row = row < w.rows ? row : rlast;
                gv[i] = ld4(gcol + row * w.gy_ld);
                long urow = row;
                if (tap9) {                                     // (wave-uniform flag; the load stays unconditional)
                    int nn;
                    const long sr = tap_row(row, nn);
                    if (sr < 0) okx &= ~(1u << i);
                    urow = sr >= 0 ? sr : 0;
                }
                uv[i] = load_raw<true>(a, urow, cc);
                rst[i] = lnp ? *reinterpret_cast<const float2*>(a.pa + row * 2) : float2{0.f, 1.f};
            }
        };
        auto store_half = [&](int kt, int half) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + (half * 4 + i) * 8;
                const long row = (long)kt * 64 + r;
                f32x4 g4 = gv[i];
                f32x4 u4;
                if (lnp) {
                    u4 = (uv[i] - rst[i].x) * rst[i].y * lng + lnb;
                    if (a.pro_silu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) u4[e] = sgd_silu(u4[e]);
                    }
                } else {
                    u4 = apply_pro(a, uv[i], kq, cc, row < w.rows ? row : w.rows - 1);
                }
                if (row >= w.rows) { g4 = f32x4{0.f, 0.f, 0.f, 0.f}; u4 = g4; }
                if (c >= cin || !((okx >> i) & 1u)) u4 = f32x4{0.f, 0.f, 0.f, 0.f};
                split_store(Gh + r * FGP + qd * 4, Gl + r * FGP + qd * 4, g4);
                split_store(Uh + r * UPITCH + qd * 4, Ul + r * UPITCH + qd * 4, u4);
                bsum += g4;
            }
        };
        // (round 5: the four waves tile the block's [128 co x 128 ci] as 2 x 2 -- 64 co x 64 ci each -- instead of 4 x 1 (32 co x
        // 128 ci): 8 + 8 transposed fragment reads per k-step instead of 4 + 16 for the same 12 MFMAs; same products, same order)
        auto mma_half = [&](int half) __attribute__((always_inline)) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int k0 = 16 * (half * 2 + s2) + kbase;
                T8 ah[2], al[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const T* g0 = Gh + k0 * FGP + (wave & 1) * 64 + cb * 32 + chl;
                    const T* g1 = Gl + k0 * FGP + (wave & 1) * 64 + cb * 32 + chl;
                    const T4 h0 = trd(g0), h1 = trd(g0 + 4 * FGP), l0 = trd(g1), l1 = trd(g1 + 4 * FGP);
                    ah[cb] = T8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    al[cb] = T8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
#pragma unroll
                for (int cj = 0; cj < 2; ++cj) {
                    const T* u0 = Uh + k0 * UPITCH + (wave >> 1) * 64 + cj * 32 + chl;
                    const T* u1 = Ul + k0 * UPITCH + (wave >> 1) * 64 + cj * 32 + chl;
                    const T4 h0 = trd(u0), h1 = trd(u0 + 4 * UPITCH), l0 = trd(u1), l1 = trd(u1 + 4 * UPITCH);
                    const T8 bh = T8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const T8 bl = T8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        f32x16& c = acc[cb * 2 + cj];
                        if constexpr (PREC == SGD_PREC_F16X3) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bh, c, 0, 0, 0);
                        } else {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh, c, 0, 0, 0);
                        }
                    }
                }
            }
        };
        int kt = ks;
        if (kt < w.ktiles) request(kt, 0);
        for (; kt < w.ktiles; kt += w.ksplit) {
            store_half(kt, 0);                                   // (every wave is past the previous tile's first half: barrier 2)
            request(kt, 1);
            __syncthreads();                                     // barrier 1: first half staged
            mma_half(0);
            store_half(kt, 1);                                   // (every wave is past the previous tile's second half: barrier 1)
            if (kt + w.ksplit < w.ktiles) request(kt + w.ksplit, 0);
            __syncthreads();                                     // barrier 2: second half staged
            mma_half(1);
        }
    } else if constexpr (FK == 0)
    for (int kt = ks; kt < w.ktiles; kt += w.ksplit) {
        const int n = CONV ? kt / ppi : 0, pr = kt - n * ppi;
        const int y0 = CONV ? (pr / pw) * 8 : 0, x0 = CONV ? (pr - (pr / pw) * pw) * 8 : 0;
        __syncthreads();
        un = n; uy0 = y0; ux0 = x0;
        // ---- stage GY[64 rows][128 co]
        // Fast path (whole 128-channel block inside cout, 16-byte rows): all eight row quads of a thread are requested
        // before the first is used, from clamped addresses, and masked afterwards.  With the bounds checks around the
        // loads the compiler emitted load -> s_waitcnt vmcnt(0) -> store per item: eight memory latencies in a row per
        // stage (SQ_WAIT_ANY 56 % of the wave cycles, matrix pipe 27 % busy).
        if (gy_fast) {
            f32x4 gv[8];
            const int qd = tid & 31, r0 = tid >> 5;
            const float* gcol = w.gy + co0 + qd * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i * 8;
                long row = CONV ? ((long)n * a.ho + y0 + (r >> 3)) * a.wo + x0 + (r & 7) : (long)kt * 64 + r;
                row = row < w.rows ? row : w.rows - 1;
                gv[i] = ld4(gcol + row * w.gy_ld);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i * 8;
                const long row = CONV ? ((long)n * a.ho + y0 + (r >> 3)) * a.wo + x0 + (r & 7) : (long)kt * 64 + r;
                if (row >= w.rows) gv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                split_store(Gh + r * FGP + qd * 4, Gl + r * FGP + qd * 4, gv[i]);
                bsum += gv[i];
            }
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * 256;
            const int r = idx >> 5, qd = idx & 31;
            const long row = CONV ? ((long)n * a.ho + y0 + (r >> 3)) * a.wo + x0 + (r & 7) : (long)kt * 64 + r;
            const int co = co0 + qd * 4;
            f32x4 gv = {0.f, 0.f, 0.f, 0.f};
            if (row < w.rows) {
                if (w.gvec && co + 3 < w.cout) gv = ld4(w.gy + row * w.gy_ld + co);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < w.cout) gv[j] = w.gy[row * w.gy_ld + co + j];
                }
            }
            split_store(Gh + r * FGP + qd * 4, Gl + r * FGP + qd * 4, gv);
            bsum += gv;                                   // this thread's co quad (idx & 31) is the same for all its items
        }
        }
        // ---- stage the activated input
        if constexpr (CONV) {
            if (u_fast) {
                u_request();
                u_finish();
            } else
            // halo U[10 x 10 px][32 ci] (conv-input space, zero outside the image)
            for (int idx = tid; idx < 100 * (CIT / 4); idx += 256) {
                const int hp = idx >> 3, qd = idx & 7;
                const int hy = hp / 10, hx = hp - hy * 10;
                const int y = y0 + hy - 1, x = x0 + hx - 1;
                const int c = ci0 + qd * 4;
                f32x4 uv = {0.f, 0.f, 0.f, 0.f};
                if (c < cin && y >= 0 && y < w.hc && x >= 0 && x < w.wc) {
                    if (a.resample == SGD_RS_AVGPOOL2) {
#pragma unroll
                        for (int sy = 0; sy < 2; ++sy)
#pragma unroll
                            for (int sx = 0; sx < 2; ++sx) {
                                long rr = ((long)n * a.hi + 2 * y + sy) * a.wi + 2 * x + sx;
                                uv += apply_pro(a, load_raw<VEC>(a, rr, c), load_coef<VEC>(a, n, rr, c), c, rr);
                            }
                        uv = uv * 0.25f;
                    } else {
                        long rr = a.resample == SGD_RS_UP2 ? ((long)n * a.hi + (y >> 1)) * a.wi + (x >> 1)
                                                           : ((long)n * a.hi + y) * a.wi + x;
                        uv = apply_pro(a, load_raw<VEC>(a, rr, c), load_coef<VEC>(a, n, rr, c), c, rr);
                    }
                }
                split_store(Uh + hp * UPITCH + qd * 4, Ul + hp * UPITCH + qd * 4, uv);
            }
        } else if (VEC && w.rows > 0) {
            // 1x1 / linear: eight row quads per thread, requested four at a time from clamped addresses (raw row + the
            // GroupNorm coefficients of its image, or the LayerNorm statistics of the row), masked after the transform
            const int qd = tid & 31, r0 = tid >> 5;
            const int c = ci0 + qd * 4;
            const int cc = c < cin ? c : 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 uraw[4];
                Coef kq[4];
                long rows4[4];
                bool okr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long row = (long)kt * 64 + r0 + (half * 4 + i) * 8;
                    rows4[i] = row < w.rows ? row : w.rows - 1;
                    int ni = a.pro == SGD_PRO_AFFINE_NC && !w.tap9 ? (int)(rows4[i] / a.rows_per_n) : 0;
                    okr[i] = true;
                    if (w.tap9) {                                     // (wave-uniform flag; the loads below stay unconditional)
                        const long sr = tap_row(rows4[i], ni);
                        okr[i] = sr >= 0;
                        rows4[i] = sr >= 0 ? sr : 0;
                    }
                    kq[i] = load_coef<VEC>(a, ni, rows4[i], cc);
                    uraw[i] = load_raw<VEC>(a, rows4[i], cc);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = r0 + (half * 4 + i) * 8;
                    f32x4 uv = apply_pro(a, uraw[i], kq[i], cc, rows4[i]);
                    if (!((long)kt * 64 + r < w.rows && c < cin && okr[i])) uv = f32x4{0.f, 0.f, 0.f, 0.f};
                    split_store(Uh + r * UPITCH + qd * 4, Ul + r * UPITCH + qd * 4, uv);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = tid + i * 256;
                const int r = idx >> 5, qd = idx & 31;
                const long row = (long)kt * 64 + r;
                const int c = ci0 + qd * 4;
                f32x4 uv = {0.f, 0.f, 0.f, 0.f};
                if (row < w.rows && c < cin) {
                    const int ni = a.pro == SGD_PRO_AFFINE_NC ? (int)(row / a.rows_per_n) : 0;
                    uv = apply_pro(a, load_raw<VEC>(a, row, c), load_coef<VEC>(a, ni, row, c), c, row);
                }
                split_store(Uh + r * UPITCH + qd * 4, Ul + r * UPITCH + qd * 4, uv);
            }
        }
        __syncthreads();
        // ---- 4 k-steps of 16 rows
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k0 = 16 * s + kbase;
            if constexpr (CONV) {
                T8 ah, al;
                {
                    const T* g0 = Gh + k0 * FGP + wave * 32 + chl;
                    const T* g1 = Gl + k0 * FGP + wave * 32 + chl;
                    const T4 h0 = trd(g0), h1 = trd(g0 + 4 * FGP), l0 = trd(g1), l1 = trd(g1 + 4 * FGP);
                    ah = T8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    al = T8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
                // halo pixel of tap (0, 0) for this lane's k (patch row k0 >> 3, col k0 & 7); read 1 = 4 pixels on
                const int hb = ((k0 >> 3) + 1) * 10 + (k0 & 7) + 1;
#pragma unroll
                for (int t = 0; t < NACC; ++t) {
                    const int off = ((t / 3 - 1) * 10 + (t % 3 - 1)) * UPITCH;
                    const T* u0 = Uh + hb * UPITCH + off + chl;
                    const T* u1 = Ul + hb * UPITCH + off + chl;
                    const T4 h0 = trd(u0), h1 = trd(u0 + 4 * UPITCH), l0 = trd(u1), l1 = trd(u1 + 4 * UPITCH);
                    const T8 bh = T8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const T8 bl = T8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    if constexpr (PREC == SGD_PREC_F16X3) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                    } else {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                    }
                }
            } else {
                // 1x1 / linear: 2 x 2 wave tiles (see mma_half above)
                T8 ah[2], al[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const T* g0 = Gh + k0 * FGP + (wave & 1) * 64 + cb * 32 + chl;
                    const T* g1 = Gl + k0 * FGP + (wave & 1) * 64 + cb * 32 + chl;
                    const T4 h0 = trd(g0), h1 = trd(g0 + 4 * FGP), l0 = trd(g1), l1 = trd(g1 + 4 * FGP);
                    ah[cb] = T8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    al[cb] = T8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
#pragma unroll
                for (int cj = 0; cj < 2; ++cj) {
                    const T* u0 = Uh + k0 * UPITCH + (wave >> 1) * 64 + cj * 32 + chl;
                    const T* u1 = Ul + k0 * UPITCH + (wave >> 1) * 64 + cj * 32 + chl;
                    const T4 h0 = trd(u0), h1 = trd(u0 + 4 * UPITCH), l0 = trd(u1), l1 = trd(u1 + 4 * UPITCH);
                    const T8 bh = T8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const T8 bl = T8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        f32x16& c = acc[cb * 2 + cj];
                        if constexpr (PREC == SGD_PREC_F16X3) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb], bh, c, 0, 0, 0);
                        } else {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh, c, 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if (w.bslab && cit == 0 && tap == 0) {
        // fold the 8 row lanes of every co quad through LDS (the planes are free after the last MFMA phase)
        __syncthreads();
        float* red = reinterpret_cast<float*>(wsm);       // [8][128]
        *reinterpret_cast<f32x4*>(red + (tid >> 5) * 128 + (tid & 31) * 4) = bsum;
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 128 + tid];
            if (co0 + tid < w.cout) w.bslab[(long)ks * w.cout + co0 + tid] = t;
        }
    }
    // ---- slab store: D rows = co (registers), cols = ci (lanes)
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        // CONV: accumulator = tap, wave = 32 co; 1x1 / linear: accumulator (cb, cj) = t >> 1, t & 1 of the wave's 64 co x 64 ci
        const int ci = ci0 + (CONV ? 0 : (wave >> 1) * 64 + (t & 1) * 32) + li;
        if (ci >= cin) continue;
        float* slab = w.slabs + ((long)ks * (w.tap9 ? 9 : TAPS) + (CONV ? t : tap)) * w.cout * cin;
        const int cow = co0 + (CONV ? wave * 32 : (wave & 1) * 64 + (t >> 1) * 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cow + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co < w.cout) slab[(long)co * cin + ci] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pre-passes of the wave-specialised weight-gradient kernel below.  Its loader waves can issue about one vector
// instruction per 16 cycles next to the MFMA waves; the GroupNorm-affine + SiLU + hi / lo split of the input halo and the
// split of the gradient rows cost ~450 of them per K tile against a 3.5k-cycle MFMA phase -- and every block repeats the
// gradient split for its 32 input channels (cin / 32 times per element) and the input transform for its 128 output
// channels (cout / 128 times).  The two element-wise kernels here do that arithmetic ONCE per element, into 16-bit hi / lo
// planes ([row][channel], 2 + 2 bytes per element: the size of the fp32 tensor); the loaders then only copy.
// Measured on the C2 training step (bs 80): the transforms cost 11.5 of the 22.5 ms of weight-gradient time.
// ---------------------------------------------------------------------------------------------
template <int PREC>
__device__ __forceinline__ void split_rows_body(const float* __restrict__ g, long rows, int c, int ld,
                                                typename Split<PREC>::T* __restrict__ hi,
                                                typename Split<PREC>::T* __restrict__ lo, int chunks,
                                                float* __restrict__ colsum, int bx, int by) {
    // block = 128 channels (32 quads) x 8 row lanes over the rows of chunk `by`; colsum[chunk][c] = column sums of
    // the chunk (the bias gradient's partial sums, same layout as sgd_wgrad's bias_slabs), or NULL
    typedef typename Split<PREC>::T T;
    typedef T T4 __attribute__((ext_vector_type(4)));
    const int q = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = bx * 128 + q * 4;
    const long per = (rows + chunks - 1) / chunks;
    const long r0 = by * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (col < c) {
        for (long r = r0 + rl; r < r1; r += 32) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long rr = r + 8 * j < r1 ? r + 8 * j : r1 - 1;
                v[j] = ld4(g + rr * ld + col);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (r + 8 * j >= r1) break;
                sum += v[j];
                T* hp = hi + (r + 8 * j) * c + col;
                T* lp = lo + (r + 8 * j) * c + col;
                if constexpr (PREC == SGD_PREC_F16X3) {
                    u32x2 h, l;
                    split4_f16(v[j], h, l);
                    *reinterpret_cast<u32x2*>(hp) = h;
                    *reinterpret_cast<u32x2*>(lp) = l;
                } else {
                    T4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { T he, le; Split<PREC>::split(v[j][e], he, le); h[e] = he; l[e] = le; }
                    *reinterpret_cast<T4*>(hp) = h;
                    *reinterpret_cast<T4*>(lp) = l;
                }
            }
        }
    }
    if (colsum) {
        __shared__ float red[8][128];
        *reinterpret_cast<f32x4*>(&red[rl][q * 4]) = sum;
        __syncthreads();
        if (threadIdx.x < 128 && bx * 128 + threadIdx.x < c) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
            colsum[(long)by * c + bx * 128 + threadIdx.x] = t;
        }
    }
}

// activated conv input (virtual concat x0 | x1, GroupNorm affine, SiLU, train-time dropout: apply_pro) as hi / lo planes
// [source row][cin]; resampling stays in the consumer's index map
// pool: the conv reads avg_pool2d(act(x)) (ResBlock(down), openaimodel.py:301-306): the planes are written at the POOLED
// resolution -- rows = n * (hi / 2) * (wi / 2) -- and the consumer sees an unresampled input of those dims (round 4: these
// launches ran on the generic per-tap kernel at ~400 us each, seven times their share)
template <int PREC>
__device__ __forceinline__ void act_split_body(const sgd_igemm_args& a, long rows, int rows_per_n,
                                               typename Split<PREC>::T* __restrict__ hi,
                                               typename Split<PREC>::T* __restrict__ lo, int pool, long blk, long nblk) {
    typedef typename Split<PREC>::T T;
    typedef T T4 __attribute__((ext_vector_type(4)));
    const int cin = a.c0 + a.c1, cq = cin >> 2;
    const long total = rows * cq;
    for (long i = blk * (long)blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
        const long row = i / cq;
        const int c = (int)(i - row * cq) * 4;
        const int n = (int)(row / rows_per_n);
        f32x4 v;
        if (pool) {
            const int wp = a.wi >> 1;
            const int pr = (int)(row - (long)n * rows_per_n), py = pr / wp, px = pr - py * wp;
            const long r00 = ((long)n * a.hi + 2 * py) * a.wi + 2 * px;
            v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const long r = r00 + dy * a.wi + dx;
                    v += apply_pro(a, load_raw<true>(a, r, c), load_coef<true>(a, n, r, c), c, r);
                }
            v = v * 0.25f;
        } else {
            v = apply_pro(a, load_raw<true>(a, row, c), load_coef<true>(a, n, row, c), c, row);
        }
        T* hp = hi + row * cin + c;
        T* lp = lo + row * cin + c;
        if constexpr (PREC == SGD_PREC_F16X3) {
            u32x2 h, l;
            split4_f16(v, h, l);
            *reinterpret_cast<u32x2*>(hp) = h;
            *reinterpret_cast<u32x2*>(lp) = l;
        } else {
            T4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { T he, le; Split<PREC>::split(v[e], he, le); h[e] = he; l[e] = le; }
            *reinterpret_cast<T4*>(hp) = h;
            *reinterpret_cast<T4*>(lp) = l;
        }
    }
}

// Both pre-passes of a weight-gradient launch in ONE launch (round 5: 88 launches fewer per training step -- each boundary
// between two short kernels costs the stream ~4 us): blocks [0, sr_blocks) split the gradient rows (block = (column tile,
// row chunk), column sums of its chunk to colsum[chunk][c]), the others the activated input.
struct PrepassArgs {
    const float* g; long grows; int gc, gld; void* gh; void* gl; int chunks; float* colsum; int coltiles, sr_blocks;
    sgd_igemm_args a; long urows; int rows_per_n; void* uh; void* ul; int pool;
};
template <int PREC>
__global__ __launch_bounds__(256) void wgrad_prepass_kernel(const PrepassArgs p) {
    typedef typename Split<PREC>::T T;
    if ((int)blockIdx.x < p.sr_blocks)
        split_rows_body<PREC>(p.g, p.grows, p.gc, p.gld, reinterpret_cast<T*>(p.gh), reinterpret_cast<T*>(p.gl), p.chunks, p.colsum,
                              (int)blockIdx.x % p.coltiles, (int)blockIdx.x / p.coltiles);
    else
        act_split_body<PREC>(p.a, p.urows, p.rows_per_n, reinterpret_cast<T*>(p.uh), reinterpret_cast<T*>(p.ul), p.pool,
                             (long)blockIdx.x - p.sr_blocks, (long)gridDim.x - p.sr_blocks);
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised form of the 3x3 weight-gradient kernel (round 3).  The kernel above stages a K tile, waits, runs
// its 108 MFMAs per wave, waits, stages the next: the matrix pipe sat idle through two memory latencies and ~300
// vector instructions per tile (27..40 % busy even with two blocks per CU taking turns).  Here a block is 8 waves on
// one CU: waves 0-3 only multiply (tile i from LDS slot i & 1), waves 4-7 only stage (tile i + 1 into the other slot:
// GroupNorm affine + SiLU, hi / lo split) with the raw rows of tile i + 2 already requested -- the conv kernel's
// loader / compute structure, one s_barrier per tile, every load consumed a whole tile period after its issue.
// Fast cases only (16-byte gradient rows of a full 128-channel block, stride 1, no avg-pool, GroupNorm-affine or no
// prologue); everything else takes the kernel above.  Same arithmetic, same slab layout, same results bit for bit.
// ---------------------------------------------------------------------------------------------
template <int PREC, bool PLANES>
__global__ __launch_bounds__(512) void wgrad_conv_ws_kernel(const WArgs w) {
    typedef typename Split<PREC>::T T;
    typedef T T4 __attribute__((ext_vector_type(4)));
    typedef T T8 __attribute__((ext_vector_type(8)));
    typedef short s4 __attribute__((ext_vector_type(4)));
    constexpr int UPITCH = 32;
    constexpr int SLOT = 2 * 64 * FGP + 2 * 100 * UPITCH;          // 16-bit elements of one K tile: Gh | Gl | Uh | Ul
    const sgd_igemm_args& a = w.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    T* const base = reinterpret_cast<T*>(wsm);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int bid = blockIdx.x;
    const int ks = bid % w.ksplit; bid /= w.ksplit;
    const int cit = bid % w.ci_tiles; bid /= w.ci_tiles;
    const int cot = bid;
    const int co0 = cot * WT, ci0 = cit * 32;
    const int cin = a.c0 + a.c1;
    const int pw = a.wo >> 3, ppi = pw * (a.ho >> 3);             // 8x8 patches per row / per image
    const int nk = ks < w.ktiles ? (w.ktiles - ks + w.ksplit - 1) / w.ksplit : 0;     // K tiles of this block

    if (tid >= 256) {
        // =============================== loader waves: global -> transform -> LDS ===============================
        const int lt = tid - 256;
        auto split_store = [&](T* hp, T* lp, f32x4 v) {
            if constexpr (PREC == SGD_PREC_F16X3) {
                u32x2 h, l;
                split4_f16(v, h, l);
                *reinterpret_cast<u32x2*>(hp) = h;
                *reinterpret_cast<u32x2*>(lp) = l;
            } else {
                T4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) { T hj, lj; Split<PREC>::split(v[j], hj, lj); h[j] = hj; l[j] = lj; }
                *reinterpret_cast<T4*>(hp) = h;
                *reinterpret_cast<T4*>(lp) = l;
            }
        };
        // The loader waves share their SIMDs' vector issue with waves that always have an MFMA waiting: every vector
        // instruction here costs ~16 cycles of the tile period (measured on the conv kernel's loaders), so the per-tile
        // index arithmetic is hoisted -- a thread's rows inside a patch and its halo pixels never change.
        __builtin_amdgcn_s_setprio(2);
        // raw data of the tile being fetched (requested one period before it is transformed)
        f32x4 gv[8], ur[4];
        long rrs[4];
        Coef ukq;
        int rn = 0, ry0 = 0, rx0 = 0;                              // patch of the requested tile
        const int gq = lt & 31, gr0 = lt >> 5;                    // gradient rows: channel quad, first row
        const int uq = lt & 7, uc = ci0 + uq * 4, ucc = uc < cin ? uc : 0;
        const float* gcol = w.gy + co0 + gq * 4;
        // row j of this thread inside an 8x8 patch: patch row j (r = gr0 + 8 j -> r >> 3 = j), patch column gr0
        const long gstep = (long)a.wo * w.gy_ld;                   // one patch row down
        int hys[4], hxs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hp = (lt >> 3) + j * 32;
            hys[j] = hp / 10 - 1;
            hxs[j] = hp - (hp / 10) * 10 - 1;
        }
        // PLANES: the operands arrive already split (pre-pass kernels above): 8 + 8 bytes per item, copied as they are
        u32x2 gph[8], gpl[8], uph[4], upl[4];
        const T* const ghp = reinterpret_cast<const T*>(w.gh) + co0 + gq * 4;
        const T* const glp = reinterpret_cast<const T*>(w.gl) + co0 + gq * 4;
        const T* const uhp = reinterpret_cast<const T*>(w.uh) + ucc;
        const T* const ulp = reinterpret_cast<const T*>(w.ul) + ucc;
        auto request = [&](int i) {                                // i-th K tile of this block (clamped: harmless duplicates)
            const int kt = ks + (i < nk ? i : nk - 1) * w.ksplit;
            rn = kt / ppi;
            const int pr = kt - rn * ppi;
            ry0 = (pr / pw) * 8;
            rx0 = (pr - (pr / pw) * pw) * 8;
            if constexpr (PLANES) {
                const long r0 = (((long)rn * a.ho + ry0) * a.wo + rx0 + gr0) * w.cout;
                const long rstep = (long)a.wo * w.cout;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    gph[j] = *reinterpret_cast<const u32x2*>(ghp + r0 + j * rstep);
                    gpl[j] = *reinterpret_cast<const u32x2*>(glp + r0 + j * rstep);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int y = ry0 + hys[j], x = rx0 + hxs[j];
                    y = y < 0 ? 0 : (y >= w.hc ? w.hc - 1 : y);
                    x = x < 0 ? 0 : (x >= w.wc ? w.wc - 1 : x);
                    const long rr = a.resample == SGD_RS_UP2 ? ((long)rn * a.hi + (y >> 1)) * a.wi + (x >> 1) : ((long)rn * a.hi + y) * a.wi + x;
                    uph[j] = *reinterpret_cast<const u32x2*>(uhp + rr * cin);
                    upl[j] = *reinterpret_cast<const u32x2*>(ulp + rr * cin);
                }
                return;
            }
            const float* g0 = gcol + (((long)rn * a.ho + ry0) * a.wo + rx0 + gr0) * w.gy_ld;
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = ld4(g0 + j * gstep);
            ukq = load_coef<true>(a, rn, 0, ucc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int y = ry0 + hys[j], x = rx0 + hxs[j];
                y = y < 0 ? 0 : (y >= w.hc ? w.hc - 1 : y);
                x = x < 0 ? 0 : (x >= w.wc ? w.wc - 1 : x);
                rrs[j] = a.resample == SGD_RS_UP2 ? ((long)rn * a.hi + (y >> 1)) * a.wi + (x >> 1) : ((long)rn * a.hi + y) * a.wi + x;
                ur[j] = load_raw<true>(a, rrs[j], ucc);
            }
        };
        f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                         // bias gradient: column sums of the staged gy rows
        auto finish = [&](int slot, float live) {                  // transform the fetched tile into LDS slot `slot`
            T* Gh = base + slot * SLOT;
            T* Gl = Gh + 64 * FGP;
            T* Uh = Gl + 64 * FGP;
            T* Ul = Uh + 100 * UPITCH;
            if constexpr (PLANES) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = gr0 + j * 8;
                    *reinterpret_cast<u32x2*>(Gh + r * FGP + gq * 4) = gph[j];
                    *reinterpret_cast<u32x2*>(Gl + r * FGP + gq * 4) = gpl[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int hp = (lt >> 3) + j * 32;
                    if (hp >= 100) break;
                    const int y = ry0 + hys[j], x = rx0 + hxs[j];
                    const bool ok = uc < cin && y >= 0 && y < w.hc && x >= 0 && x < w.wc;     // zero padding of the conv
                    u32x2 h = uph[j], l = upl[j];
                    if (!ok) { h.x = h.y = 0; l.x = l.y = 0; }
                    *reinterpret_cast<u32x2*>(Uh + hp * UPITCH + uq * 4) = h;
                    *reinterpret_cast<u32x2*>(Ul + hp * UPITCH + uq * 4) = l;
                }
                (void)live;
                return;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = gr0 + j * 8;
#if defined(SGDM_WS_ABL) && (SGDM_WS_ABL & 1)      /* timing experiment: gradient rows stored without the split arithmetic */
                *reinterpret_cast<float2*>(Gh + r * FGP + gq * 4) = float2{gv[j][0], gv[j][1]};
                *reinterpret_cast<float2*>(Gl + r * FGP + gq * 4) = float2{gv[j][2], gv[j][3]};
#else
                split_store(Gh + r * FGP + gq * 4, Gl + r * FGP + gq * 4, gv[j]);
                bsum += gv[j] * live;                              // (a clamped duplicate past the block's last tile: x 0)
#endif
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hp = (lt >> 3) + j * 32;
                if (hp >= 100) break;
                const int y = ry0 + hys[j], x = rx0 + hxs[j];
#if defined(SGDM_WS_ABL) && (SGDM_WS_ABL & 2)      /* timing experiment: input rows stored without transform / split */
                *reinterpret_cast<float2*>(Uh + hp * UPITCH + uq * 4) = float2{ur[j][0], ur[j][1]};
                *reinterpret_cast<float2*>(Ul + hp * UPITCH + uq * 4) = float2{ur[j][2], ur[j][3]};
                (void)y; (void)x;
#else
                f32x4 uv = apply_pro(a, ur[j], ukq, ucc, rrs[j]);
                if (!(uc < cin && y >= 0 && y < w.hc && x >= 0 && x < w.wc)) uv = f32x4{0.f, 0.f, 0.f, 0.f};
                split_store(Uh + hp * UPITCH + uq * 4, Ul + hp * UPITCH + uq * 4, uv);
#endif
            }
        };
        if (nk > 0) {
            request(0);
            finish(0, 1.f);
            request(1);
        }
        __syncthreads();                                           // barrier 0: slot 0 holds tile 0
        for (int i = 0; i < nk; ++i) {
            finish((i + 1) & 1, i + 1 < nk ? 1.f : 0.f);           // tile i + 1 (fetched during the previous period)
            request(i + 2);
            __syncthreads();                                       // barrier i + 1
        }
        // bias gradient: fold the 8 row lanes of every co quad through LDS (the slots are free now)
        float* red = reinterpret_cast<float*>(wsm);                // [8][128]
        if (!PLANES && w.bslab && cit == 0) *reinterpret_cast<f32x4*>(red + (lt >> 5) * 128 + (lt & 31) * 4) = bsum;
        __syncthreads();
        if (!PLANES && w.bslab && cit == 0 && lt < 128) {         // (PLANES: the split pre-pass wrote the column sums)
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 128 + lt];
            if (co0 + lt < w.cout) w.bslab[(long)ks * w.cout + co0 + lt] = t;
        }
        return;
    }

    // =================================== compute waves: LDS -> MFMA ===================================
    // v_mfma_f32_16x16x32 (round 3): the same matrix-pipe cycles per flop as 32x32x16, and a measured +14..16 % clock under
    // this load (profiles/r3_pmc_mfma_form_clock.txt).  The wave's [64 co x 16 ci] tile of a tap is 4 x 1 blocks of 16 x 16,
    // a K step is 32 pixels of the 8x8 patch: lane group g = lane >> 4 addresses pixel rows 8 g + q (and + 4) of the step,
    // the transposing read hands lane j of a group channel j of the 16-channel block, 4 + 4 consecutive pixels.
    // (round 5: the four waves tile the block's [128 co x 32 ci] as 2 x 2 -- 64 co x 16 ci each -- instead of 4 x 1 (32 co x 32
    // ci): per K step a wave still issues 108 MFMAs but reads 16 + 36 transposed 8-byte LDS fragments instead of 8 + 72: the
    // input-side fragment is re-read for every tap, the gradient-side one only once per step, so the wave takes more of the
    // latter.  Same products in the same order into every accumulator: bit-identical slabs.)
    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int gidx = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int wco = wave & 1, wci = wave >> 1;
    // one per-lane element offset per operand; everything else (K step, block, tap, plane) is a compile-time constant that
    // folds into the read's immediate offset: pixel 32 s + 8 g + q of the patch sits at halo index
    // (4 s + g + 1) * 10 + q + 1
    const int goff = (8 * gidx + q) * FGP + wco * 64 + 4 * pp;
    const int uoff = ((gidx + 1) * 10 + q + 1) * UPITCH + wci * 16 + 4 * pp;
    auto trd = [&](const T* p) -> T4 {
        s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
        return __builtin_bit_cast(T4, v);
    };
    auto rd8 = [&](const T* p, int pitch4) -> T8 {                 // 8 consecutive pixels of this lane's channel
        const T4 x0 = trd(p), x1 = trd(p + pitch4);
        return T8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    };
    __syncthreads();                                               // barrier 0
    for (int i = 0; i < nk; ++i) {
        const T* Gh = base + (i & 1) * SLOT + goff;
        const T* Gl = Gh + 64 * FGP;
        const T* Uh = base + (i & 1) * SLOT + 2 * 64 * FGP + uoff;
        const T* Ul = Uh + 100 * UPITCH;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            T8 ah[4], al[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                ah[cb] = rd8(Gh + 32 * s * FGP + cb * 16, 4 * FGP);
                al[cb] = rd8(Gl + 32 * s * FGP + cb * 16, 4 * FGP);
            }
            // The three taps of a kernel row read three windows -- columns [0, 8), [1, 9), [2, 10) -- of the same ten halo pixels:
            // 12 consecutive pixels come in once (three transposed reads per plane; the last two values belong to the next halo
            // row and are not used) and the windows are formed in registers (the odd one costs four 16-bit funnel shifts per
            // plane).  18 input-side reads per K step instead of 36; the same operands into the same MFMAs.
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int off = (40 * s + (dy - 1) * 10 - 1) * UPITCH;
                const T4 h0 = trd(Uh + off), h1 = trd(Uh + off + 4 * UPITCH), h2 = trd(Uh + off + 8 * UPITCH);
                const T4 l0 = trd(Ul + off), l1 = trd(Ul + off + 4 * UPITCH), l2 = trd(Ul + off + 8 * UPITCH);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    T8 bh, bl;
                    if (dx == 0) {
                        bh = T8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                        bl = T8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    } else if (dx == 1) {
                        bh = T8{h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3], h2[0]};
                        bl = T8{l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3], l2[0]};
                    } else {
                        bh = T8{h0[2], h0[3], h1[0], h1[1], h1[2], h1[3], h2[0], h2[1]};
                        bl = T8{l0[2], l0[3], l1[0], l1[1], l1[2], l1[3], l2[0], l2[1]};
                    }
                    const int t = dy * 3 + dx;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        f32x4& c = acc[t][cb];
                        if constexpr (PREC == SGD_PREC_F16X3) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cb], bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cb], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cb], bh, c, 0, 0, 0);
                        } else {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cb], bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[cb], bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cb], bh, c, 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();                                           // barrier i + 1
    }
    __syncthreads();                                               // pairs with the loaders' bias-reduction barrier
    // ---- slab store: D block [16 co x 16 ci]: rows 4 (lane >> 4) + r in registers, column lane & 15
    const int ci = ci0 + wci * 16 + (lane & 15);
    if (ci >= cin) return;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* slab = w.slabs + ((long)ks * 9 + t) * w.cout * cin;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wco * 64 + cb * 16 + 4 * gidx + r;
                if (co < w.cout) slab[(long)co * cin + ci] = acc[t][cb][r];
            }
    }
}

template <int PREC, bool PLANES>
static void launch_wgrad_ws(const WArgs& w, long grid, hipStream_t st) {
    // two K-tile slots (+ 2 halo rows: the compute waves' 12-pixel reads of the last halo row run two rows past a plane)
    constexpr size_t smem = (size_t)2 * (2 * 64 * FGP + 2 * 100 * 32) * 2 + 2 * 32 * 2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)wgrad_conv_ws_kernel<PREC, PLANES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_conv_ws_kernel<PREC, PLANES>), dim3((unsigned)grid), dim3(512), smem, st, w);
}

// row chunks of the gradient split (one block per (128-column tile, chunk)): ~512 blocks -- they run next to the thousands
// of blocks of the activation split in the same launch, and every chunk is a partial row the bias fold has to read
static long planes_chunks(long grows, int cout) {
    const int coltiles = (cout + 127) / 128;
    long chunks = (512 + coltiles - 1) / coltiles;
    if (chunks > (grows + 31) / 32) chunks = (grows + 31) / 32;
    return chunks < 1 ? 1 : chunks;
}

// pre-passes + the planes form of the wave-specialised kernel.  scratch: [gh | gl | uh | ul], 2 bytes per element each
template <int PREC>
static void launch_wgrad_planes(WArgs& w, long grid, void* scratch, hipStream_t st) {
    typedef typename Split<PREC>::T T;
    const sgd_igemm_args& a = w.a;
    const int cin = a.c0 + a.c1;
    const bool pool = a.resample == SGD_RS_AVGPOOL2;
    const long grows = w.rows, urows = pool ? (long)a.n * (a.hi / 2) * (a.wi / 2) : (long)a.n * a.hi * a.wi;
    T* gh = reinterpret_cast<T*>(scratch);
    T* gl = gh + grows * w.cout;
    T* uh = gl + grows * w.cout;
    T* ul = uh + urows * cin;
    // enough row chunks to fill the chip (the K split of the main kernel can be as small as 2); the bias gradient's column
    // sums of the chunks go straight into the caller's bias rows -- sgd_wgrad_bias_rows() of them, folded by
    // sgd_wgrad_reduce_bias (round 5: the fold into row 0 was a launch of its own)
    const int coltiles = (w.cout + 127) / 128;
    const long chunks = planes_chunks(grows, w.cout);
    const long quads = urows * (cin / 4);
    long ablk = (quads + 255) / 256;
    if (ablk > 16384) ablk = 16384;
    PrepassArgs pp;
    pp.g = w.gy; pp.grows = grows; pp.gc = w.cout; pp.gld = w.gy_ld; pp.gh = gh; pp.gl = gl; pp.chunks = (int)chunks;
    pp.colsum = w.bslab; pp.coltiles = coltiles; pp.sr_blocks = (int)(coltiles * chunks);
    pp.a = a; pp.urows = urows; pp.rows_per_n = pool ? (a.hi / 2) * (a.wi / 2) : a.hi * a.wi; pp.uh = uh; pp.ul = ul;
    pp.pool = pool ? 1 : 0;
    hipLaunchKernelGGL((wgrad_prepass_kernel<PREC>), dim3((unsigned)(pp.sr_blocks + ablk)), dim3(256), 0, st, pp);
    w.gh = gh; w.gl = gl; w.uh = uh; w.ul = ul;
    if (pool) {                   // the main kernel sees the pooled planes as an unresampled input
        w.a.hi /= 2;
        w.a.wi /= 2;
        w.a.resample = SGD_RS_NONE;
    }
    launch_wgrad_ws<PREC, true>(w, grid, st);
}

template <int PREC, bool VEC, int TAPS, int FK = 0>
static void launch_wgrad_fast(const WArgs& w, long grid, hipStream_t st) {
    constexpr size_t smem = (2 * 64 * FGP + 2 * (TAPS == 9 ? 100 * 32 : 64 * FGP)) * 2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)wgrad_conv_kernel<PREC, VEC, TAPS, FK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
        attr = true;
    }
    hipLaunchKernelGGL((wgrad_conv_kernel<PREC, VEC, TAPS, FK>), dim3((unsigned)grid), dim3(256), smem, st, w);
}

// which specialised instance of the 1x1 / linear kernel serves the WHOLE launch (0: the general one): the pipelined staging's
// conditions (wgrad_conv_kernel, flat_pipe) with "this block's gradient columns are a whole 128-channel block" true for all blocks
static int wgrad_flat_kind(const sgd_igemm_args& a, const WArgs& w, bool vec) {
    if (!vec || !w.gvec || w.cout % WT != 0 || a.drop_p != 0.f || w.no_flat_pipe || w.rows <= 0) return 0;
    if (w.tap9) return (a.pro == SGD_PRO_NONE || (a.pro == SGD_PRO_AFFINE_NC && (a.ho * a.wo) % 64 == 0)) ? 2 : 0;
    if (a.pro == SGD_PRO_NONE) return 1;
    if (a.pro == SGD_PRO_AFFINE_NC) return (a.rows_per_n > 0 && a.rows_per_n % 64 == 0) ? 1 : 0;
    return a.pro == SGD_PRO_LN_ROW ? 3 : 0;
}
template <int PREC>
static void launch_wgrad_flat(const WArgs& w, long grid, hipStream_t st, bool vec) {
    switch (wgrad_flat_kind(w.a, w, vec)) {
    case 1: launch_wgrad_fast<PREC, true, 1, 1>(w, grid, st); break;
    case 2: launch_wgrad_fast<PREC, true, 1, 2>(w, grid, st); break;
    case 3: launch_wgrad_fast<PREC, true, 1, 3>(w, grid, st); break;
    default:
        if (vec) launch_wgrad_fast<PREC, true, 1>(w, grid, st);
        else launch_wgrad_fast<PREC, false, 1>(w, grid, st);
    }
}

// =============================================================================================
// weight gradients of the two convs with a handful of channels on one side (round 4): the stem (3 or 4 input channels,
// input_blocks.0.0, openaimodel.py:531-537) and the output head (3 output channels, out.2, :830-835).  2 GFLOP against
// 170 MB of gradient / activation rows each: HBM-bound, and ~0.4 ms apiece on the generic per-tap MFMA kernel.  Here a
// lane owns one channel of the WIDE side and walks a slab of rows; the narrow side of a pixel neighbourhood is a
// handful of wave-uniform scalars.  Pure fp32 FMA (every arithmetic mode), deterministic, same slab format as the
// other kernels: slabs[k][tap][co][ci], bslab[k][cout].
// =============================================================================================
// One kernel for both: per pixel r a lane holds v(r) -- gy[r][co] (stem) or act(x)[r][ci] (head) -- and the block shares
// the pixel's NARROW vector nv(r)[9 taps][NC] through LDS -- x[r + tap][ci] (stem: an im2col row) or gy[r - tap][co]
// (head: the 9 output pixels input pixel r feeds) -- zero outside the image: acc[t][c] += v * nv[t][c], 9 NC FMAs for one
// vector load and 9 NC / 4 broadcast LDS reads.  A block owns one slab of the rows and walks it in chunks of WN_ROWS.
constexpr int WN_ROWS = 128;
template <int NC, bool HEAD>
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(const float* __restrict__ x, const float* __restrict__ pa,
                                                           const float* __restrict__ pb, int silu, const float* __restrict__ gy,
                                                           int gy_ld, int n, int h, int w, int wide, int ksplit,
                                                           float* __restrict__ slabs, float* __restrict__ bslab) {
    constexpr int NV = 9 * NC;                                 // narrow values per pixel
    constexpr int NVP = (NV + 3) & ~3;                         // padded to 16 bytes
    __shared__ __attribute__((aligned(16))) float nvs[WN_ROWS * NVP];
    __shared__ float red[256 * (NV + 1)];
    const int lanes = wide < 256 ? wide : 256;                 // host: wide % 64 == 0, 256 % lanes == 0
    const int rsub = 256 / lanes;                              // row interleave inside a chunk
    const int ch = threadIdx.x % lanes;
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x / lanes);
    const long rows = (long)n * h * w;
    const long per = ((rows + ksplit - 1) / ksplit + WN_ROWS - 1) / WN_ROWS * WN_ROWS;     // rows of a slab: whole chunks
    const long s0 = (long)blockIdx.x * per, s1 = s0 + per < rows ? s0 + per : rows;
    // the narrow tensor: gy for the head (NC = cout channels per row), x for the stem (NC = cin)
    const float* nar = HEAD ? gy : x;
    const int nar_ld = HEAD ? gy_ld : NC;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    float accb = 0.f;
    for (long c0 = s0; c0 < s1; c0 += WN_ROWS) {
        __syncthreads();
        // narrow vectors of the chunk's pixels: (pixel, tap) items, NC values each
        for (int it = threadIdx.x; it < WN_ROWS * 9; it += 256) {
            const int pr = it / 9, t = it - pr * 9;
            const long r = c0 + pr;
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            float vals[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) vals[c] = 0.f;
            if (r < s1) {
                const int px = (int)(r % w), py = (int)((r / w) % h);
                const int yy = HEAD ? py - dy : py + dy, xx = HEAD ? px - dx : px + dx;
                if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                    const long rr = HEAD ? r - (long)dy * w - dx : r + (long)dy * w + dx;
#pragma unroll
                    for (int c = 0; c < NC; ++c) vals[c] = nar[rr * nar_ld + c];
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) nvs[pr * NVP + t * NC + c] = vals[c];
        }
        __syncthreads();
        const int nr = (int)(s1 - c0 < WN_ROWS ? s1 - c0 : WN_ROWS);
        // head: GroupNorm coefficients of the chunk's image once per chunk when a chunk cannot straddle two images
        const bool one_img = HEAD && pa && ((long)h * w) % WN_ROWS == 0;
        float pav = 1.f, pbv = 0.f;
        if (one_img) {
            const long img = c0 / ((long)h * w);
            pav = pa[img * wide + ch];
            pbv = pb[img * wide + ch];
        }
        for (int pb0 = sub; pb0 < nr; pb0 += 8 * rsub) {
            // eight rows of this lane requested before the first is used
            float vq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pr = pb0 + j * rsub;
                const long r = c0 + (pr < nr ? pr : nr - 1);
                vq[j] = HEAD ? x[r * wide + ch] : gy[r * gy_ld + ch];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pr = pb0 + j * rsub;
                if (pr >= nr) break;                                       // wave-uniform
                float v = vq[j];
                if (HEAD) {
                    if (one_img) {
                        v = v * pav + pbv;
                    } else if (pa) {
                        const long img = (c0 + pr) / ((long)h * w);
                        v = v * pa[img * wide + ch] + pb[img * wide + ch];
                    }
                    if (silu) v = sgd_silu(v);
                    if (ch < NC) accb += nvs[pr * NVP + 4 * NC + ch];    // centre tap = gy of this pixel: the bias gradient
                } else {
                    accb += v;
                }
                const f32x4* nq = reinterpret_cast<const f32x4*>(nvs + pr * NVP);
#pragma unroll
                for (int i4 = 0; i4 < NVP / 4; ++i4) {
                    const f32x4 q4 = nq[i4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (i4 * 4 + e < NV) acc[i4 * 4 + e] += v * q4[e];
                }
            }
        }
    }
    // the rsub row interleaves of the block are folded through LDS: ONE slab per block
    float* mine = red + (size_t)threadIdx.x * (NV + 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) mine[i] = acc[i];
    mine[NV] = accb;
    __syncthreads();
    if (threadIdx.x < lanes) {
        for (int s2 = 1; s2 < rsub; ++s2) {
            const float* o = red + (size_t)(s2 * lanes + ch) * (NV + 1);
#pragma unroll
            for (int i = 0; i < NV; ++i) acc[i] += o[i];
            accb += o[NV];
        }
        // slab layout [tap][co][ci]: stem wide = co (cin = NC), head wide = ci (cout = NC)
        float* sl = slabs + (size_t)blockIdx.x * NV * wide;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                sl[HEAD ? ((size_t)t * NC + c) * wide + ch : ((size_t)t * wide + ch) * NC + c] = acc[t * NC + c];
        if (bslab) {
            if (!HEAD) bslab[(size_t)blockIdx.x * wide + ch] = accb;
            else if (ch < NC) bslab[(size_t)blockIdx.x * NC + ch] = accb;
        }
    }
}

// bslab / dbias (optional): the bias gradient rides on the same launch -- its [ksplit][cout] partial column sums (written
// by the weight-gradient kernel) are folded by the first `cout` threads in the fixed order and double accumulation of
// colsum_stage2_kernel (round 4: 71 sgd_colsum_fold launches of ~6.5 us per training step)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int ksplit, int taps, int cout, int cin,
                                    float* __restrict__ dw, int accumulate, float scale,
                                    const float* __restrict__ bslab, float* __restrict__ dbias, int brows) {
    const long per = (long)taps * cout * cin;
    if (bslab && blockIdx.x * 32 < cout) {
        // block b folds columns 32 b .. 32 b + 31: 8 slab lanes x 32 columns, independent loads in flight, fixed order
        // (colsum_stage2_kernel's scheme; a serial loop over up to 512 slabs in one thread was a 30 us tail of the launch)
        const int col = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
        double t = 0;
        if (col < cout) {
            // eight partial rows of this lane in flight (the planes form writes a few hundred of them), added in index order
            int k = rl;
            for (; k + 56 < brows; k += 64) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = bslab[(long)(k + 8 * j) * cout + col];
#pragma unroll
                for (int j = 0; j < 8; ++j) t += v[j];
            }
            for (; k < brows; k += 8) t += bslab[(long)k * cout + col];
        }
        __shared__ double red[8][32];
        red[rl][threadIdx.x & 31] = t;
        __syncthreads();
        if (threadIdx.x < 32 && col < cout) {
            double u = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) u += red[k][threadIdx.x];
            u *= scale;
            dbias[col] = accumulate ? dbias[col] + (float)u : (float)u;
        }
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        // i indexes [tap][co][ci]
        const int ci = i % cin;
        const long t = i / cin;
        const int co = t % cout, tap = t / cout;
        // the slabs are summed in index order (deterministic); eight independent loads in flight per thread -- the plain
        // loop compiled to load -> s_waitcnt vmcnt(0) -> add per slab, one memory latency each
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= ksplit; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = slabs[(k + j) * per + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k < ksplit; ++k) s += slabs[k * per + i];
        const long o = ((long)co * cin + ci) * taps + tap;
        s *= scale;
        dw[o] = accumulate ? dw[o] + s : s;
    }
}

// column sums, deterministic two-stage: stage 1 = grid (32-column slab, row chunk), 8 row lanes x 32 columns
// per block, partial sums to work[chunk][c]; stage 2 folds the chunks (fixed order, double accumulation).
// With ONE row chunk (rows <= 256: the per-sample GroupNorm dgamma / dbeta tables, bias gradients of the embedding linears)
// the fold of stage 2 is the identity on one float, so stage 1 writes the scaled result itself: one launch instead of two,
// bit-identical (float row sums, then double * scale exactly as stage 2 does).
// blockIdx.z == 1 (sgd_colsum_pair): the second matrix / output of a pair with the same shape (GroupNorm dgamma + dbeta)
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const float* __restrict__ g, int rows, int c, int ld,
                                                            int chunks, float* __restrict__ work,
                                                            float* __restrict__ out, int accumulate, float scale,
                                                            const float* __restrict__ g2 = nullptr,
                                                            float* __restrict__ out2 = nullptr) {
    if (blockIdx.z == 1) { g = g2; out = out2; }
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;     // 8 row lanes
    const int chunk = blockIdx.y;
    const long per = ((long)rows + chunks - 1) / chunks;
    const long r0 = chunk * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    float s = 0.f;
    if (col < c) {
        long r = r0 + rl;
        for (; r + 56 < r1; r += 64) {                   // eight rows of this lane in flight (same order of additions)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = g[(r + 8 * j) * ld + col];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; r < r1; r += 8) s += g[r * ld + col];
    }
    __shared__ float red[8][32];
    red[rl][threadIdx.x & 31] = s;
    __syncthreads();
    if (threadIdx.x < 32 && col < c) {
        float t = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
        if (out) {
            double u = t;
            u *= scale;
            out[col] = accumulate ? out[col] + (float)u : (float)u;
        } else {
            work[(long)chunk * c + col] = t;
        }
    }
}

// stage 2: 8 chunk lanes x 32 columns per block (independent loads in flight), fixed-order double accumulation
__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ work, int chunks, int c,
                                                            float* __restrict__ out, int accumulate, float scale) {
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
    double t = 0;
    if (col < c) {
#pragma unroll 4
        for (int k = rl; k < chunks; k += 8) t += work[(long)k * c + col];
    }
    __shared__ double red[8][32];
    red[rl][threadIdx.x & 31] = t;
    __syncthreads();
    if (threadIdx.x < 32 && col < c) {
        double u = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) u += red[k][threadIdx.x];
        u *= scale;
        out[col] = accumulate ? out[col] + (float)u : (float)u;
    }
}

// =============================================================================================
// GroupNorm(+FiLM)+SiLU backward
// =============================================================================================
// gradient arriving at the activated tensor for x-pixel (n, y, x), channels cq..cq+3 of the gu tensor
__device__ __forceinline__ f32x4 fetch_g(const float* g, int ld, int mode, int n, int y, int x, int h, int w, int cq) {
    if (mode == SGD_RS_NONE) return ld4(g + (((long)n * h + y) * w + x) * ld + cq);
    if (mode == SGD_RS_AVGPOOL2)       // forward pooled 2x2: each input pixel gets a quarter of the pooled gradient
        return 0.25f * ld4(g + (((long)n * (h >> 1) + (y >> 1)) * (w >> 1) + (x >> 1)) * ld + cq);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};    // forward upsampled x2: sum over the 2x2 block that copied this pixel
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) s += ld4(g + (((long)n * 2 * h + 2 * y + dy) * (2 * w) + 2 * x + dx) * ld + cq);
    return s;
}

// gradient through the train-time dropout that sits between the activation and the consumer conv
__device__ __forceinline__ f32x4 drop_mask(f32x4 g, float p, uint32_t seed, long base) { return sgd_drop4(g, p, seed, base); }

// block per (n, 32-channel slab): 8 channel quads x 32 row lanes
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x, int h, int w, int c,
                                                            int c_total, int c_off, const float* __restrict__ a,
                                                            const float* __restrict__ b, int silu,
                                                            const float* __restrict__ gu, int gu_ld, int gu_mode,
                                                            float drop_p, uint32_t drop_seed,
                                                            float* __restrict__ S) {
    const int slabs = (c + 31) / 32;
    const int n = blockIdx.x / slabs, slab = blockIdx.x % slabs;
    const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int ch = slab * 32 + q * 4;
    const int hw = h * w;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (ch < c) {
        const f32x4 av = ld4(a + (long)n * c_total + c_off + ch), bv = ld4(b + (long)n * c_total + c_off + ch);
        auto fold = [&](const f32x4& xv, f32x4 gv, int p) {
            if (drop_p > 0.f) gv = drop_mask(gv, drop_p, drop_seed, ((long)n * hw + p) * c_total + c_off + ch);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float gp = gv[j];
                if (silu) gp *= dsilu(av[j] * xv[j] + bv[j]);
                s1[j] += gp;
                s2[j] += gp * xv[j];
            }
        };
        int p = rl;
        if (gu_mode == SGD_RS_NONE) {
            // same-resolution gradient (every launch but the resampling blocks): four pixels of this lane in flight -- the
            // plain loop was one load pair -> s_waitcnt vmcnt(0) per pixel; the additions keep their order
            for (; p + 96 < hw; p += 128) {
                f32x4 xv[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = ld4(x + ((long)n * hw + p + 32 * u) * c + ch);
                    gv[u] = ld4(gu + ((long)n * hw + p + 32 * u) * gu_ld + c_off + ch);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) fold(xv[u], gv[u], p + 32 * u);
            }
        }
        for (; p < hw; p += 32) {
            const f32x4 xv = ld4(x + ((long)n * hw + p) * c + ch);
            const f32x4 gv = fetch_g(gu, gu_ld, gu_mode, n, p / w, p % w, h, w, c_off + ch);
            fold(xv, gv, p);
        }
    }
    __shared__ double red[32][8][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[rl][q][j] = s1[j]; red[rl][q][4 + j] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int qq = threadIdx.x >> 3, jj = threadIdx.x & 7;
        double t = 0;
        for (int r = 0; r < 32; ++r) t += red[r][qq][jj];
        const int cc = slab * 32 + qq * 4 + (jj & 3);
        if (cc < c) S[((long)n * c_total + c_off + cc) * 2 + (jj >> 2)] = (float)t;
    }
}

// S of the GroupNorm-backward coefficient kernels: [n][c][2] (chunks = 1, sgd_gn_bwd_reduce) or the per-chunk partial sums
// [n][chunks <= 16][c][2] of sgd_gn_bwd_reduce_rows, folded here -- no launch of its own for the fold: the partials of one
// (image, channel) are requested together (sixteen independent loads; a `t += S[..]` loop is one memory latency per chunk), added
// in chunk order in double and rounded to float ONCE, as a separate fold kernel writing S[n][c][2] would have done
__device__ __forceinline__ float s_fold(const float* __restrict__ S, int chunks, int c, int nn, int cc, int which) {
    if (chunks == 1) return S[((long)nn * c + cc) * 2 + which];
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = S[(((long)nn * chunks + (k < chunks ? k : chunks - 1)) * c + cc) * 2 + which];
    double t = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += k < chunks ? (double)v[k] : 0.0;
    return (float)t;
}

__global__ void gn_bwd_coef_kernel(const float* __restrict__ S, int chunks, const float* __restrict__ sums,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ film, int film_ld, int n, int c, int groups, int hw,
                                   float eps, float* __restrict__ A, float* __restrict__ B, float* __restrict__ Cc,
                                   float* __restrict__ dg_nc, float* __restrict__ db_nc, float* __restrict__ dfilm) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)n * c) return;
    const int nn = i / c, cc = i % c;
    const int cpg = c / groups, g0 = (cc / cpg) * cpg;
    double s = 0, ss = 0;
    for (int k = 0; k < cpg; ++k) {
        s += sums[((long)nn * c + g0 + k) * 2];
        ss += sums[((long)nn * c + g0 + k) * 2 + 1];
    }
    const double m = (double)cpg * hw;
    const double mean = s / m;
    double var = ss / m - mean * mean;
    if (var < 0) var = 0;
    const double r = 1.0 / sqrt(var + (double)eps);
    // group means of dxhat and dxhat*xhat
    double m1 = 0, m2 = 0;
    for (int k = 0; k < cpg; ++k) {
        const long j = (long)nn * c + g0 + k;
        const double sc = film ? 1.0 + film[(long)nn * film_ld + g0 + k] : 1.0;
        const double gp = gamma[g0 + k] * sc;
        const double S1 = s_fold(S, chunks, c, nn, g0 + k, 0), X = r * (s_fold(S, chunks, c, nn, g0 + k, 1) - mean * S1);
        m1 += gp * S1;
        m2 += gp * X;
        (void)j;
    }
    m1 /= m;
    m2 /= m;
    const double sc = film ? 1.0 + film[(long)nn * film_ld + cc] : 1.0;
    const double S1 = s_fold(S, chunks, c, nn, cc, 0), X = r * (s_fold(S, chunks, c, nn, cc, 1) - mean * S1);
    A[i] = (float)(r * gamma[cc] * sc);
    B[i] = (float)(-r * r * m2);
    Cc[i] = (float)(r * r * m2 * mean - r * m1);
    dg_nc[i] = (float)(X * sc);
    db_nc[i] = (float)(S1 * sc);
    if (dfilm) {
        dfilm[(long)nn * film_ld + cc] = (float)(X * gamma[cc] + S1 * beta[cc]);      // d/d scale
        dfilm[(long)nn * film_ld + c + cc] = (float)S1;                               // d/d shift
    }
}

// gn_bwd_coef_kernel + the dgamma / dbeta column sums (sgd_colsum_pair) in ONE launch (round 5: 49 launches fewer per
// training step).  One block per GROUP: thread (nn, k) forms the coefficients of channel g0 + k of image nn exactly as
// gn_bwd_coef_kernel does, leaves its dgamma / dbeta contributions in LDS tables [n][cpg], and the block then sums the
// tables' columns over the images in colsum_stage1_kernel's order (8 row lanes, rows rl, rl + 8, .. in float; the lanes
// folded in float; double * scale): bit-identical to the two-launch route.
__global__ __launch_bounds__(256) void gn_bwd_coef_fold_kernel(const float* __restrict__ S, int chunks, const float* __restrict__ sums,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ film, int film_ld, int n, int c,
                                                               int groups, int hw, float eps, float* __restrict__ A,
                                                               float* __restrict__ B, float* __restrict__ Cc,
                                                               float* __restrict__ dfilm, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int accumulate, float scale) {
    extern __shared__ float tab[];                    // dg[n][cpg] | db[n][cpg] | per image: mean, r, m1, m2 (doubles)
    const int cpg = c / groups, g0 = blockIdx.x * cpg;
    float* const tdg = tab;
    float* const tdb = tab + (size_t)n * cpg;
    double* const gst = reinterpret_cast<double*>(tab + (size_t)2 * n * cpg + ((2 * n * cpg) & 1));   // [n][4], 8-byte aligned
    // chunked S (sgd_gn_bwd_reduce_rows): the group's folded sums once, by all threads, into LDS [n][cpg][2] behind the other
    // tables -- the thread-per-image loop below would otherwise walk cpg x 2 x chunks dependent global loads
    float* const sl = reinterpret_cast<float*>(gst + (size_t)n * 4);
    if (chunks > 1) {
        for (int it = threadIdx.x; it < n * cpg * 2; it += blockDim.x) {
            const int nn = it / (cpg * 2), r = it - nn * (cpg * 2);
            sl[it] = s_fold(S, chunks, c, nn, g0 + (r >> 1), r & 1);
        }
        __syncthreads();
    }
    auto sget = [&](int nn, int kk, int which) -> double {
        return chunks > 1 ? (double)sl[(nn * cpg + kk) * 2 + which] : (double)S[((long)nn * c + g0 + kk) * 2 + which];
    };
    // the group quantities ONCE per image (gn_bwd_coef_kernel recomputes them in every one of the group's cpg threads -- fine
    // over n * c / 256 blocks, not inside the 32 blocks of this launch): same additions in the same order, so the same bits
    for (int nn = threadIdx.x; nn < n; nn += blockDim.x) {
        double s = 0, ss = 0;
        for (int k = 0; k < cpg; ++k) {
            s += sums[((long)nn * c + g0 + k) * 2];
            ss += sums[((long)nn * c + g0 + k) * 2 + 1];
        }
        const double m = (double)cpg * hw;
        const double mean = s / m;
        double var = ss / m - mean * mean;
        if (var < 0) var = 0;
        const double r = 1.0 / sqrt(var + (double)eps);
        double m1 = 0, m2 = 0;
        for (int k = 0; k < cpg; ++k) {
            const long j = (long)nn * c + g0 + k;
            const double sc = film ? 1.0 + film[(long)nn * film_ld + g0 + k] : 1.0;
            const double gp = gamma[g0 + k] * sc;
            const double S1 = sget(nn, k, 0), X = r * (sget(nn, k, 1) - mean * S1);
            m1 += gp * S1;
            m2 += gp * X;
            (void)j;
        }
        m1 /= m;
        m2 /= m;
        gst[nn * 4] = mean; gst[nn * 4 + 1] = r; gst[nn * 4 + 2] = m1; gst[nn * 4 + 3] = m2;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < n * cpg; it += blockDim.x) {
        const int nn = it / cpg, kk = it - nn * cpg, cc = g0 + kk;
        const long i = (long)nn * c + cc;
        const double mean = gst[nn * 4], r = gst[nn * 4 + 1], m1 = gst[nn * 4 + 2], m2 = gst[nn * 4 + 3];
        const double sc = film ? 1.0 + film[(long)nn * film_ld + cc] : 1.0;
        const double S1 = sget(nn, kk, 0), X = r * (sget(nn, kk, 1) - mean * S1);
        A[i] = (float)(r * gamma[cc] * sc);
        B[i] = (float)(-r * r * m2);
        Cc[i] = (float)(r * r * m2 * mean - r * m1);
        tdg[it] = (float)(X * sc);
        tdb[it] = (float)(S1 * sc);
        if (dfilm) {
            dfilm[(long)nn * film_ld + cc] = (float)(X * gamma[cc] + S1 * beta[cc]);      // d/d scale
            dfilm[(long)nn * film_ld + c + cc] = (float)S1;                               // d/d shift
        }
    }
    __syncthreads();
    // column sums over the images: thread = (table, column kk, row lane rl)
    __shared__ float red[2][8][32];
    const int t2 = threadIdx.x >> 7, rem = threadIdx.x & 127;          // 2 tables x (up to 16 columns x 8 lanes) per pass
    for (int k0 = 0; k0 < cpg; k0 += 16) {
        const int kk = k0 + (rem & 15), rl = rem >> 4;
        float sacc = 0.f;
        if (kk < cpg) {
            const float* tp = t2 ? tdb : tdg;
            int r = rl;
            for (; r + 56 < n; r += 64) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tp[(r + 8 * j) * cpg + kk];
#pragma unroll
                for (int j = 0; j < 8; ++j) sacc += v[j];
            }
            for (; r < n; r += 8) sacc += tp[r * cpg + kk];
        }
        red[t2][rl][rem & 15] = sacc;
        __syncthreads();
        if (rl == 0 && kk < cpg) {
            float t = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[t2][k][rem & 15];
            double u = t;
            u *= scale;
            float* out = t2 ? dbeta : dgamma;
            out[g0 + kk] = accumulate ? out[g0 + kk] + (float)u : (float)u;
        }
        __syncthreads();
    }
}

__global__ void gn_bwd_apply_kernel(const float* __restrict__ x, int n, int h, int w, int c, int c_total, int c_off,
                                    const float* __restrict__ a, const float* __restrict__ b, int silu,
                                    const float* __restrict__ gu, int gu_ld, int gu_mode, float drop_p,
                                    uint32_t drop_seed,
                                    const float* __restrict__ A, const float* __restrict__ B,
                                    const float* __restrict__ Cc, const float* __restrict__ gres, int gres_ld,
                                    int gres_mode, float* __restrict__ dst, int dst_ld, int dst_off, int accumulate) {
    const int cq = c >> 2;
    const long total = (long)n * h * w * cq;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cq) * 4;
        const long row = i / cq;
        const int p = row % (h * w), nn = row / (h * w);
        const int y = p / w, xx = p % w;
        const long ci = (long)nn * c_total + c_off + ch;
        const f32x4 xv = ld4(x + row * c + ch);
        f32x4 gv = fetch_g(gu, gu_ld, gu_mode, nn, y, xx, h, w, c_off + ch);
        if (drop_p > 0.f) gv = drop_mask(gv, drop_p, drop_seed, row * c_total + c_off + ch);
        if (silu) {
            const f32x4 av = ld4(a + ci), bv = ld4(b + ci);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] *= dsilu(av[j] * xv[j] + bv[j]);
        }
        f32x4 o = ld4(A + ci) * gv + ld4(B + ci) * xv + ld4(Cc + ci);
        if (gres) o += fetch_g(gres, gres_ld, gres_mode, nn, y, xx, h, w, c_off + ch);
        float* dp = dst + row * dst_ld + dst_off + ch;
        if (accumulate) o += ld4(dp);
        *reinterpret_cast<f32x4*>(dp) = o;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the REDUCE pass of the GroupNorm backward as a row stream.  The round-1 kernel above reads the tensors in 128-byte
// segments (a block per (image, 32-channel slab): eight pixel rows of 128 bytes per wave instruction, four blocks sharing every
// 512-byte row of a 128-channel map): 4.5-5.1 TB/s, 3.9 with the dropout hash.  tools/hbm_probe_sweep.py says what the memory
// system wants (profiles/r6_hbm_probe_sweep.txt): every wave walking CONTIGUOUS 8 KiB pieces -- eight consecutive 1 KiB wave
// instructions in flight -- reads at 6.6 TB/s, against 4.7 for the same bytes with a lane's loads megabytes apart.  So: a block =
// one image chunk, all channels; a wave = 8 KiB pieces of the chunk's rows; a lane's channel quad is fixed (c <= 256) or cycles
// through NSET = c / 256 sets with the piece's instructions, so the per-(image, channel) coefficients are loaded once per block;
// per-chunk partial sums, folded by the coefficient launch (s_fold).  tools/bench_gn_bwd.py, UNet batch 80 (profiles/
// r6_gn_bwd_rows.txt): 5.4-5.9 TB/s on the 64x64 and 32x32 maps (+13-20 %; with dropout 5.5 vs 3.9), SLOWER on 16x16 maps (one or
// two windows per block): sgd_gn_bwd_rows_chunks serves h * w >= 1024 only.  The APPLY pass as the same stream (eight quads of x,
// gradient, residual gradient in flight per lane, non-temporal loads) measured 3.1-3.5 TB/s against the round-1 kernel's 5.5-6.0
// (one quad per thread, 40,960 blocks): not shipped -- that pass already runs at the copy rate of this memory system.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GR_U = 8;                              // 1 KiB wave instructions per piece
constexpr int GR_WIN = 4 * GR_U * 64;                // quads of one block window (4 waves x 8 KiB)

template <int NSET>
__global__ __launch_bounds__(256) void gn_bwd_reduce_rows_kernel(const float* __restrict__ x, int hw, int c, int c_total, int c_off,
                                                                 const float* __restrict__ a, const float* __restrict__ b, int silu,
                                                                 const float* __restrict__ gu, int gu_ld, float drop_p,
                                                                 uint32_t drop_seed, int chunks, float* __restrict__ P) {
    const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
    const int cq = c >> 2, cq_l2 = 31 - __builtin_clz(cq);          // c is a power of two (sgd_gn_bwd_rows_chunks)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long chunk_quads = (long)hw * cq / chunks;             // a multiple of GR_WIN
    const long q0 = (long)n * hw * cq + chunk * chunk_quads;     // first quad of the chunk in x
    f32x4 av[NSET], bv[NSET];
    float s1[NSET][4], s2[NSET][4];
#pragma unroll
    for (int sx = 0; sx < NSET; ++sx) {
        const int ch = ((sx * 64 + lane) % cq) * 4;
        av[sx] = ld4(a + (long)n * c_total + c_off + ch);
        bv[sx] = ld4(b + (long)n * c_total + c_off + ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[sx][j] = s2[sx][j] = 0.f;
    }
    for (long wq = 0; wq < chunk_quads; wq += GR_WIN) {
        const long base = q0 + wq + (long)wave * (GR_U * 64) + lane;       // this lane's quad of instruction u: base + 64 u
        f32x4 xv[GR_U], gv[GR_U];
#pragma unroll
        for (int u = 0; u < GR_U; ++u) {
            const long q = base + 64 * u;
            const long row = q >> cq_l2;
            const int ch = (int)(q & (cq - 1)) * 4;
            xv[u] = ld4(x + q * 4);
            gv[u] = ld4(gu + row * gu_ld + c_off + ch);
        }
#pragma unroll
        for (int u = 0; u < GR_U; ++u) {
            const int sx = NSET == 1 ? 0 : (u % NSET);           // (wave * GR_U * 64 is a multiple of 256 quads: the set follows u)
            f32x4 g = gv[u];
            if (drop_p > 0.f) {
                const long q = base + 64 * u;
                g = sgd_drop4(g, drop_p, drop_seed, (q >> cq_l2) * c_total + c_off + (int)(q & (cq - 1)) * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float gp = g[j];
                if (silu) gp *= dsilu(av[sx][j] * xv[u][j] + bv[sx][j]);
                s1[sx][j] += gp;
                s2[sx][j] += gp * xv[u][j];
            }
        }
    }
    // fold the threads that share a channel quad (fixed order, double): thread t, set sx holds quad (sx * 64 + lane) % cq
    __shared__ float red[256][NSET][8];
#pragma unroll
    for (int sx = 0; sx < NSET; ++sx)
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[threadIdx.x][sx][j] = s1[sx][j]; red[threadIdx.x][sx][4 + j] = s2[sx][j]; }
    __syncthreads();
    const int cl = cq < 64 ? cq : 64;                              // distinct quads per (wave, set)
    for (int o = threadIdx.x; o < cq * 8; o += 256) {
        const int qd = o >> 3, j = o & 7;
        const int sx = qd / 64, l0 = qd % 64;                      // cq < 64: sx = 0, lanes l0, l0 + cq, ..
        double t = 0;
        for (int w = 0; w < 4; ++w)
            for (int l = l0; l < 64; l += cl) t += red[w * 64 + l][sx][j];
        P[(((long)n * chunks + chunk) * c_total + c_off + qd * 4 + (j & 3)) * 2 + (j >> 2)] = (float)t;
    }
}

// adjoint of the 2x resamplers applied to a gradient map (Upsample / avg-pool Downsample layers):
//   mode UP2 (forward nearest-upsampled): dst[n,y,x,:] (+)= sum of the 2x2 block of g (g at 2x resolution)
//   mode AVGPOOL2 (forward pooled)      : dst[n,y,x,:] (+)= g[n,y/2,x/2,:] / 4      (g at 1/2 resolution)
__global__ void resample_bwd_kernel(const float* __restrict__ g, int n, int h, int w, int c, int mode,
                                    float* __restrict__ dst, int accumulate) {
    const int cq = c >> 2;
    const long total = (long)n * h * w * cq;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cq) * 4;
        const long row = i / cq;
        const int p = row % (h * w), nn = row / (h * w);
        f32x4 o = fetch_g(g, c, mode, nn, p / w, p % w, h, w, ch);
        float* dp = dst + row * c + ch;
        if (accumulate) o += ld4(dp);
        *reinterpret_cast<f32x4*>(dp) = o;
    }
}

__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, long count,
                                float* __restrict__ gx) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < count) gx[i] = g[i] * dsilu(x[i]);
}

__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                const int64_t* __restrict__ t, const float* __restrict__ sa,
                                const float* __restrict__ s1, int b, long chw, float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)b * chw) return;
    const int n = i / chw;
    const int64_t tt = t[n];
    out[i] = sa[tt] * x0[i] + s1[tt] * noise[i];              // ddpm_sampler.py:116-119
}

// one block per sample: per-sample mean of squared error + gradient wrt eps (NHWC)
__global__ __launch_bounds__(256) void mse_loss_kernel(const float* __restrict__ eps, const float* __restrict__ noise,
                                                       int b, int c, int hw, float* __restrict__ per_sample,
                                                       float* __restrict__ geps) {
    const int n = blockIdx.x;
    const int chw = c * hw;
    const float gscale = -2.0f / ((float)chw * (float)b);     // d mean_b(mean_chw((noise-eps)^2)) / d eps
    double s = 0;
    for (int i = threadIdx.x; i < chw; i += 256) {
        const int p = i / c, cc = i % c;                      // NHWC walk
        const float d = noise[((long)n * c + cc) * hw + p] - eps[(long)n * chw + i];
        s += (double)d * d;
        if (geps) geps[(long)n * chw + i] = gscale * d;
    }
    s = wave_sum_d(s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) per_sample[n] = (float)((red[0] + red[1] + red[2] + red[3]) / chw);
}

inline unsigned nblk(long total, int cap = 1 << 20) {
    long b = (total + 255) / 256;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace

// which kernel serves a weight-gradient launch: shared by wgrad_impl and sgd_wgrad_bias_rows (the caller sizes the bias rows)
static bool wgrad_is_narrow(const sgd_igemm_args& a, int cout, long rows, int ksplit) {
    // stem / head: a handful of channels on one side, a whole number of waves of lanes on the other (see the kernels)
    const int cin = a.c0 + a.c1;
    if (!(a.mode == SGD_MODE_CONV3 && a.stride == 1 && a.resample == SGD_RS_NONE && a.c1 == 0 && a.drop_p == 0.f
          && ksplit <= (rows + 63) / 64 && !(a.tune & SGD_TUNE_WGRAD_GENERIC_NARROW)))
        return false;
    const bool lanes_co = cout % 64 == 0 && (cout == 256 || (cout < 256 && 256 % cout == 0));
    const bool lanes_ci = cin % 64 == 0 && (cin == 256 || (cin < 256 && 256 % cin == 0));
    if (cin <= 4 && cin >= 3 && a.pro == SGD_PRO_NONE && !a.pro_silu && lanes_co) return true;
    return cout == 3 && (a.pro == SGD_PRO_NONE || a.pro == SGD_PRO_AFFINE_NC) && lanes_ci;
}
static bool wgrad_fast_conv(const sgd_igemm_args& a) {
    return a.mode == SGD_MODE_CONV3 && a.stride == 1 && a.resample != SGD_RS_ZEROUP2 && a.ho > 0 && a.wo > 0 && a.ho % 8 == 0
           && a.wo % 8 == 0;
}
static bool wgrad_ws_ok(const sgd_igemm_args& a, int cout, int gy_ld) {          // wave-specialised kernel (without planes: no pool)
    const bool vec = (a.c0 % 4 == 0) && (a.c1 % 4 == 0);
    const bool pooled = a.resample == SGD_RS_AVGPOOL2 && !(a.tune & SGD_TUNE_WGRAD_NO_POOLED_PLANES);
    return wgrad_fast_conv(a) && a.prec != SGD_PREC_F32 && !(a.tune & SGD_TUNE_WGRAD_F32) && vec && gy_ld % 4 == 0 && cout % WT == 0
           && (a.resample == SGD_RS_NONE || a.resample == SGD_RS_UP2 || pooled)
           && (a.pro == SGD_PRO_NONE || a.pro == SGD_PRO_AFFINE_NC) && !(a.tune & SGD_TUNE_WGRAD_NO_WS);
}
static int64_t wgrad_planes_need(const sgd_igemm_args& a, int cout) {
    const int64_t rows = (int64_t)a.n * a.ho * a.wo;
    return 4 * (rows * cout + (int64_t)a.n * a.hi * a.wi * (a.c0 + a.c1));
}
static bool wgrad_planes_ok(const sgd_igemm_args& a, int cout, int gy_ld, bool have_scratch, int64_t scratch_bytes) {
    if (!(wgrad_ws_ok(a, cout, gy_ld) && have_scratch && scratch_bytes >= wgrad_planes_need(a, cout) && !(a.tune & SGD_TUNE_WGRAD_NO_PLANES)))
        return false;
    // The pre-pass moves 8 bytes per element of both operands through HBM -- rows x (cin + cout) -- against rows x cin x cout x 9
    // products: its share of the launch goes as 1 / cin + 1 / cout.  On the narrow layers (the 64x64 levels of the UNet, 128
    // output channels) it costs more than the loaders' in-kernel transform does.  Measured round 5 (C2 batch 80, per launch,
    // planes -> in-kernel split; tools/profile_train_layers.py with SGDM_WGRAD_TUNE): 128 -> 128 @64^2 0.328 -> 0.267 ms (with
    // dropout 0.329 -> 0.305), 256 -> 128 0.608 -> 0.524, 384 -> 128 0.886 -> 0.772, 128 -> 256 @32^2 0.158 -> 0.139; from
    // 256 -> 256 on the planes win (0.275 vs 0.31; 512 -> 512 @16^2 0.243 vs 0.31).  A third form -- planes for the gradient rows
    // only, which cin / 32 blocks stage each -- was built and lost to both (0.38 ms on 128 -> 128).  The pooled form
    // (ResBlock(down)) exists only with planes.
    const long cin = a.c0 + a.c1;
    return cin * cout > 100 * (cin + cout) || a.resample == SGD_RS_AVGPOOL2 || (a.tune & SGD_TUNE_WGRAD_PLANES_ALWAYS);
}

static int wgrad_impl(const sgd_igemm_args* fwd, const float* gy, int32_t gy_ld, int32_t cout, float* slabs,
                      int32_t ksplit, float* bias_slabs, void* scratch, int64_t scratch_bytes, void* stream) {
    SGD_CLEAR_ERR();
    if (!fwd || !gy || !slabs || cout <= 0 || ksplit <= 0 || gy_ld < cout) return SGD_ERR_ARG;
    WArgs w;
    w.a = *fwd;
    w.no_flat_pipe = (fwd->tune & SGD_TUNE_WGRAD_NO_PIPE) ? 1 : 0;
    w.tap9 = 0;
    const sgd_igemm_args& a = w.a;
    if (!a.x0 || a.c0 <= 0 || a.c1 < 0 || (a.c1 > 0 && !a.x1)) return SGD_ERR_ARG;
    if (a.pro != SGD_PRO_NONE && (!a.pa || !a.pb)) return SGD_ERR_ARG;
    const int cin = a.c0 + a.c1;
    if (a.mode == SGD_MODE_CONV3) {
        if (a.stride != 1 && a.stride != 2) return SGD_ERR_ARG;
        w.hc = a.resample == SGD_RS_AVGPOOL2 ? a.hi / 2 : (a.resample == SGD_RS_UP2 ? a.hi * 2 : a.hi);
        w.wc = a.resample == SGD_RS_AVGPOOL2 ? a.wi / 2 : (a.resample == SGD_RS_UP2 ? a.wi * 2 : a.wi);
        w.taps = 9;
        w.rows = a.n * a.ho * a.wo;
        if (a.ho <= 0 || a.wo <= 0 || (a.ho & (a.ho - 1)) || (a.wo & (a.wo - 1))) return SGD_ERR_ARG;
        w.wo_l2 = __builtin_ctz(a.wo);
        w.ho_l2 = __builtin_ctz(a.ho);
    } else if (a.mode == SGD_MODE_FLAT) {
        w.hc = w.wc = 1;
        w.wo_l2 = w.ho_l2 = 0;
        w.taps = 1;
        w.rows = a.m;
    } else {
        return SGD_ERR_ARG;
    }
    if (w.rows <= 0) return SGD_ERR_ARG;
    if (wgrad_is_narrow(a, cout, w.rows, ksplit)) {
        hipStream_t st0 = (hipStream_t)stream;
        if (cin <= 4) {
            if (cin == 3) hipLaunchKernelGGL((wgrad_narrow_kernel<3, false>), dim3(ksplit), dim3(256), 0, st0, a.x0, (const float*)nullptr, (const float*)nullptr, 0, gy, gy_ld, a.n, a.ho, a.wo, cout, ksplit, slabs, bias_slabs);
            else hipLaunchKernelGGL((wgrad_narrow_kernel<4, false>), dim3(ksplit), dim3(256), 0, st0, a.x0, (const float*)nullptr, (const float*)nullptr, 0, gy, gy_ld, a.n, a.ho, a.wo, cout, ksplit, slabs, bias_slabs);
            return sgd_check_launch();
        }
        hipLaunchKernelGGL((wgrad_narrow_kernel<3, true>), dim3(ksplit), dim3(256), 0, st0, a.x0, a.pro == SGD_PRO_AFFINE_NC ? a.pa : nullptr,
                           a.pro == SGD_PRO_AFFINE_NC ? a.pb : nullptr, a.pro_silu, gy, gy_ld, a.n, a.ho, a.wo, cin, ksplit, slabs,
                           bias_slabs);
        return sgd_check_launch();
    }
    w.gy = gy; w.gy_ld = gy_ld; w.cout = cout; w.slabs = slabs; w.bslab = bias_slabs;
    w.co_tiles = (cout + WT - 1) / WT;
    w.ci_tiles = (cin + WT - 1) / WT;
    w.ktiles = (w.rows + WK - 1) / WK;
    w.ksplit = ksplit > w.ktiles ? w.ktiles : ksplit;
    if (w.ksplit != ksplit) return SGD_ERR_ARG;          // caller sizes the slabs: must agree
    const bool vec = (a.c0 % 4 == 0) && (a.c1 % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    // split-precision kernels: 3x3 stride 1 on whole 8x8 output patches (all taps per block), and 1x1 / linear
    const bool fast_conv = wgrad_fast_conv(a);
    // (round 5: the LayerNorm-row prologue -- Attention_LR's to_q / to_kv, crossattetion_lr.py:81-88 -- runs on the split kernel
    // too: its staging takes the per-row statistics through load_coef like every other prologue; those six to_q launches of C5
    // ran on the exact-f32 kernel at 56 TF, 0.19 ms each)
    const bool fast_flat = a.mode == SGD_MODE_FLAT && cout >= 32 && cin >= 32;
    // Strided 3x3 convs (Downsample, openaimodel_ca.py:167-174) in a split mode: nine 1x1-style weight gradients, one tap per
    // block of the 1x1 / linear split kernel, whose input rows are the tap's (strided, shifted) pixels -- round 5: these two
    // launches of C5 / C4 ran on the exact-f32 per-tap kernel below at 43-45 TF (0.55 ms each)
    if (a.mode == SGD_MODE_CONV3 && a.stride == 2 && a.resample == SGD_RS_NONE && a.prec != SGD_PREC_F32 && vec && gy_ld % 4 == 0
        && (a.pro == SGD_PRO_NONE || a.pro == SGD_PRO_AFFINE_NC) && a.drop_p == 0.f && cout >= 32 && cin >= 32
        && !(a.tune & (SGD_TUNE_WGRAD_F32 | SGD_TUNE_WGRAD_GENERIC_NARROW))) {
        w.tap9 = 1;
        w.gvec = 1;
        const long fgrid = 9L * w.co_tiles * w.ci_tiles * w.ksplit;
        if (fgrid > 0x7fffffffL) return SGD_ERR_ARG;
        if (a.prec == SGD_PREC_F16X3) launch_wgrad_flat<SGD_PREC_F16X3>(w, fgrid, st, true);
        else launch_wgrad_flat<SGD_PREC_BF16X3>(w, fgrid, st, true);
        return sgd_check_launch();
    }
    if ((fast_conv || fast_flat) && a.prec != SGD_PREC_F32 && !(a.tune & SGD_TUNE_WGRAD_F32)) {
        w.gvec = gy_ld % 4 == 0;
        if (fast_conv) w.ci_tiles = (cin + 31) / 32;
        const long fgrid = (long)w.co_tiles * w.ci_tiles * w.ksplit;
        if (fgrid > 0x7fffffffL) return SGD_ERR_ARG;
        // wave-specialised kernel: 16-byte gradient rows of whole 128-channel blocks, whole rows (cout % 128 == 0), vector
        // input rows, GroupNorm-affine / no prologue, no avg-pool.  (Dropout does not exclude it: with pre-split planes
        // the pre-pass applies the keep mask through apply_pro, and the in-kernel loader does the same.)
        // the fused average pool (ResBlock(down)) only through the planes: the pre-pass writes them pooled
        const bool pooled = a.resample == SGD_RS_AVGPOOL2 && !(a.tune & SGD_TUNE_WGRAD_NO_POOLED_PLANES);
        const bool ws0 = wgrad_ws_ok(a, cout, gy_ld);
        // ... and with a scratch buffer for the pre-split operand planes: the loaders only copy
        const bool planes = wgrad_planes_ok(a, cout, gy_ld, scratch != nullptr, scratch_bytes);
        const bool ws = ws0 && (planes || !pooled);
#define SGD_WG(P, V)                                                             \
        do { if (planes) launch_wgrad_planes<P>(w, fgrid, scratch, st);           \
             else if (ws) launch_wgrad_ws<P, false>(w, fgrid, st);               \
             else if (fast_conv) launch_wgrad_fast<P, V, 9>(w, fgrid, st); else launch_wgrad_flat<P>(w, fgrid, st, V); } while (0)
        if (a.prec == SGD_PREC_F16X3) { if (vec) SGD_WG(SGD_PREC_F16X3, true); else SGD_WG(SGD_PREC_F16X3, false); }
        else { if (vec) SGD_WG(SGD_PREC_BF16X3, true); else SGD_WG(SGD_PREC_BF16X3, false); }
#undef SGD_WG
        return sgd_check_launch();
    }
    if (bias_slabs)          // exact-f32 / strided fallback: the same [ksplit][cout] partial column sums, by the colsum kernel
        hipLaunchKernelGGL(colsum_stage1_kernel, dim3((cout + 31) / 32, w.ksplit), dim3(256), 0, st, gy, w.rows, cout, gy_ld,
                           w.ksplit, bias_slabs, (float*)nullptr, 0, 1.f);
    const long grid = (long)w.taps * w.co_tiles * w.ci_tiles * w.ksplit;
    if (grid > 0x7fffffffL) return SGD_ERR_ARG;
    if (vec) hipLaunchKernelGGL((wgrad_kernel<true>), dim3((unsigned)grid), dim3(256), 0, st, w);
    else hipLaunchKernelGGL((wgrad_kernel<false>), dim3((unsigned)grid), dim3(256), 0, st, w);
    return sgd_check_launch();
}

extern "C" int sgd_wgrad(const sgd_igemm_args* fwd, const float* gy, int32_t gy_ld, int32_t cout, float* slabs,
                         int32_t ksplit, float* bias_slabs, void* stream) {
    return wgrad_impl(fwd, gy, gy_ld, cout, slabs, ksplit, bias_slabs, nullptr, 0, stream);
}

extern "C" int64_t sgd_wgrad_scratch_bytes(const sgd_igemm_args* fwd, int32_t cout) {
    if (!fwd || fwd->mode != SGD_MODE_CONV3) return 0;
    const int64_t rows = (int64_t)fwd->n * fwd->ho * fwd->wo, urows = (int64_t)fwd->n * fwd->hi * fwd->wi;
    return 4 * (rows * cout + urows * (fwd->c0 + fwd->c1));        // the four operand planes
}

extern "C" int sgd_wgrad_bias_rows(const sgd_igemm_args* fwd, int32_t cout, int32_t gy_ld, int32_t ksplit, int64_t scratch_bytes) {
    if (!fwd || cout <= 0 || ksplit <= 0) return 0;
    const sgd_igemm_args& a = *fwd;
    const long rows = a.mode == SGD_MODE_CONV3 ? (long)a.n * a.ho * a.wo : (long)a.m;
    if (wgrad_is_narrow(a, cout, rows, ksplit)) return ksplit;
    if (wgrad_planes_ok(a, cout, gy_ld, scratch_bytes > 0, scratch_bytes)) return (int)planes_chunks(rows, cout);
    return ksplit;
}

extern "C" int sgd_wgrad_scratch(const sgd_igemm_args* fwd, const float* gy, int32_t gy_ld, int32_t cout, float* slabs,
                                 int32_t ksplit, float* bias_slabs, void* scratch, int64_t scratch_bytes, void* stream) {
    return wgrad_impl(fwd, gy, gy_ld, cout, slabs, ksplit, bias_slabs, scratch, scratch_bytes, stream);
}

extern "C" int sgd_wgrad_reduce(const float* slabs, int32_t ksplit, int32_t taps, int32_t cout, int32_t cin,
                                float* dw, int32_t accumulate, float scale, void* stream) {
    SGD_CLEAR_ERR();
    if (!slabs || !dw || ksplit <= 0 || taps <= 0 || cout <= 0 || cin <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk((long)taps * cout * cin, 8192)), dim3(256), 0,
                       (hipStream_t)stream, slabs, ksplit, taps, cout, cin, dw, accumulate, scale, (const float*)nullptr,
                       (float*)nullptr, 0);
    return sgd_check_launch();
}

extern "C" int sgd_wgrad_reduce_bias(const float* slabs, int32_t ksplit, int32_t taps, int32_t cout, int32_t cin, float* dw,
                                     int32_t accumulate, float scale, const float* bias_slabs, int32_t bias_rows, float* dbias,
                                     void* stream) {
    SGD_CLEAR_ERR();
    if (!slabs || !dw || !bias_slabs || !dbias || ksplit <= 0 || taps <= 0 || cout <= 0 || cin <= 0 || bias_rows <= 0) return SGD_ERR_ARG;
    unsigned grid = nblk((long)taps * cout * cin, 8192);
    if (grid < (unsigned)((cout + 31) / 32)) grid = (cout + 31) / 32;        // one block per 32 bias columns
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs, ksplit, taps, cout, cin, dw,
                       accumulate, scale, bias_slabs, dbias, bias_rows);
    return sgd_check_launch();
}

// two column sums of one shape in ONE launch, rows <= 256 (the per-sample GroupNorm dgamma / dbeta tables [n, c])
extern "C" int sgd_colsum_pair(const float* g1, const float* g2, int32_t rows, int32_t c, int32_t ld, float* out1, float* out2,
                               int32_t accumulate, float scale, void* stream) {
    SGD_CLEAR_ERR();
    if (!g1 || !g2 || !out1 || !out2 || rows <= 0 || rows > 256 || c <= 0 || ld < c) return SGD_ERR_ARG;
    hipLaunchKernelGGL(colsum_stage1_kernel, dim3((c + 31) / 32, 1, 2), dim3(256), 0, (hipStream_t)stream, g1, rows, c, ld, 1,
                       (float*)nullptr, out1, accumulate, scale, g2, out2);
    return sgd_check_launch();
}

extern "C" int sgd_colsum(const float* g, int32_t rows, int32_t c, int32_t ld, float* out, int32_t accumulate,
                          float scale, float* work, int32_t work_chunks, void* stream) {
    SGD_CLEAR_ERR();
    if (!g || !out || !work || rows <= 0 || c <= 0 || ld < c || work_chunks <= 0) return SGD_ERR_ARG;
    int chunks = (rows + 255) / 256;
    if (chunks > work_chunks) chunks = work_chunks;
    hipStream_t st = (hipStream_t)stream;
    if (chunks == 1) {
        hipLaunchKernelGGL(colsum_stage1_kernel, dim3((c + 31) / 32, 1), dim3(256), 0, st, g, rows, c, ld, 1, work, out,
                           accumulate, scale);
        return sgd_check_launch();
    }
    hipLaunchKernelGGL(colsum_stage1_kernel, dim3((c + 31) / 32, chunks), dim3(256), 0, st, g, rows, c, ld, chunks, work,
                       (float*)nullptr, 0, 1.f);
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((c + 31) / 32), dim3(256), 0, st, work, chunks, c, out, accumulate,
                       scale);
    return sgd_check_launch();
}

extern "C" int sgd_colsum_fold(const float* partial, int32_t chunks, int32_t c, float* out, int32_t accumulate,
                               float scale, void* stream) {
    SGD_CLEAR_ERR();
    if (!partial || !out || chunks <= 0 || c <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((c + 31) / 32), dim3(256), 0, (hipStream_t)stream, partial, chunks, c, out,
                       accumulate, scale);
    return sgd_check_launch();
}

extern "C" int sgd_gn_bwd_reduce(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t c_total,
                                 int32_t c_off, const float* a, const float* b, int32_t silu, const float* gu,
                                 int32_t gu_ld, int32_t gu_mode, float drop_p, uint32_t drop_seed, float* S,
                                 void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !a || !b || !gu || !S || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || (c_off & 3) ||
        c_off + c > c_total || (gu_ld & 3))
        return SGD_ERR_ARG;
    if (gu_mode == SGD_RS_AVGPOOL2 && ((h | w) & 1)) return SGD_ERR_ARG;
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(n * ((c + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x, h, w, c,
                       c_total, c_off, a, b, silu, gu, gu_ld, gu_mode, drop_p, drop_seed, S);
    return sgd_check_launch();
}

// chunks of an image for the row-stream reduce (0: the shape is not served -- use sgd_gn_bwd_reduce): at least 32 x 32 pixels, the
// image a whole number of 32 KiB windows, at most 16 chunks, the channel count a power of two that the lane map closes over
extern "C" int sgd_gn_bwd_rows_chunks(int32_t n, int32_t h, int32_t w, int32_t c) {
    if (n <= 0 || h <= 0 || w <= 0 || c < 16 || c > 1024 || (c & (c - 1)) || (long)h * w < 1024) return 0;
    const long quads = (long)h * w * (c >> 2);
    if (quads % GR_WIN) return 0;
    long k = quads / GR_WIN;
    while (k > 16 && (k & 1) == 0) k >>= 1;
    return k <= 16 ? (int)k : 0;
}

extern "C" int sgd_gn_bwd_reduce_rows(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t c_total, int32_t c_off,
                                      const float* a, const float* b, int32_t silu, const float* gu, int32_t gu_ld, float drop_p,
                                      uint32_t drop_seed, int32_t chunks, float* P, void* stream) {
    SGD_CLEAR_ERR();
    // (any divisor of the shape's own chunk count: the sources of a concatenated GroupNorm share one partial table)
    const int kmax = sgd_gn_bwd_rows_chunks(n, h, w, c);
    if (!x || !a || !b || !gu || !P || (c_off & 3) || c_off + c > c_total || (gu_ld & 3) || chunks <= 0 || kmax <= 0 || kmax % chunks)
        return SGD_ERR_ARG;
    const int nset = c <= 256 ? 1 : c / 256;
    const dim3 grid(n * chunks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (nset == 1) hipLaunchKernelGGL(gn_bwd_reduce_rows_kernel<1>, grid, blk, 0, st, x, h * w, c, c_total, c_off, a, b, silu, gu, gu_ld, drop_p, drop_seed, chunks, P);
    else if (nset == 2) hipLaunchKernelGGL(gn_bwd_reduce_rows_kernel<2>, grid, blk, 0, st, x, h * w, c, c_total, c_off, a, b, silu, gu, gu_ld, drop_p, drop_seed, chunks, P);
    else hipLaunchKernelGGL(gn_bwd_reduce_rows_kernel<4>, grid, blk, 0, st, x, h * w, c, c_total, c_off, a, b, silu, gu, gu_ld, drop_p, drop_seed, chunks, P);
    return sgd_check_launch();
}

extern "C" int sgd_gn_bwd_coef(const float* S, int32_t s_chunks, const float* sums, const float* gamma, const float* beta,
                               const float* film, int32_t film_ld, int32_t n, int32_t c, int32_t groups, int32_t hw,
                               float eps, float* A, float* B, float* Cc, float* dgamma_nc, float* dbeta_nc,
                               float* dfilm, void* stream) {
    SGD_CLEAR_ERR();
    if (!S || s_chunks <= 0 || s_chunks > 16 || !sums || !gamma || !beta || !A || !B || !Cc || !dgamma_nc || !dbeta_nc || n <= 0 || c <= 0 ||
        groups <= 0 || c % groups || hw <= 0)
        return SGD_ERR_ARG;
    if ((film || dfilm) && film_ld < 2 * c) return SGD_ERR_ARG;
    hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(nblk((long)n * c)), dim3(256), 0, (hipStream_t)stream, S, s_chunks, sums, gamma,
                       beta, film, film_ld, n, c, groups, hw, eps, A, B, Cc, dgamma_nc, dbeta_nc, dfilm);
    return sgd_check_launch();
}

extern "C" int sgd_gn_bwd_coef_fold(const float* S, int32_t s_chunks, const float* sums, const float* gamma, const float* beta, const float* film,
                                    int32_t film_ld, int32_t n, int32_t c, int32_t groups, int32_t hw, float eps, float* A, float* B,
                                    float* Cc, float* dfilm, float* dgamma, float* dbeta, int32_t accumulate, float scale,
                                    void* stream) {
    SGD_CLEAR_ERR();
    if (!S || s_chunks <= 0 || !sums || !gamma || !beta || !A || !B || !Cc || !dgamma || !dbeta || n <= 0 || n > 256 || c <= 0 || groups <= 0 ||
        c % groups || hw <= 0)
        return SGD_ERR_ARG;
    if ((film || dfilm) && film_ld < 2 * c) return SGD_ERR_ARG;
    const int cpg = c / groups;
    if (s_chunks > 16) return SGD_ERR_ARG;
    const size_t lds = (size_t)(2 * n * cpg + 1) * sizeof(float) + (size_t)n * 4 * sizeof(double)
                       + (s_chunks > 1 ? (size_t)2 * n * cpg * sizeof(float) : 0);           // + the folded sums of a chunked S
    // (+ the kernel's static red[2][8][32]: 2 KB of the same 64 KB default limit; callers fall back to sgd_gn_bwd_coef +
    // sgd_colsum_pair -- train.Backward.gn_bwd applies the same bound)
    if (lds + 2048 > 64 * 1024) return SGD_ERR_ARG;
    hipLaunchKernelGGL(gn_bwd_coef_fold_kernel, dim3(groups), dim3(256), lds, (hipStream_t)stream, S, s_chunks, sums, gamma, beta, film,
                       film_ld, n, c, groups, hw, eps, A, B, Cc, dfilm, dgamma, dbeta, accumulate, scale);
    return sgd_check_launch();
}

extern "C" int sgd_gn_bwd_apply(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t c_total,
                                int32_t c_off, const float* a, const float* b, int32_t silu, const float* gu,
                                int32_t gu_ld, int32_t gu_mode, float drop_p, uint32_t drop_seed, const float* A,
                                const float* B, const float* Cc,
                                const float* gres, int32_t gres_ld, int32_t gres_mode, float* dst, int32_t dst_ld,
                                int32_t dst_off, int32_t accumulate, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !a || !b || !gu || !A || !B || !Cc || !dst || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) ||
        (c_off & 3) || c_off + c > c_total || (gu_ld & 3) || (dst_ld & 3) || (dst_off & 3) || (gres && (gres_ld & 3)))
        return SGD_ERR_ARG;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nblk((long)n * h * w * (c / 4), 65536)), dim3(256), 0,
                       (hipStream_t)stream, x, n, h, w, c, c_total, c_off, a, b, silu, gu, gu_ld, gu_mode, drop_p, drop_seed, A, B, Cc,
                       gres, gres_ld, gres_mode, dst, dst_ld, dst_off, accumulate);
    return sgd_check_launch();
}

extern "C" int sgd_silu_bwd(const float* x, const float* g, int64_t count, float* gx, void* stream) {
    SGD_CLEAR_ERR();
    if (!x || !g || !gx || count <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(nblk(count)), dim3(256), 0, (hipStream_t)stream, x, g, (long)count, gx);
    return sgd_check_launch();
}

extern "C" int sgd_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac,
                            const float* sqrt_1mac, int32_t b, int64_t chw, float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (!x0 || !noise || !t || !sqrt_ac || !sqrt_1mac || !out || b <= 0 || chw <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(q_sample_kernel, dim3(nblk((long)b * chw)), dim3(256), 0, (hipStream_t)stream, x0, noise, t,
                       sqrt_ac, sqrt_1mac, b, (long)chw, out);
    return sgd_check_launch();
}

extern "C" int sgd_mse_loss(const float* eps_nhwc, const float* noise_nchw, int32_t b, int32_t c, int32_t hw,
                            float* per_sample, float* geps_nhwc, void* stream) {
    SGD_CLEAR_ERR();
    if (!eps_nhwc || !noise_nchw || !per_sample || b <= 0 || c <= 0 || hw <= 0) return SGD_ERR_ARG;
    hipLaunchKernelGGL(mse_loss_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, eps_nhwc, noise_nchw, b, c, hw,
                       per_sample, geps_nhwc);
    return sgd_check_launch();
}

extern "C" int sgd_resample_bwd(const float* g, int32_t n, int32_t h, int32_t w, int32_t c, int32_t mode, float* dst,
                                int32_t accumulate, void* stream) {
    SGD_CLEAR_ERR();
    if (!g || !dst || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 3) || (mode != SGD_RS_UP2 && mode != SGD_RS_AVGPOOL2))
        return SGD_ERR_ARG;
    hipLaunchKernelGGL(resample_bwd_kernel, dim3(nblk((long)n * h * w * (c / 4), 65536)), dim3(256), 0,
                       (hipStream_t)stream, g, n, h, w, c, mode, dst, accumulate);
    return sgd_check_launch();
}
