// Fused implicit-GEMM convolution / linear kernel for gfx950 (MI355X).
//
//   M = output rows (pixels of NHWC maps, or flat rows), N = Cout, K = taps * Cin.
//   Tile 128 rows x BN cols, K step = one tap x 32 input channels.
//
// Wave-specialised block (512 threads = 8 waves, 1 block per CU, 2 waves per SIMD):
//   waves 0-3  "compute": MFMA (+ the epilogue).  Each owns a 64x64 (BN=128) or 32x32 (BN=32) output sub-tile.
//              INPUT fragments are ds_read_b128 reads of the LDS tile (those of the next K step are read BEFORE the
//              step barrier so the matrix pipe never waits on LDS latency); WEIGHT fragments never touch LDS: the
//              weights are packed in MFMA fragment order (pack_weight_kernel), so a wave's slice of one K step is
//              eight fully coalesced 1 KiB global loads straight into the operand registers, requested one K step
//              ahead (L2-resident: every CU of an N tile streams the same slices).  Measured reason: with the weight
//              slices staged through LDS (18 KB of ds_write_b128 per step next to 64 KB of fragment reads) the LDS
//              array was ~80 % busy per MFMA-bound step and every ds_read between two MFMAs stalled its wave; the
//              matrix pipe and the LDS phase ran back to back instead of overlapped (profiles/r2_ablation.txt).
//   waves 4-7  "loader": global -> registers -> LDS, inputs only.  The activated input tile (GroupNorm apply +
//              SiLU / LayerNorm, avg-pool / nearest-upsample, channel concat, 16-bit hi/lo split) is staged ONCE per
//              32-channel chunk as a halo tile and re-read by the 9 taps.  The loaders' VALU work (exp, rcp, cvt)
//              executes on the SIMDs' vector pipe while the compute wave of the same SIMD keeps the matrix pipe busy.
//   One s_barrier per K step; A tile double-buffered (triple for 1x1) so no extra barrier at chunk seams.
//
// Arithmetic: PREC_F32 uses v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
//             PREC_F16X3 / BF16X3 split every operand x = hi + lo into two 16-bit floats and
//             accumulate lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_{f16,bf16} in fp32
//             (BASELINE.md section 2 "precision headroom": 4.3e-6 / 2.6e-5 rel. error per UNet eval).
//
// Replaces (reference): nn.Conv2d/Conv1d/Linear call sites listed in include/sgdm_hip.h.
#include <type_traits>
#include <utility>

#include "igemm_shared.h"
#include "prologue.h"

namespace {

// Diagnostic build only (build.py --probe -> libsgdm_hip_probe.so, never loaded by the product path): ablation knobs
// and per-wave cycle accounting.  In the shipped library the knobs fold to constants and no stamp executes.
#ifdef SGDM_PROBE
#define DBG(bit) 0          /* run-time knobs retired: they perturbed the MFMA/LDS interleave; use build.py --abl */
#define SYNC() do { if (g.stamp) { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); __syncthreads(); \
                    const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); \
                    if (g.trace && blockIdx.x < 16 && pr_nbar < 512 && (threadIdx.x & 63) == 0) { \
                        unsigned long long* tp_ = g.trace + (((size_t)blockIdx.x * (NTHREADS / 64) + (threadIdx.x >> 6)) * 512 + pr_nbar) * 2; \
                        tp_[0] = t0_; tp_[1] = t1_; } \
                    pr_bar += t1_ - t0_; ++pr_nbar; } else if (!DBG(64)) { __syncthreads(); } } while (0)
#define PROBE_BEGIN() unsigned long long pr_bar = 0, pr_nbar = 0, pr_epi = 0; const unsigned long long pr_t0 = __builtin_amdgcn_s_memtime()
#define PROBE_END(role) do { if (g.stamp && (threadIdx.x & 63) == 0) { \
        unsigned long long* sp_ = g.stamp + ((size_t)blockIdx.x * (NTHREADS / 64) + (threadIdx.x >> 6)) * 4; \
        sp_[0] = __builtin_amdgcn_s_memtime() - pr_t0; sp_[1] = pr_bar; sp_[2] = pr_nbar; sp_[3] = pr_epi; } } while (0)
#define PROBE_EPI(expr) do { if (g.stamp) { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); expr; pr_epi += __builtin_amdgcn_s_memtime() - t0_; } else { expr; } } while (0)
#else
#define DBG(bit) 0
#define SYNC() __syncthreads()
#define PROBE_BEGIN()
#define PROBE_END(role)
#define PROBE_EPI(expr) expr
#endif

// Compile-time ablations (build.py --abl MASK -> libsgdm_hip_abl<MASK>.so, tools/ only): unlike the run-time DBG knobs of the
// probe build they cost nothing themselves, so the time of a phase is the difference between two such libraries.
//   1 no output stores   2 no residual / bias loads   4 no epilogue statistics   8 no epilogue at all   16 no MFMA
//   32 no weight-fragment loads   64 no input-fragment LDS reads   256 lean conv loader: no transform math   512: no global loads
//   1024: no LDS stores (split + ds_write)   128 no per-step barrier in the compute/loader loops (WRONG results)
#ifdef SGDM_ABL
#define ABL(bit) (((SGDM_ABL) & (bit)) != 0)
#else
#define ABL(bit) false
#endif

// keep a value live without using it (ablation builds); the "v" constraint exists in the device pass only
#if defined(__HIP_DEVICE_COMPILE__)
#define KEEP_LIVE(x) asm volatile("" :: "v"(x))
#else
#define KEEP_LIVE(x) (void)(x)
#endif

// ---------------------------------------------------------------------------------------------
// LDS element packing per precision
// ---------------------------------------------------------------------------------------------
// F32   : row = 32 floats (channel order) + 4 pad.
// split : row = 4 groups of 8 channels; group = hi[8] (16 B) | lo[8] (16 B); + 16 B pad. Same 144 B.
template <int PREC>
__device__ __forceinline__ void lds_store_act(float* rowp, int c4, f32x4 v) {
    if constexpr (PREC == SGD_PREC_F32) {
        *reinterpret_cast<f32x4*>(rowp + c4 * 4) = v;
    } else if constexpr (PREC == SGD_PREC_F16X3) {
        // channels c4*4 .. +3 -> group g = c4 >> 1, position (c4 & 1) * 4 inside the 8-group (2-byte elements)
        _Float16* base = reinterpret_cast<_Float16*>(rowp) + (c4 >> 1) * 16 + (c4 & 1) * 4;
        u32x2 h, l;
        split4_f16(v, h, l);
        *reinterpret_cast<u32x2*>(base) = h;
        *reinterpret_cast<u32x2*>(base + 8) = l;
    } else {
        typedef typename Split<PREC>::T T;
        T* base = reinterpret_cast<T*>(rowp) + (c4 >> 1) * 16 + (c4 & 1) * 4;
        T h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Split<PREC>::split(v[j], h[j], l[j]);
        }
        typedef T T4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<T4*>(base) = T4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<T4*>(base + 8) = T4{l[0], l[1], l[2], l[3]};
    }
}

// MFMA operand fragments.  AU: the input side (pixels) of ONE stage = one 32-row block x one K sub-step, read from the LDS
// tile; B: the weight side of one K sub-step for the wave's NT 32-column blocks, loaded from the fragment-ordered packed
// weights in global memory (`p` already carries lane * 16).  A stage multiplies one AU into NT accumulator tiles.
template <int PREC, int NT> struct Frag {
    typedef typename Split<PREC>::T T;
    typedef T T8 __attribute__((ext_vector_type(8)));
    static constexpr int NKS = KC / 16;           // 16 channels per sub-step
    static constexpr int NREADS = 2;              // ds_read_b128 per stage
    static constexpr int NWLOADS = 2 * NT;        // 16-byte global loads per sub-step
    static constexpr int NMMA = 3 * NT;           // MFMAs per stage
    struct AU {
        T8 h, l;
        __device__ __forceinline__ void load(const float* rowp, int ks, int lh) {
            const int goff = (ks * 2 + lh) * 8;   // float offset of this lane-half's 8-channel group
            h = *reinterpret_cast<const T8*>(rowp + goff);
            l = *reinterpret_cast<const T8*>(rowp + goff + 4);
        }
    };
    struct B {
        T8 h[NT], l[NT];
        __device__ __forceinline__ void load(const char* p, int ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                h[nt] = *reinterpret_cast<const T8*>(p + nt * WUNIT + ks * 2048);
                l[nt] = *reinterpret_cast<const T8*>(p + nt * WUNIT + ks * 2048 + 1024);
            }
        }
    };
    static __device__ __forceinline__ void mma(f32x16 (&acc)[NT], const AU& a, const B& b) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (PREC == SGD_PREC_F16X3) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h[nt], a.l, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.l[nt], a.h, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h[nt], a.h, acc[nt], 0, 0, 0);
            } else {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.h[nt], a.l, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.l[nt], a.h, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.h[nt], a.h, acc[nt], 0, 0, 0);
            }
        }
    }
};
// v_mfma_f32_16x16x32_{f16,bf16} form of the split-precision stage: one stage = a 16-row block x the WHOLE 32-channel K
// step, multiplied into NCB 16-column accumulator tiles (4 registers each).  Same MFMA cycles per flop as the 32x32x16
// form, but the chip holds a higher clock on this shape under load (MI355X_MICROARCH.md, DVFS give-back item 7: +12..15 %
// on random data).  The weights are read from the SAME packed units as the 32x32 form -- a lane of this form needs
// W[co = 16 c + (lane & 15)][ci = 8 (lane >> 4) + j], which sits at sub-step (lane >> 5), lane-half (lane >> 4) & 1 of the
// 32x32 fragment order: only the per-lane offset inside the 4 KiB unit differs (`p` carries it).
template <int PREC, int NCB> struct Frag16 {
    typedef typename Split<PREC>::T T;
    typedef T T8 __attribute__((ext_vector_type(8)));
    static constexpr int NREADS = 2;              // ds_read_b128 per stage
    static constexpr int NWLOADS = 2 * NCB;       // 16-byte global loads per K step
    static constexpr int NMMA = 3 * NCB;          // MFMAs per stage
    struct AU {
        T8 h, l;
        __device__ __forceinline__ void load(const float* rowp, int kg) {      // kg = lane >> 4: this lane's 8-channel group
            h = *reinterpret_cast<const T8*>(rowp + kg * 8);
            l = *reinterpret_cast<const T8*>(rowp + kg * 8 + 4);
        }
    };
    struct B {
        T8 h[NCB], l[NCB];
        __device__ __forceinline__ void load(const char* p) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                h[cb] = *reinterpret_cast<const T8*>(p + (cb >> 1) * WUNIT + (cb & 1) * 256);
                l[cb] = *reinterpret_cast<const T8*>(p + (cb >> 1) * WUNIT + (cb & 1) * 256 + 1024);
            }
        }
    };
    static __device__ __forceinline__ f32x4 mma1(f32x4 acc, const AU& a, const B& b) {      // column block 0 only
        if constexpr (PREC == SGD_PREC_F16X3) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.h[0], a.l, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.l[0], a.h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.h[0], a.h, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.h[0], a.l, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.l[0], a.h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.h[0], a.h, acc, 0, 0, 0);
        }
        return acc;
    }
    static __device__ __forceinline__ void mma(f32x4 (&acc)[NCB], const AU& a, const B& b) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if constexpr (PREC == SGD_PREC_F16X3) {
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.h[cb], a.l, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.l[cb], a.h, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.h[cb], a.h, acc[cb], 0, 0, 0);
            } else {
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.h[cb], a.l, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.l[cb], a.h, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.h[cb], a.h, acc[cb], 0, 0, 0);
            }
        }
    }
};
template <int NCB> struct Frag16<SGD_PREC_F32, NCB> {     // never used (the exact mode keeps v_mfma_f32_32x32x2_f32)
    static constexpr int NREADS = 1, NWLOADS = 1, NMMA = 1;
    struct AU { __device__ __forceinline__ void load(const float*, int) {} };
    struct B { __device__ __forceinline__ void load(const char*) {} };
    static __device__ __forceinline__ void mma(f32x4 (&)[NCB], const AU&, const B&) {}
    static __device__ __forceinline__ f32x4 mma1(f32x4 acc, const AU&, const B&) { return acc; }
};
template <int NT> struct Frag<SGD_PREC_F32, NT> {
    static constexpr int NKS = KC / 8;            // 8 channels per sub-step (4 MFMA k-pairs)
    static constexpr int NREADS = 1;
    static constexpr int NWLOADS = NT;
    static constexpr int NMMA = 4 * NT;
    struct AU {
        f32x4 v;
        __device__ __forceinline__ void load(const float* rowp, int ks, int lh) {
            v = *reinterpret_cast<const f32x4*>(rowp + ks * 8 + lh * 4);
        }
    };
    struct B {
        f32x4 v[NT];
        __device__ __forceinline__ void load(const char* p, int ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = *reinterpret_cast<const f32x4*>(p + nt * WUNIT + ks * 1024);
        }
    };
    static __device__ __forceinline__ void mma(f32x16 (&acc)[NT], const AU& a, const B& b) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b.v[nt][j], a.v[j], acc[nt], 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------------------------
// the kernel.  TAPS = 9 (CONV3) or 1 (FLAT).  Persistent: a block walks its tiles as ONE continuous
// stream of K steps, so the loaders are already staging tile t+1 while the compute waves finish and
// store tile t (no per-tile prologue / epilogue bubble on the matrix pipe).
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, true));
}
// sum over the 32 lanes that share (lane >> 5), DPP only (no LDS round trip); the total lands in lanes 16..31 / 48..63
__device__ __forceinline__ float half_wave_sum_hi(float v) {
    v = dpp_add<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);      // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);      // row_mirror: every lane holds its row-of-16 total
    return dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3: + the total of the row below
}

// compile-time loop: f(std::integral_constant<int, 0>()) ... f(std::integral_constant<int, N - 1>())
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>());
}

// sum over the 16 lanes of a DPP row; every lane of the row ends with the total
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);      // row_half_mirror
    return dpp_add<0x140, 0xF>(v);   // row_mirror
}

struct Tile {
    int n0c;                 // first output column
    int img0, ty0, tx0;      // CONV3 origin
    long m0;                 // FLAT origin
};

__device__ __forceinline__ Tile tile_at(const Geo& g, int lin, int bn, int tw, int th) {
    Tile t;
    const int mtile = lin / g.nt, ntile = lin - mtile * g.nt;
    t.n0c = ntile * bn;
    const int per_img = g.tiles_x * g.tiles_y;
    const int it = mtile / per_img, rem = mtile - it * per_img;
    t.img0 = it * g.nb;
    t.ty0 = (rem / g.tiles_x) * th;
    t.tx0 = (rem % g.tiles_x) * tw;
    t.m0 = (long)mtile * BM;
    return t;
}


// One block of the persistent grid: the compile-time shape of an instance, the run-time schedule of this block, and the two
// wave roles as member functions -- the kernel itself (below) only carves up LDS, builds the first tile tables and hands the
// waves their role.  (Round 6: the 1,350-line kernel body cut into units; the ISA of all 25 instances is unchanged.)
template <int BN, int PREC, bool VEC, int TAPS, bool DEFER>
struct IgemmBlock {
    // ------------------------------------------------------------------------------------------ compile-time shape
    static constexpr bool CONV = TAPS == 9;
    // FLAT with TAPS = FG > 1 (round 4): a "chunk" is FG consecutive 32-channel planes of the 128 input rows, staged side
    // by side in one ring slot ([FG][128][LDA]) and consumed as FG K steps between two barriers -- the 1x1 / linear
    // launches then run in the conv kernel's rhythm (one s_barrier per chunk, loaders two chunks ahead in LDS) instead of
    // one barrier per K step, where the two roles took turns (measured round 2: 2.4k cycles of work + 1.2k in the barrier
    // on BOTH sides per step).  The packed weights are unchanged: [32-channel plane][output block] is already the order.
    static constexpr int FG = CONV ? 1 : TAPS;
    static constexpr int KCC = KC * FG;                  // input channels per chunk
    static constexpr int NA = 3;                         // A tile ring depth (chunks; the loaders run two ahead)
    // wave tile: BN >= 128 -> every MFMA wave owns ALL 128 rows x its own BN / 4 columns, so the four waves of a block load
    // disjoint weight fragments (a 64 x 64 split made two waves fetch the same 8 KB per step: the per-CU vector memory
    // pipe was > 50 % busy and its full queue stalled the in-order MFMA waves at their loads); the input fragments they
    // share come from LDS.  BN = 256 (64 columns per wave, 128 accumulator registers) multiplies every staged input
    // chunk into twice the MFMAs: the loaders' transform work per flop halves.  BN = 32: four 32 x 32 waves stacked along M.
    static constexpr int WM = (BN >= 128) ? 128 : 32;    // wave tile rows
    static constexpr int WN = (BN >= 128) ? BN / 4 : 32; // wave tile cols
    static constexpr int MT = WM / 32, NT = WN / 32;
    static constexpr int WAVES_N = BN / WN;
    typedef Frag<PREC, NT> FragT;
    // 16x16x32 MFMA form (split modes, 128- and 32-column tiles; the 64-column wave tile has no registers left for the
    // double-buffered weight fragments this form needs): accumulators are [RB row blocks of RBH rows][CBN column blocks]
    // Measured (round 3, tools/ab_conv.py, UNet batch 80, against the 32x32x16 form of the same kernel): +4 % on
    // 128-channel 3x3 layers, +7..9 % at 256..384 input channels, +13..16 % at 512..1024, 1x1 launches unchanged; the
    // sampling step 19.5 -> 18.4 ms.  Same matrix-pipe cycles per flop, but the chip is power-limited under this load
    // and the smaller MFMA sustains a higher clock.  The loader-side epilogue (DEFER) and the 128 x 256 tile keep the
    // 32x32x16 form (the 16x16 form on the wide tile was built and lost 0..14 % to the 128 x 128 tile on every shape:
    // profiles/r4_ab_m16_256.txt; round 6 ported DEFER to the 16x16 form: still slower, profiles/r6_ab_defer_m16.txt).
    static constexpr bool M16 = PREC != SGD_PREC_F32 && BN <= 128 && !DEFER;
    static constexpr int RB = M16 ? WM / 16 : MT, RBH = M16 ? 16 : 32;
    static constexpr int CBN = M16 ? WN / 16 : NT, CBW = M16 ? 16 : 32;      // column blocks of a wave tile and their width
    static constexpr int QPB = M16 ? 1 : 4;                                  // 4-channel quads a lane holds per (row block, column block)
    typedef Frag16<PREC, CBN> Frag16T;
    typedef std::conditional_t<M16, f32x4, f32x16> AccV;
    static constexpr int NKS = FragT::NKS;

    static constexpr int STG_LD = BN + 4;
    static_assert(!DEFER || (BN == 128 && TAPS == 9 && VEC && !M16), "loader-side epilogue: 3x3, 128-column tiles");

    // ------------------------------------------------------------------------------------------ run-time state of the block
    const sgd_igemm_args& a;
    const Geo& g;
    float* As;                    // [NA][pix][LDA]
    int2* pixtab;                 // [4][pix] (source row or -1, image n)
    float* bias_s;                // [cout_p]
    float* stg;                   // [BM][STG_LD], DEFER only
    int a_floats;
    bool bias_lds;
    int tid, lane, wave;
    int total, xchunk, nloc, xcd, loc, xbeg, xend, cin, nchunks, xtiles, nfull, xrem, split;
    int ntiles, rem_lin, rem_part, last_c0, last_c1;
    int Q, S, s, TW, TH, M;

    __device__ __forceinline__ int lin_of(int k) const { return (rem_lin >= 0 && k == ntiles - 1) ? rem_lin : xbeg + loc + k * nloc; }
    // chunk range of the block's k-th tile: only the last one can be partial
    __device__ __forceinline__ int cbeg(int k) const { return k == ntiles - 1 ? last_c0 : 0; }
    __device__ __forceinline__ int cend(int k) const { return k == ntiles - 1 ? last_c1 : nchunks; }

    // LDS carve-up, XCD-aware schedule and balanced tail of this block; false: the block has no tile
    __device__ __forceinline__ bool init(float* smem) {
        a_floats = g.pix * LDA;
        As = smem;                                       // [NA][pix][LDA]
        pixtab = reinterpret_cast<int2*>(smem + (size_t)NA * a_floats);   // [4][pix] (source row or -1, image n)
        // bias of the whole layer (zeros without one): the epilogue reads it from LDS -- a global load there is a ~2k-cycle
        // dependent wait per batch of quads on a wave that has nothing else to issue (measured: 7.8k cycles per tile for the
        // bias-only epilogue of a conv without residual)
        bias_s = smem + (size_t)NA * a_floats + (size_t)g.pix * 8;       // [cout_p]
        bias_lds = a.cout_p <= BIAS_LDS_MAX;        // very wide layers (all FiLM projections as one GEMM) read it from global
        // DEFER kernels (3x3, 128-column tiles, 16-byte outputs, no or same-row residual, bias in LDS): the epilogue of every
        // tile but a block's last runs on the LOADER waves.  Measured (round 3, compile-time ablations at UNet batch 80): the
        // epilogue costs a compute wave 8..10 us per tile -- 30..40 % of a 128-channel 3x3 layer, 12..14 % of a 512-channel
        // one -- and nearly all of it is waiting: vmcnt counts loads and stores in issue order, so the residual loads cost a
        // round trip per batch and the next tile's first weight wait sits behind the acknowledgement of all 16 stores, with
        // the matrix pipe idle (moving the epilogue into the next K loop of the SAME wave moves the stall, it does not remove
        // it: tried).  So the compute waves copy their accumulators to an LDS staging tile (stg, [BM][BN + 4] floats: 0.3 us)
        // and go on; the loaders, whose own waits have a whole chunk period of slack, drain it slice by slice.
        stg = bias_s + a.cout_p;                         // [BM][STG_LD], DEFER only (sgd_igemm sizes the allocation)

        tid = threadIdx.x; lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keeps its arithmetic scalar

        // XCD-aware persistent schedule: blocks b, b+8, .. share an XCD (and its L2).  XCD x owns the tile
        // range [x*xchunk, (x+1)*xchunk); its blocks stride through it together, so tiles in flight on one
        // XCD are neighbours (the N tiles of one M tile share the input tile, neighbours share halos).
        total = g.mt * g.nt;
        xchunk = (total + 7) >> 3;
        nloc = gridDim.x >> 3;              // blocks per XCD
        xcd = blockIdx.x & 7; loc = blockIdx.x >> 3;
        xbeg = xcd * xchunk; xend = (xbeg + xchunk < total) ? xbeg + xchunk : total;
        cin = a.c0 + a.c1;
        nchunks = (cin + KCC - 1) / KCC;
        // Balanced tail (args.work): the XCD's tiles are nfull whole rounds of its nloc blocks plus R < nloc tiles.  Instead of
        // a last round that keeps R blocks busy and nloc - R idle, each of those R tiles is split along K into `split` chunk
        // ranges computed by `split` different blocks at the same time: parts 0 .. split-2 store their partial accumulators to
        // the workspace and signal, the block of the LAST range adds them in part order and runs the epilogue.  Producers
        // never wait, so the protocol cannot deadlock whatever the residency of the blocks; every block meets its split tile
        // LAST, after its whole tiles.
        xtiles = xend > xbeg ? xend - xbeg : 0;
        nfull = xtiles / nloc; xrem = xtiles - nfull * nloc;
        split = (a.work && !ABL(2048)) ? tail_split(xrem, nloc, nchunks, TAPS) : 0;
        rem_lin = -1; rem_part = 0; last_c0 = 0; last_c1 = nchunks;
        if (!split) {
            ntiles = loc < xtiles ? (xtiles - loc + nloc - 1) / nloc : 0;
        } else {
            ntiles = nfull;
            if (loc < xrem * split) {
                rem_lin = xbeg + nfull * nloc + loc / split;
                rem_part = loc % split;
                last_c0 = rem_part * nchunks / split;
                last_c1 = (rem_part + 1) * nchunks / split;
                ++ntiles;
            }
        }
        if (ntiles == 0) return false;

        Q = (ntiles - 1) * nchunks + (last_c1 - last_c0);   // channel chunks of this block
        S = Q * TAPS;                       // K steps of this block
        s = CONV ? a.stride : 1;
        TW = 1 << g.tw_l2; TH = 1 << g.th_l2;
        M = CONV ? a.n * a.ho * a.wo : a.m;

        // source-row table of tile k's A rows (index math once per tile, not per chunk)
        return true;
    }

    // source-row table of tile k's A rows (index math once per tile, not per chunk)
    __device__ __forceinline__ void build_pixtab(int k, int t0, int nthr) const {
        if (k >= ntiles) return;
        const Tile T = tile_at(g, lin_of(k), BN, TW, TH);
        int2* tab = pixtab + (size_t)(k & 3) * g.pix;
        for (int pix = t0; pix < g.pix; pix += nthr) {
            int2 e;
            e.x = -1;
            e.y = 0;
            if (CONV) {
                int hx = pix % g.hw, t = pix / g.hw;
                int hy = t % g.hh, nb = t / g.hh;
                int n = T.img0 + nb, y = T.ty0 * s - 1 + hy, x = T.tx0 * s - 1 + hx;
                if (n < a.n && y >= 0 && y < g.hc && x >= 0 && x < g.wc) {
                    if (a.resample == SGD_RS_UP2) e.x = (n * a.hi + (y >> 1)) * a.wi + (x >> 1);
                    else if (a.resample == SGD_RS_AVGPOOL2) e.x = (n * a.hi + 2 * y) * a.wi + 2 * x;
                    else if (a.resample == SGD_RS_ZEROUP2) e.x = ((y | x) & 1) ? -1 : (n * a.hi + (y >> 1)) * a.wi + (x >> 1);
                    else e.x = (n * a.hi + y) * a.wi + x;
                    e.y = n;
                }
            } else {
                long row = T.m0 + pix;
                if (row < M) {
                    e.x = (int)row;
                    e.y = (a.pro == SGD_PRO_AFFINE_NC) ? (int)(row / a.rows_per_n) : 0;
                }
            }
            tab[pix] = e;
        }
    }

    // bytes of one block's partial accumulators (balanced tail)
    static constexpr size_t SLAB = (size_t)RB * CBN * QPB * NCOMP * 16;
    // quad q of a lane's (row block, column block) accumulator tile
    static __device__ __forceinline__ f32x4 accq(const AccV (&acc)[RB][CBN], int mt, int nt, int q) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][(M16 ? 0 : q * 4) + j];
        return v;
    }
    // Which tile pixel a lane's MFMA row stands for.  A 32-row block covers two image rows of a 16-wide tile whose halo
    // rows are 18 pixels apart in LDS; with lane == pixel the 16 lanes ds_read_b128 serves per LDS cycle ({0-3,12-15,
    // 20-27} / {4-11,16-19,28-31}) meet pixels that collide mod 16 -> 2-way bank conflicts on every input fragment read.
    // The permutation gives each such lane group 16 pixels distinct mod 16 (36-float pixel pitch: bank = 9*pixel mod
    // 16 quads).  The epilogue uses the same map, so results are unchanged.
    // (a function of the lane: the epilogue recomputes it from an opaque copy of `lane` instead of keeping it -- and what
    // is derived from it -- in registers across the K loop)
    __device__ __forceinline__ int pixel_of_lane(int ln) const {
    const int l5 = ln & 31;
        int px = l5;
        if (CONV && g.tw_l2 == 4 && s == 1) {
            if (l5 < 4) px = l5;
            else if (l5 < 12) px = l5 + 4;
            else if (l5 < 16) px = l5 - 8;
            else if (l5 < 20) px = l5;
            else if (l5 < 28) px = l5 + 2;
            else px = l5 < 30 ? l5 - 8 : l5;
        }
        if constexpr (M16) {
            // 16-row blocks: lane l reads row prow(l & 15), channel group l >> 4 (32 bytes further per group).  ds_read_b128
            // serves {0-3,12-15,20-27} / {4-11,16-19,28-31} per LDS cycle: 8 lanes at group g and 8 at g + 1; with the
            // 36-float pixel pitch the quad bank is 9 * pixel + 2 * group mod 16, conflict-free iff the rows read at the odd
            // group all have one parity: lanes 4..11 take the odd rows, lanes 0..3 / 12..15 the even ones
            const int i16 = ln & 15;
            px = i16 < 4 ? 2 * i16 : (i16 < 12 ? 2 * (i16 - 4) + 1 : 2 * i16 - 16);
        }
        return px;
    }
    // epilogue of one tile on a compute wave: bias (+ residual, + the other blocks' partial accumulators), stores, GroupNorm partial
    // statistics.  RES: 0 none, 1 same rows, 2 avg-pool of 2x map, 3 nearest of 1/2 map; PART: balanced tail (add the producers' slabs)
    template <int RES, bool PART>
    __device__ __forceinline__ void epilogue(const AccV (&acc)[RB][CBN], const Tile& T, int wm, int wn, int lane_e, int cb, float wsk,
                                             const char* part_base, int nparts) const;
    __device__ __forceinline__ void loader_role();
    __device__ __forceinline__ void compute_role();
};

// =========================================================================================
// loader role (waves 4-7)
// =========================================================================================
template <int BN, int PREC, bool VEC, int TAPS, bool DEFER>
__device__ __forceinline__ void IgemmBlock<BN, PREC, VEC, TAPS, DEFER>::loader_role() {
    PROBE_BEGIN();
    // The loader shares its SIMD's vector issue with an MFMA wave that always has an instruction waiting; at equal
    // priority the older (MFMA) wave wins every arbitration and the loader got ~1 issue slot per MFMA (measured: ~460
    // vector instructions per chunk took 10k cycles and the compute waves waited 27 % of the time at the chunk
    // barrier).  An MFMA needs one issue slot per 32 cycles, so handing the loader the priority costs the matrix
    // pipe nothing as long as the loader's own stream has dependency gaps.
    __builtin_amdgcn_s_setprio(2);
    // ---- loader-side epilogue (DEFER) ----------------------------------------------------------------------------
    // Period q (between barriers q and q+1) belongs to tile dk; tile dk - 1 was staged in its first period.  Its 128 x 32
    // quads are 16 per loader thread (thread = channel quad dcq, rows drg + 8 i), drained in dnck - 1 slices of <= 8:
    // period j of the tile FINISHES slice j - 1 (staged sums * scale + bias + residual -> store, GroupNorm partial sums)
    // and REQUESTS the residual quads of slice j.  Every period issues exactly 8 + 2 buffer stores and 8 buffer loads --
    // slots without work carry an out-of-range offset, which the hardware drops -- so the loader stays branch-free
    // around its vector memory operations and the compiler's counted waits stay exact: the input loads of the chunk
    // pipeline are never waited for behind this period's stores.
    const int dlt = tid - NCOMP;
    const int dcq = (dlt >> 6) * 8 + (dlt & 7), drg = (dlt & 63) >> 3;
    int dk = 0, dqs = 0, dnck = cend(0) - cbeg(0);
    Tile dT = tile_at(g, lin_of(0), BN, TW, TH);       // tile dk - 1
    int dcol = 0;                                      // first output channel of this thread's quad in tile dk - 1
    unsigned dsoff = 0xFFFFFFFFu;                      // statistics slot of tile dk - 1 (byte offset), this thread's quad
    f32x4 dres[8], ds1 = {0.f, 0.f, 0.f, 0.f}, ds2 = ds1;
    unsigned dyoff[8];
    int di0 = 0;                                       // first item of the slice requested one period ago
#pragma unroll
    for (int j = 0; j < 8; ++j) { dres[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dyoff[j] = 0xFFFFFFFFu; }
    const float dwsi = (DEFER && a.w_scale_inv) ? *a.w_scale_inv : 1.f;
    const long drows = (long)a.n * a.ho * a.wo;
    const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(drows * a.y_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t dr_rs = __builtin_amdgcn_make_buffer_rsrc(a.res ? const_cast<float*>(a.res) : a.y, 0,
                                                                           a.res ? (int)(drows * a.cout * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ds_rs = __builtin_amdgcn_make_buffer_rsrc(a.stats ? a.stats : a.y, 0,
                                             a.stats ? (int)((long)a.n * g.sparts * 2 * a.cout * 4) : 0, 0x00020000);
    // slots per period, one value per launch (whole tiles have nchunks - 1 slices; a block's last, possibly partial tile
    // has at least as many periods as MIN_PART_STEPS / 9 = 3): the period loops below are instantiated per value
    const int dnq = !DEFER ? 0 : (nchunks - 1 >= 8 && cend(ntiles - 1) - cbeg(ntiles - 1) - 1 >= 8) ? 2
                               : (nchunks - 1 >= 4 && cend(ntiles - 1) - cbeg(ntiles - 1) - 1 >= 4) ? 4 : 8;
    auto with_nq = [&](auto&& f) __attribute__((always_inline)) {
        if (dnq == 2) f(std::integral_constant<int, 2>());
        else if (dnq == 4) f(std::integral_constant<int, 4>());
        else if (dnq == 8) f(std::integral_constant<int, 8>());
        else f(std::integral_constant<int, 0>());
    };
    auto drain = [&](int q, auto nqc) __attribute__((always_inline)) {
        constexpr int NQ = decltype(nqc)::value;     // quad slots per period: >= ceil(16 / (chunks per tile - 1))
        if constexpr (DEFER && NQ > 0) {
            if (q >= dqs + dnck) {                     // first period of the next tile: the one before it is pending
                dqs += dnck;
                ++dk;
                dnck = cend(dk) - cbeg(dk);
            }
            const int j = q - dqs, nsl = dnck - 1;
            // ---- finish the slice requested one period ago
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_s + (dcol < a.cout_p ? dcol : 0));
#pragma unroll
            for (int jj = 0; jj < NQ; ++jj) {
                const int frow = drg + 8 * (di0 + jj < 16 ? di0 + jj : 15);
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + (size_t)frow * STG_LD + dcq * 4);
                v = v * dwsi + bq + dres[jj];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), dy_rs, dyoff[jj], 0, 0);
                if (dyoff[jj] != 0xFFFFFFFFu) {
                    ds1 += v;
                    ds2 += v * v;
                }
            }
            // ---- GroupNorm partial sums of the tile: after its last slice
            const bool last = dk > 0 && j == nsl;
            f32x4 t1 = ds1, t2 = ds2;
            if (last) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t1[e] += __shfl_xor(t1[e], o, 64);
                        t2[e] += __shfl_xor(t2[e], o, 64);
                    }
                ds1 = ds2 = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const unsigned so = (last && drg == 0) ? dsoff : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, t1), ds_rs, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, t2), ds_rs,
                                                   so == 0xFFFFFFFFu ? so : so + (unsigned)a.cout * 4u, 0, 0);
            // ---- request slice j of tile dk - 1
            const bool req = dk > 0 && j < nsl;
            int i0 = 0, i1 = 0;
            if (req) { i0 = (16 * j) / nsl; i1 = (16 * (j + 1)) / nsl; }
            di0 = i0;
            if (dk > 0 && j == 0) {                    // per tile: its first channel / statistics slot for this thread
                dT = tile_at(g, lin_of(dk - 1), BN, TW, TH);
                const Tile& T = dT;
                dcol = T.n0c + dcq * 4;
                const int part = (T.ty0 >> g.th_l2) * g.tiles_x + (T.tx0 >> g.tw_l2);
                dsoff = (a.stats && dcol < a.cout) ? (unsigned)((((long)T.img0 * g.sparts + part) * 2 * a.cout + dcol) * 4) : 0xFFFFFFFFu;
            }
            const Tile& T = dT;
#pragma unroll
            for (int jj = 0; jj < NQ; ++jj) {
                const int i = i0 + jj;
                const int row = drg + 8 * (i < 16 ? i : 15);
                const int tx = row & (TW - 1), ty = (row >> g.tw_l2) & (TH - 1), nb = row >> (g.tw_l2 + g.th_l2);
                const int n = T.img0 + nb;
                const bool ok = i < i1 && nb < g.nb && n < a.n && dcol < a.cout;
                const int orow = (n * a.ho + T.ty0 + ty) * a.wo + T.tx0 + tx;
                dyoff[jj] = ok ? (unsigned)(((long)orow * a.y_ld + dcol) * 4) : 0xFFFFFFFFu;
                const unsigned ro = ok ? (unsigned)(((long)orow * a.cout + dcol) * 4) : 0xFFFFFFFFu;
                dres[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dr_rs, ro, 0, 0));
            }
        }
        (void)q;
    };
    if constexpr (CONV) {
        // =================================================================================
        // lean loader (the sampler's ResBlock convs: stride 1, per-image GroupNorm affine (+SiLU) or no prologue,
        // whole 32-channel chunks from one source, no dropout).  At taps 1..6 of a chunk every loader thread transforms
        // ONE raw row quad requested nine steps earlier into LDS and requests the same item of the next chunk; the
        // per-tile work (source rows, padding flags) is hoisted, chunk pointers are wave-uniform scalars.  Everything is
        // branch-free so the in-order vmcnt waits stay exact.
        // =================================================================================
        // two straight-line instantiations: GroupNorm affine + SiLU (every ResBlock conv of the sampler) and no prologue at
        // all (input gradients, plain convs).  With the mode as RUN-time flags each of the six items of a chunk was a
        // chain of small basic blocks (selects between the transformed and untransformed value, moves at the joins),
        // which the scheduler could neither interleave nor strip -- and loader issue slots are what paces the block.
        const bool uni_rt = VEC && g.nb == 1 && a.pro == SGD_PRO_AFFINE_NC;
        // (round 5: train-time dropout -- the 22 out_layers.3 convs of a training forward -- rides on the GroupNorm + SiLU
        // instantiation: two hashes per item, sgd_drop4; those launches took the general loader before, +13 % each)
        const bool drop_rt = a.drop_p > 0.f;
        const bool lean2 = g.fast_a && VEC && ((uni_rt && a.pro_silu) || (a.pro == SGD_PRO_NONE && !a.pro_silu))
                           && (!drop_rt || (uni_rt && a.pro_silu && a.resample != SGD_RS_AVGPOOL2))
                           && cin % KC == 0 && (a.c1 == 0 || a.c0 % KC == 0) && !DBG(3);
        if (lean2) {
            constexpr int LT = A_THREADS;                            // 256 loader threads
            constexpr int AJ = (FAST_PIX * 8 + LT - 1) / LT;         // input items per thread per chunk (6)
            const int lt = tid - NCOMP;
            const int c4 = lt & 7;                                   // channel quad (inputs and weights alike)
            const int items = g.pix * 8;
            auto go = [&](auto unic, auto poolc, auto dropc) __attribute__((always_inline)) {
                constexpr bool uni = decltype(unic)::value;          // true: GN affine + SiLU, false: raw input
                constexpr bool DROP = decltype(dropc)::value;        // train-time dropout behind the SiLU (uni, no pool)
                constexpr bool LATE = false;
                constexpr int NS = decltype(poolc)::value ? 4 : 1;   // source pixels per item (fused 2x2 average pool)
                constexpr int T0 = LATE ? 5 : 1;
                typedef std::integral_constant<int, 0> R0;
                typedef std::integral_constant<int, 1> R1;
                typedef std::integral_constant<int, 2> R2;
                // ---- inputs
                int pixj[AJ], rows2[AJ];
                bool live[AJ];
#pragma unroll
                for (int j = 0; j < AJ; ++j) {
                    const int idx = lt + j * LT;
                    live[j] = idx < items;
                    pixj[j] = (live[j] ? idx : items - 1) >> 3;
                }
                unsigned valid1 = 0, valid2 = 0;
                auto load_rows = [&](int k) __attribute__((always_inline)) {
                    const int2* tab = pixtab + (size_t)(k & 3) * g.pix;
                    valid2 = 0;
#pragma unroll
                    for (int j = 0; j < AJ; ++j) {
                        const int ex = tab[pixj[j]].x;
                        rows2[j] = ex < 0 ? 0 : ex;
                        if (ex >= 0) valid2 |= 1u << j;
                    }
                };
                struct S { int k, chunk, img0; const float* src; int stride; const float* ka; const float* kb; };
                auto fill = [&](S& c) __attribute__((always_inline)) {
                    const int ch = c.chunk * KC;
                    if (ch < a.c0) { c.src = a.x0 + ch; c.stride = a.c0; }
                    else { c.src = a.x1 + (ch - a.c0); c.stride = a.c1; }
                    const long ko = (long)c.img0 * cin + ch;
                    c.ka = uni ? a.pa + ko : a.x0;   // no prologue: 32 harmless bytes instead of a branch around the load
                    c.kb = uni ? a.pb + ko : a.x0;
                };
                auto advance = [&](S c) __attribute__((always_inline)) {
                    if (++c.chunk == cend(c.k)) {
                        if (c.k + 1 < ntiles) {      // tile index math (integer divisions) once per tile, not per chunk
                            ++c.k; c.chunk = cbeg(c.k);
                            c.img0 = tile_at(g, lin_of(c.k), BN, TW, TH).img0;
                        }
                        else c.chunk = cend(c.k) - 1;
                    }
                    fill(c);
                    return c;
                };
                f32x4 araw[AJ][NS];
                Coef kq;
                auto transform = [&](f32x4 v, bool ok) __attribute__((always_inline)) {
                    if (ABL(256)) return v;
                    if constexpr (uni) {
                        // SiLU with the padding mask folded into the denominator: t / (den + e^-t), den = 1 or +inf
                        // (rcp(inf) = 0 and t is finite: a clamped, real input row) -- one select per item instead of four
                        const float den = ok ? 1.0f : __builtin_inff();
                        v = v * kq.p + kq.q;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(den + __expf(-v[e]));
                    } else {
                        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    return v;
                };
                auto issue_item = [&](const S& c, int j) __attribute__((always_inline)) {
                    const float* p0 = c.src + (long)rows2[j] * c.stride + c4 * 4;
                    if (ABL(512)) { KEEP_LIVE(p0); return; }
                    araw[j][0] = ld4(p0);
                    if constexpr (NS == 4) {            // ResBlock(down): the conv reads avg_pool2d(SiLU(GN(x))) (openaimodel.py:301-306)
                        araw[j][1] = ld4(p0 + c.stride);
                        araw[j][2] = ld4(p0 + (long)a.wi * c.stride);
                        araw[j][3] = ld4(p0 + (long)(a.wi + 1) * c.stride);
                    }
                };
                auto issue_coef = [&](const S& c) __attribute__((always_inline)) {
                    if constexpr (uni) { kq.p = ld4(c.ka + c4 * 4); kq.q = ld4(c.kb + c4 * 4); }
                };
                // Branch-free: an item slot past the end of the tile is a DUPLICATE of the tile's last pixel (same source
                // row, same channel quad as the thread that owns it, hence the same bytes to the same LDS address).  A
                // branch per item made six separate basic blocks: the six dependency chains (affine -> exp -> rcp -> split)
                // ran one after the other, and next to an MFMA wave a single serial chain gets ~2 issue slots per MFMA
                // (measured 7.9k cycles for ~400 instructions per chunk -- the loaders paced the whole block).
                S s2;                                               // chunk whose raw rows are in flight / in registers
                auto finish = [&](int slot, int j) __attribute__((always_inline)) {
                    const bool ok = (valid1 >> j) & 1u;
                    f32x4 v = transform(araw[j][0], ok);
                    if constexpr (NS == 4)
                        v = 0.25f * (v + transform(araw[j][1], ok) + transform(araw[j][2], ok) + transform(araw[j][3], ok));
                    // the mask of element (source row, concat channel), as apply_pro forms it (a padding pixel is 0 either way);
                    // s2 / rows2 still describe the chunk being transformed: stage() advances them after finish_all()
                    if constexpr (DROP)
                        v = sgd_drop4(v, a.drop_p, a.drop_seed, (long)rows2[j] * cin + s2.chunk * KC + c4 * 4);
                    if (!ABL(1024)) lds_store_act<PREC>(As + (size_t)(slot % NA) * a_floats + (size_t)pixj[j] * LDA, c4, v);
                    else KEEP_LIVE(v);
                };
                // ---- ONE barrier per chunk; the loaders run two chunks ahead of the compute waves in LDS (ring of 3)
                // and three ahead in global memory: in period q they transform chunk q+2 (requested in period q-1) into
                // slot (q+2) % 3 and request chunk q+3.  Prologue: chunks 0 and 1 staged before barrier 0.
                auto stage = [&](int slot) __attribute__((always_inline)) {                        // transform the chunk under s2, request the next one
                    valid1 = valid2;
                    auto finish_all = [&]() {
#pragma unroll
                        for (int j = 0; j < AJ; ++j) finish(slot, j);
                    };
                    PROBE_EPI(finish_all());                        // probe build: cycles of the transform phase
                    const int kprev = s2.k;
                    s2 = advance(s2);
                    if (s2.k != kprev) load_rows(s2.k);
                    issue_coef(s2);
#pragma unroll
                    for (int j = 0; j < AJ; ++j) issue_item(s2, j);
                };
                s2.k = 0; s2.chunk = cbeg(0); s2.img0 = tile_at(g, lin_of(0), BN, TW, TH).img0; fill(s2);
                load_rows(0);
                issue_coef(s2);
#pragma unroll
                for (int j = 0; j < AJ; ++j) issue_item(s2, j);
                stage(0);                                           // chunk 0 -> slot 0, request chunk 1
                stage(1);                                           // chunk 1 -> slot 1, request chunk 2
                SYNC();                                             // barrier 0: the compute waves' fragment
                                                                    // prefetch runs DEPTH stages into the next chunk
                auto periods = [&](auto nqc) __attribute__((always_inline)) {
                    for (int q = 0; q < Q; ++q) {
                        drain(q, nqc);                              // DEFER: a slice of the previous tile's epilogue
                        stage((q + 2) % NA);                        // chunk q+2 -> slot (q+2) % 3, request chunk q+3
                        // table of the tile that chunk q+4 opens (read by load_rows one period later)
                        if (q + 4 < Q && (q + 4) % nchunks == 0) build_pixtab((q + 4) / nchunks, lt, LT);
                        SYNC();                                     // barrier q+1
                    }
                };
                // (sgd_igemm never pairs the fused average pool -- 96 registers of raw rows -- with the loader-side epilogue)
                if constexpr (NS == 4) periods(std::integral_constant<int, 0>());
                else with_nq(periods);
            };
            if (uni_rt) {
                if (a.resample == SGD_RS_AVGPOOL2) go(std::true_type(), std::true_type(), std::false_type());
                else if (drop_rt) go(std::true_type(), std::false_type(), std::true_type());
                else go(std::true_type(), std::false_type(), std::false_type());
            } else {
                if (a.resample == SGD_RS_AVGPOOL2) go(std::false_type(), std::true_type(), std::false_type());
                else go(std::false_type(), std::false_type(), std::false_type());
            }
            PROBE_END(1);
            return;
        }
    } else {
        // =================================================================================
        // lean loader, 1x1 convs / linears (skip connections, attention qkv / proj_out): every K step needs a fresh
        // 128 x 32 input tile, so each of the 256 loader threads moves four input quads per step through a 3-deep
        // register ring (requested 3 steps before they are staged).
        // Lean cases only: no prologue, or a per-image GroupNorm affine whose image boundaries fall on tile
        // boundaries (rows_per_n % 128 == 0), so the coefficients of a tile are one (n, channel quad) vector.
        // =================================================================================
        const bool tile_uni_rt = a.pro == SGD_PRO_AFFINE_NC && a.rows_per_n % BM == 0;
        const bool ln_rt = a.pro == SGD_PRO_LN_ROW;           // per-row (mean, rstd) + per-channel gamma / beta
        const bool leanf = VEC && (a.pro == SGD_PRO_NONE || tile_uni_rt || ln_rt) && a.drop_p == 0.f && cin % KC == 0
                           && (a.c1 == 0 || a.c0 % KC == 0) && !DBG(3);
        // straight-line instantiations per (prologue, SiLU) -- same reason as the conv loader above: with run-time mode
        // flags every item was a chain of small blocks, and on the 1x1 launches the loader IS the critical path
        auto flat_loader = [&](auto modec, auto siluc) {
            constexpr int MODE = decltype(modec)::value;              // 0 none, 1 per-image GroupNorm affine, 2 LayerNorm rows
            constexpr bool SILU = decltype(siluc)::value;
            constexpr bool tile_uni = MODE == 1, ln = MODE == 2;
            constexpr int AI = BM * 8 / A_THREADS;                    // input quads per thread per step (4)
            constexpr int AROWS = A_THREADS / 8;                      // rows covered by one pass of the loader threads (32)
            const int lt = tid - NCOMP;
            const int c4 = lt & 7;
            const int arow = lt >> 3;                                 // + AROWS * j
            typedef std::integral_constant<int, 0> R0;
            typedef std::integral_constant<int, 1> R1;
            typedef std::integral_constant<int, 2> R2;
            f32x4 araw[NB_RING][AI];
            Coef kq[NB_RING];
            float2 rst[NB_RING][AI];                                  // LayerNorm row statistics of the items
            struct Cur { int k, chunk; int m0; const float* ka; const float* kb; };
            auto open_tile = [&](Cur& c) {                           // per-tile scalars
                const Tile T = tile_at(g, lin_of(c.k), BN, TW, TH);
                c.m0 = (int)T.m0;
                const long ko = tile_uni ? (long)(c.m0 / a.rows_per_n) * cin : 0;
                // coefficient quads of a chunk: GroupNorm a / b of the tile's image, or LayerNorm gamma / beta;
                // no prologue (or no beta): harmless bytes of the input instead of a branch around the load
                c.ka = tile_uni ? a.pa + ko + c4 * 4 : (ln ? a.pb + c4 * 4 : a.x0);
                c.kb = tile_uni ? a.pb + ko + c4 * 4 : ((ln && a.pc) ? a.pc + c4 * 4 : a.x0);
            };
            // (the cursors count 32-channel PLANES: FG per chunk)
            auto advance = [&](Cur& c) {
                if (++c.chunk == cend(c.k) * FG) {
                    if (c.k + 1 < ntiles) { ++c.k; c.chunk = cbeg(c.k) * FG; open_tile(c); }
                    else c.chunk = cend(c.k) * FG - 1;
                }
            };
            Cur ci, cf;
            ci.k = 0; ci.chunk = cbeg(0) * FG; open_tile(ci);
            cf = ci;
            auto issue = [&](auto rc) {                              // request the chunk under the issue cursor
                constexpr int R = decltype(rc)::value;
                const int ch = ci.chunk * KC;
                const float* src;
                int stride;
                if (ch < a.c0) { src = a.x0 + ch; stride = a.c0; }
                else { src = a.x1 + (ch - a.c0); stride = a.c1; }
                if constexpr (MODE != 0) {
                    kq[R].p = ld4(ci.ka + ch);
                    kq[R].q = ld4(ci.kb + ((tile_uni || a.pc) ? ch : 0));
                }
#pragma unroll
                for (int j = 0; j < AI; ++j) {
                    int row = ci.m0 + arow + j * AROWS;
                    row = row < M ? row : M - 1;
                    araw[R][j] = ld4(src + (long)row * stride + c4 * 4);
                    if constexpr (ln) rst[R][j] = *reinterpret_cast<const float2*>(a.pa + (long)row * 2);
                }
                advance(ci);
            };
            auto finish = [&](int slot, auto rc, int sub = 0) {      // stage the plane under the finish cursor
                constexpr int R = decltype(rc)::value;
#pragma unroll
                for (int j = 0; j < AI; ++j) {
                    f32x4 v = araw[R][j];
                    if constexpr (tile_uni) v = v * kq[R].p + kq[R].q;
                    if constexpr (ln) {
                        v = (v - rst[R][j].x) * rst[R][j].y * kq[R].p;
                        if (a.pc) v += kq[R].q;
                    }
                    if constexpr (SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = sgd_silu(v[e]);
                    }
                    if (cf.m0 + arow + j * AROWS >= M) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    lds_store_act<PREC>(As + (size_t)slot * a_floats + (size_t)(sub * BM + arow + j * AROWS) * LDA, c4, v);
                }
                advance(cf);
            };
            if constexpr (FG == 1) {
                issue(R0());
                issue(R1());
                finish(0, R0());
                finish(1, R1());
                issue(R2());
                issue(R0());
                issue(R1());
                SYNC();
                auto body = [&](int slot, auto rc) {
                    finish(slot, rc);
                    issue(rc);
                    SYNC();
                };
                int step = 0;
                for (; step + 3 <= S; step += 3) {
                    body(2, R2());
                    body(0, R0());
                    body(1, R1());
                }
                if (step < S) {
                    body(2, R2());
                    if (step + 1 < S) body(0, R0());
                }
            } else {
                static_assert(FG <= 2, "two planes per chunk");
                // plane p = FG * chunk + sub lives in register set p % 3 (requested three planes before it is staged) and
                // goes to ring slot chunk % 3; chunks 0 and 1 are staged before barrier 0, period q stages chunk q + 2
                issue(R0());
                issue(R1());
                issue(R2());
                finish(0, R0(), 0); issue(R0());            // plane 0, request plane 3
                finish(0, R1(), 1); issue(R1());            // plane 1, request plane 4
                finish(1, R2(), 0); issue(R2());            // plane 2, request plane 5
                finish(1, R0(), 1); issue(R0());            // plane 3, request plane 6
                SYNC();                                     // barrier 0
                auto period = [&](int slot, auto ra, auto rb) {      // planes 2q + 4 (set ra) and 2q + 5 (set rb)
                    finish(slot, ra, 0); issue(ra);
                    finish(slot, rb, 1); issue(rb);
                    SYNC();
                };
                int q = 0;
                for (; q + 3 <= Q; q += 3) {
                    period(2, R1(), R2());
                    period(0, R0(), R1());
                    period(1, R2(), R0());
                }
                if (q < Q) {
                    period(2, R1(), R2());
                    if (q + 1 < Q) period(0, R0(), R1());
                }
            }
        };
        if (FG > 1 && !leanf) __builtin_trap();           // sgd_igemm picks the multi-plane instance for lean launches only
        if (leanf) {
            typedef std::integral_constant<int, 0> M0;
            typedef std::integral_constant<int, 1> M1;
            typedef std::integral_constant<int, 2> M2;
            if (tile_uni_rt) { if (a.pro_silu) flat_loader(M1(), std::true_type()); else flat_loader(M1(), std::false_type()); }
            else if (ln_rt) { if (a.pro_silu) flat_loader(M2(), std::true_type()); else flat_loader(M2(), std::false_type()); }
            else { if (a.pro_silu) flat_loader(M0(), std::true_type()); else flat_loader(M0(), std::false_type()); }
            PROBE_END(1);
            return;
        }
    }
    // =====================================================================================
    // general input-tile loader (strided / big-halo / partial-chunk / dropout cases).  Branch-free steady state so
    // the compiler's in-order vmcnt bookkeeping stays exact (a conditional load anywhere degrades every later wait
    // to vmcnt(0) == full memory latency per K step): raw loads of chunk q+1 requested at the taps of chunk q,
    // transformed and written to LDS one chunk later.  Out-of-range prefetches are clamped to the last valid chunk
    // (harmless duplicates written to ring slots nobody reads any more) instead of being branched around.
    // =====================================================================================
    // -------------------------------------------------------------------- A loader
    const int lt = tid - NCOMP;
    const int c4 = lt & 7;                        // this thread's channel quad inside every chunk
    const int items = g.pix * 8;                  // float4 items of one A tile
    // row entry of tile-row `pix`: CONV reads the tile table; FLAT rows are m0 + pix (tab carries m0)
    struct TabRef { const int2* tab; long m0; };
    auto tabref = [&](int k) {
        TabRef t;
        t.tab = pixtab + (size_t)(k & 3) * g.pix;
        t.m0 = CONV ? 0 : tile_at(g, lin_of(k), BN, TW, TH).m0;
        return t;
    };
    auto entry = [&](const TabRef& t, int pix) -> int2 {
        if (CONV) return t.tab[pix];
        int2 e;
        const long row = t.m0 + pix;
        e.x = row < M ? (int)row : -1;
        e.y = (a.pro == SGD_PRO_AFFINE_NC) ? (int)(row / a.rows_per_n) : 0;
        return e;
    };
    auto item_sync = [&](float* abuf, const TabRef& tab, int idx, int kc0) {
        const int pix = idx >> 3;
        const int c = kc0 + c4 * 4;
        const int2 e = entry(tab, pix);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e.x >= 0 && c < cin) {
            if (CONV && a.resample == SGD_RS_AVGPOOL2) {
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        long r = (long)e.x + dy * a.wi + dx;
                        v += apply_pro(a, load_raw<VEC>(a, r, c), load_coef<VEC>(a, e.y, r, c), c, r);
                    }
                v = v * 0.25f;
            } else {
                v = apply_pro(a, load_raw<VEC>(a, e.x, c), load_coef<VEC>(a, e.y, e.x, c), c, e.x);
            }
        }
        lds_store_act<PREC>(abuf + (size_t)pix * LDA, c4, v);
    };
    // everything at once: first chunk of the stream, avg-pool, big halo tiles
    auto stage_A_sync = [&](int slot, int q) __attribute__((always_inline)) {
        q = q < Q ? q : Q - 1;
        const int k = q / nchunks, chunk = q - k * nchunks + cbeg(k);
        float* abuf = As + (size_t)(slot % NA) * a_floats;
        const TabRef tab = tabref(k);
        for (int idx = lt; idx < items; idx += A_THREADS) item_sync(abuf, tab, idx, chunk * KC);
    };
    auto finish_item = [&](int slot, const TabRef& tab, int chunk, int idx, f32x4 raw, bool kuse, const Coef& kuni) __attribute__((always_inline)) {
        // (coefficients by value + flag: a `cond ? &k : nullptr` pointer kept the struct in scratch memory, and a kernel
        // with any scratch pays for it on every launch)
        float* abuf = As + (size_t)(slot % NA) * a_floats;
        const int c = chunk * KC + c4 * 4;
        if (idx < items) {
            const int pix = idx >> 3;
            const int2 e = entry(tab, pix);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < cin && e.x >= 0) v = apply_pro(a, raw, kuse ? kuni : load_coef<VEC>(a, e.y, e.x, c), c, e.x);
            lds_store_act<PREC>(abuf + (size_t)pix * LDA, c4, v);
        }
    };
    // branch-free: out-of-tile items, padding pixels and channels past cin load a valid (clamped) address
    // and are zeroed in finish_item
    auto raw_item = [&](const TabRef& tab, int idx, int c) -> f32x4 {
        const int ii = idx < items ? idx : items - 1;
        const int2 e = entry(tab, ii >> 3);
        return load_raw<VEC>(a, e.x >= 0 ? e.x : 0, c < cin ? c : 0);
    };

    if (CONV) {
        // ONE barrier per chunk, loaders two chunks ahead in LDS (see the lean loader above)
        constexpr int AJ = (FAST_PIX * 8 + A_THREADS - 1) / A_THREADS;   // item slots per thread (6)
        stage_A_sync(0, 0);
        if (DBG(1)) {
            for (int q = 0; q <= Q; ++q) SYNC();
        } else if (g.fast_a && a.resample != SGD_RS_AVGPOOL2) {
            // Split-phase staging: the raw row quads of chunk q+3 are requested in period q and transformed into LDS
            // in period q+1, so no load is consumed sooner than a whole chunk (9 K steps) after its issue and none
            // sits behind a branch (the in-order vmcnt stays exact).
            const bool uni = VEC && g.nb == 1 && a.pro == SGD_PRO_AFFINE_NC;
            const bool kshared = uni || a.pro == SGD_PRO_NONE;
            f32x4 araw[AJ];
            Coef kq;
            struct Ctx { const int2* tab; int c; int chunk; long ko; };
            auto ctx_of = [&](int q) {
                q = q < Q ? q : Q - 1;
                Ctx cx;
                const int k = q / nchunks;
                cx.chunk = q - k * nchunks + cbeg(k);
                cx.tab = pixtab + (size_t)(k & 3) * g.pix;
                cx.c = cx.chunk * KC + c4 * 4;
                // per-image GroupNorm coefficients of this thread's channel quad (tile = one image)
                cx.ko = (long)tile_at(g, lin_of(k), BN, TW, TH).img0 * cin + (cx.c < cin ? cx.c : 0);
                return cx;
            };
            Ctx cx = ctx_of(1);
            auto request = [&]() __attribute__((always_inline)) {
                TabRef tr;
                tr.tab = cx.tab;
                tr.m0 = 0;
                // other prologues read 32 harmless bytes of the input instead of branching around the loads
                kq.p = ld4(uni ? a.pa + cx.ko : a.x0);
                kq.q = ld4(uni ? a.pb + cx.ko : a.x0);
#pragma unroll
                for (int j = 0; j < AJ; ++j) araw[j] = raw_item(tr, lt + j * A_THREADS, cx.c);
            };
            auto stage = [&](int slot, int qnext) __attribute__((always_inline)) {
                TabRef tr;
                tr.tab = cx.tab;
                tr.m0 = 0;
#pragma unroll
                for (int j = 0; j < AJ; ++j)
                    finish_item(slot, tr, cx.chunk, lt + j * A_THREADS, araw[j], kshared, kq);
                cx = ctx_of(qnext);
                request();
            };
            request();
            stage(1, 2);
            SYNC();                                                  // barrier 0 (chunks 0 and 1 staged)
            with_nq([&](auto nqc) __attribute__((always_inline)) {
                for (int q = 0; q < Q; ++q) {
                    drain(q, nqc);
                    stage(q + 2, q + 3);
                    if (q + 4 < Q && (q + 4) % nchunks == 0) build_pixtab((q + 4) / nchunks, lt, A_THREADS);
                    SYNC();
                }
            });
        } else {
            stage_A_sync(1, 1);
            SYNC();                                                  // barrier 0 (chunks 0 and 1 staged)
            with_nq([&](auto nqc) __attribute__((always_inline)) {
                for (int q = 0; q < Q; ++q) {
                    drain(q, nqc);
                    stage_A_sync(q + 2, q + 2);
                    if (q + 4 < Q && (q + 4) % nchunks == 0) build_pixtab((q + 4) / nchunks, lt, A_THREADS);
                    SYNC();
                }
            });
        }
        PROBE_END(1);
        return;
    } else {
        constexpr int AJ = (BM * 8 + A_THREADS - 1) / A_THREADS;      // 128 rows * 8 quads / 384 threads (3)
        f32x4 araw[NB_RING][AJ];                      // A(q) raw rows live in araw[q % 3]
        stage_A_sync(0, 0);
        if (Q > 1) stage_A_sync(1, 1);
        auto issue = [&](int q, auto rc) {
            constexpr int R = decltype(rc)::value;
            q = q < Q ? q : Q - 1;
            const int k = q / nchunks, chunk = q - k * nchunks + cbeg(k);
            const TabRef tab = tabref(k);
            const int c = chunk * KC + c4 * 4;
#pragma unroll
            for (int j = 0; j < AJ; ++j) araw[R][j] = raw_item(tab, lt + j * A_THREADS, c);
        };
        auto finish = [&](int slot, auto rc) {
            constexpr int R = decltype(rc)::value;
            const int q = slot < Q ? slot : Q - 1;
            const int k = q / nchunks, chunk = q - k * nchunks + cbeg(k);
            const TabRef tab = tabref(k);
            Coef knone;
            knone.p = f32x4{0.f, 0.f, 0.f, 0.f};
            knone.q = knone.p;
#pragma unroll
            for (int j = 0; j < AJ; ++j)
                finish_item(slot, tab, chunk, lt + j * A_THREADS, araw[R][j], a.pro == SGD_PRO_NONE, knone);
        };
        typedef std::integral_constant<int, 0> R0;
        typedef std::integral_constant<int, 1> R1;
        typedef std::integral_constant<int, 2> R2;
        issue(2, R2());
        issue(3, R0());
        issue(4, R1());
        SYNC();
        auto body = [&](int step, auto rc) {
            finish(step + 2, rc);               // chunks past the end: clamped duplicates into ring slots nobody reads
            issue(step + 5, rc);
            SYNC();
        };
        int step = 0;
        for (; step + 3 <= S; step += 3) {
            body(step, R2());
            body(step + 1, R0());
            body(step + 2, R1());
        }
        if (step < S) {
            body(step, R2());
            if (step + 1 < S) body(step + 1, R0());
        }
        PROBE_END(1);
        return;
    }
}

// =========================================================================================
// epilogue of a tile (compute waves)
// =========================================================================================
template <int BN, int PREC, bool VEC, int TAPS, bool DEFER>
template <int RES, bool PART>
__device__ __forceinline__ void IgemmBlock<BN, PREC, VEC, TAPS, DEFER>::epilogue(const AccV (&acc)[RB][CBN], const Tile& T, int wm, int wn,
                                                                                 int lane_e, int cb, float wsk, const char* part_base,
                                                                                 int nparts) const {
    const int li = lane & 31;
    // per M block: output row / residual row of this lane's pixel
    bool okm[RB];
    // rows as 32-bit indices, addresses formed at the use: as 64-bit pointers the 16 (M16) row pointers were the
    // epilogue's spill traffic, and a kernel with scratch costs small launches ~3 us each (C1: 29 -> 25 ms)
    int orw[RB], rrw[RB];
    // the lane's pixel goes through an opaque register: its tile-invariant row arithmetic (tx, ty, image of every row
    // block) is then redone per tile -- ~20 integer instructions -- instead of living in ~10 registers across the
    // K loop, which the allocator spilled and reloaded behind the stores (vmcnt is in order)
    const int plie = pixel_of_lane(lane_e);
#pragma unroll
    for (int mt = 0; mt < RB; ++mt) {
        const int row = wm * WM + mt * RBH + plie;
        int orow, n = 0, oy = 0, ox = 0;
        if (CONV) {
            const int tx = row & (TW - 1), ty = (row >> g.tw_l2) & (TH - 1), nb = row >> (g.tw_l2 + g.th_l2);
            n = T.img0 + nb; oy = T.ty0 + ty; ox = T.tx0 + tx;
            okm[mt] = nb < g.nb && n < a.n;
            orow = (n * a.ho + oy) * a.wo + ox;
        } else {
            orow = (int)T.m0 + row;
            okm[mt] = orow < M;
        }
        if (!okm[mt]) orow = 0;                      // keep the addresses valid; the lane is masked below
        int rrow = orow;
        if (RES == 2) rrow = (n * a.ho * 2 + 2 * oy) * (a.wo * 2) + 2 * ox;
        if (RES == 3) rrow = (n * (a.ho >> 1) + (oy >> 1)) * (a.wo >> 1) + (ox >> 1);
        if (!okm[mt]) rrow = 0;
        if (!CONV && a.orows_in > 0)
            orow = (int)((unsigned)orow / (unsigned)a.orows_in) * a.orows_out + a.orow_off
                   + (int)((unsigned)orow % (unsigned)a.orows_in);
        orw[mt] = orow;
        rrw[mt] = rrow;
    }
    const bool vec = ((a.cout | a.y_ld) & 3) == 0;
    if (vec) {
        // cout % 4 == 0: 16-byte quads.  Quad-outer / row-inner: the residual loads of both rows are in flight
        // together, and the GroupNorm statistics of a quad (args.stats) live in 8 registers at a time.
        float* sp = nullptr;
        if (a.stats) {
            int n_img, part;
            if (CONV) {
                n_img = T.img0;
                part = ((T.ty0 >> g.th_l2) * g.tiles_x + (T.tx0 >> g.tw_l2)) * (BM / WM) + wm;
            } else {
                n_img = (int)(T.m0 / a.rows_per_n);
                part = (int)((T.m0 % a.rows_per_n) / BM) * (BM / WM) + wm;
            }
            sp = a.stats + ((long)n_img * g.sparts + part) * 2 * a.cout;
        }
        {
            // Column slots cs = (column block nt, quad gq) of the lane's output: c = cb + nt * CBW + gq * 8.
            // load phase: bias + residual of a batch of slots x row blocks -- independent loads in flight (the compiler
            // may not hoist them itself: y and res could alias) -- then the stores.  Every batch is one exposed memory
            // round trip for a wave that has nothing else to issue, so batches are as big as the registers allow: with
            // the 16x16x32 form (operands not carried across the epilogue) ALL of a tile's 16 residual quads.
            constexpr int CS = CBN * QPB;                                      // column slots per lane
            constexpr int QB = M16 ? ((RES == 2 || PART) ? 1 : (CBN > 2 ? 2 : CBN)) : ((RES == 2 || NT > 1) ? 1 : 2);   // slots per batch
            constexpr int RBB = M16 ? (RES == 2 || RB < 4 ? 2 : ((PART || CBN > 2) ? 4 : RB)) : RB;   // row blocks per batch
            static_assert(RB % RBB == 0 && CS % QB == 0, "batches must tile the wave's rows and columns");
#pragma unroll
            for (int q0 = 0; q0 < CS; q0 += QB) {
            f32x4 s1[QB], s2[QB];
#pragma unroll
            for (int i = 0; i < QB; ++i) s1[i] = s2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mb0 = 0; mb0 < RB; mb0 += RBB) {
            // (opaque per batch: the addresses of a batch's rows are formed here, not hoisted as RB 64-bit values)
#pragma unroll
            for (int mt = mb0; mt < mb0 + RBB; ++mt) asm volatile("" : "+v"(orw[mt]), "+v"(rrw[mt]));
            f32x4 rv[QB][RBB];
            f32x4 pv[QB][RBB];
            if constexpr (PART) {
#pragma unroll
                for (int cs = q0; cs < q0 + QB; ++cs)
#pragma unroll
                    for (int mt = mb0; mt < mb0 + RBB; ++mt) {
                        const char* pp = part_base + (size_t)((mt * CS + cs) * NCOMP) * 16;
                        f32x4 sum = *reinterpret_cast<const f32x4*>(pp);
                        for (int pi = 1; pi < nparts; ++pi) sum += *reinterpret_cast<const f32x4*>(pp + pi * SLAB);
                        pv[cs - q0][mt - mb0] = sum;
                    }
            }
#pragma unroll
            for (int cs = q0; cs < q0 + QB; ++cs) {
                const int c = cb + (cs / QPB) * CBW + (cs % QPB) * 8;
                const int cl = c < a.cout ? c : 0;    // clamped: the quad is skipped below
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (bias_lds) bv = *reinterpret_cast<const f32x4*>(bias_s + cl);
                else if (a.bias && !ABL(2)) bv = ld4(a.bias + cl);
#pragma unroll
                for (int mt = mb0; mt < mb0 + RBB; ++mt) {
                    f32x4& r = rv[cs - q0][mt - mb0];
                    r = bv;
                    if ((RES == 1 || RES == 3) && !ABL(2)) r += ld4(a.res + (long)rrw[mt] * a.cout + cl);
                    if (RES == 2 && !ABL(2)) {
                        const long rw = (long)a.wo * 2 * a.cout;
                        const float* rp = a.res + (long)rrw[mt] * a.cout;
                        r += 0.25f * (ld4(rp + cl) + ld4(rp + a.cout + cl) + ld4(rp + rw + cl) + ld4(rp + rw + a.cout + cl));
                    }
                }
            }
#pragma unroll
            for (int cs = q0; cs < q0 + QB; ++cs) {
                const int c = cb + (cs / QPB) * CBW + (cs % QPB) * 8;
                if (c >= a.cout) continue;            // uniform within a lane half / a row of 16 lanes
#pragma unroll
                for (int mt = mb0; mt < mb0 + RBB; ++mt) {
                    f32x4 v = accq(acc, mt, cs / QPB, cs % QPB);
                    if constexpr (PART) v += pv[cs - q0][mt - mb0];
                    v = v * wsk + rv[cs - q0][mt - mb0];
                    if (okm[mt]) {
                        if (!(DBG(8)) && !ABL(1)) *reinterpret_cast<f32x4*>(a.y + (long)orw[mt] * a.y_ld + c) = v;
                        else KEEP_LIVE(v);
                        s1[cs - q0] += v;
                        s2[cs - q0] += v * v;
                    }
                }
            }
            }
            if (sp && !DBG(256) && !ABL(4)) {
#pragma unroll
                for (int cs = q0; cs < q0 + QB; ++cs) {
                    const int c = cb + (cs / QPB) * CBW + (cs % QPB) * 8;
                    if (c >= a.cout) continue;
                    // sum over the pixel lanes that share this lane's channels: the 32 lanes of a lane half (4 DPP steps
                    // inside a row of 16, then across rows) or, M16, the 16 lanes of a DPP row
                    f32x4 t1 = s1[cs - q0], t2 = s2[cs - q0];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        t1[j] = M16 ? row16_sum(t1[j]) : half_wave_sum_hi(t1[j]);
                        t2[j] = M16 ? row16_sum(t2[j]) : half_wave_sum_hi(t2[j]);
                    }
                    if (M16 ? (lane & 15) == 0 : li == 16) {
                        *reinterpret_cast<f32x4*>(sp + c) = t1;
                        *reinterpret_cast<f32x4*>(sp + a.cout + c) = t2;
                    }
                }
            }
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < RB; ++mt) {
            if (!okm[mt]) continue;
            float* yp = a.y + (long)orw[mt] * a.y_ld;
            const float* rp = RES ? a.res + (long)rrw[mt] * a.cout : nullptr;
#pragma unroll
            for (int nt = 0; nt < CBN; ++nt)
#pragma unroll
                for (int gq = 0; gq < QPB; ++gq)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = cb + nt * CBW + gq * 8 + j;
                        if (c >= a.cout) continue;
                        float x = acc[mt][nt][(M16 ? 0 : gq * 4) + j];
                        if constexpr (PART) {
                            const float* pp = reinterpret_cast<const float*>(part_base + (size_t)(((mt * CBN + nt) * QPB + gq) * NCOMP) * 16) + j;
                            for (int pi = 0; pi < nparts; ++pi) x += pp[pi * (SLAB / 4)];
                        }
                        x = x * wsk + (bias_lds ? bias_s[c] : (a.bias ? a.bias[c] : 0.f));
                        if (RES == 1 || RES == 3) x += rp[c];
                        if (RES == 2) {
                            const long rw = (long)a.wo * 2 * a.cout;
                            x += 0.25f * (rp[c] + rp[a.cout + c] + rp[rw + c] + rp[rw + a.cout + c]);
                        }
                        yp[c] = x;
                    }
        }
    }
}

template <int BN, int PREC, bool VEC, int TAPS, bool DEFER>
__device__ __forceinline__ void IgemmBlock<BN, PREC, VEC, TAPS, DEFER>::compute_role() {
    PROBE_BEGIN();
    // =========================================================================================
    // compute role
    // =========================================================================================
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // per-lane LDS float offsets of this wave's MFMA row tiles (row -> halo pixel of tap (0,0))
    const int pli = pixel_of_lane(lane);
    int aoff[RB];
#pragma unroll
    for (int mt = 0; mt < RB; ++mt) {
        int r = wm * WM + mt * RBH + pli;
        int p;
        if (CONV) {
            int tx = r & (TW - 1), ty = (r >> g.tw_l2) & (TH - 1), nb = r >> (g.tw_l2 + g.th_l2);
            p = (nb * g.hh + ty * s) * g.hw + tx * s;
        } else {
            p = r;
        }
        aoff[mt] = p * LDA;
    }
    // weight fragments: wave (wm, wn) needs N blocks wn*NT .. wn*NT+NT-1 of its tile; consecutive K steps of the stream
    // [chunk][tap] are `wstep` bytes apart (all N blocks of the layer for that step)
    const size_t wstep = (size_t)(a.cout_p >> 5) * WUNIT;
    // (the lane's offset inside a unit is re-derived per tile from the hardware lane index: as a 64-bit per-lane pointer held
    // from here on it was the kernel's last spilled value, and a kernel with any scratch pays ~3 us on every launch)
    auto wtile_of = [&](int k) __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        const int loff = M16 ? (l >> 5) * 2048 + (((l >> 4) & 1) * 32 + (l & 15)) * 16 : l * 16;
        return reinterpret_cast<const char*>(a.w) + (size_t)(wn * NT + (tile_at(g, lin_of(k), BN, TW, TH).n0c >> 5)) * WUNIT + loff;
    };

    AccV acc[RB][CBN];
    // 2^-k of the packed weights (exact); once per block, held in a SCALAR register: as a vector register it was the value the
    // allocator spilled, and its reload in front of every output quad waited (vmcnt is in order) for the previous store
    const float wsi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
        __builtin_bit_cast(int, a.w_scale_inv ? *a.w_scale_inv : 1.f)));

    // The K loop is a software pipeline over STAGES: a stage = (K sub-step ks, 32-row block mt) multiplies ONE input
    // fragment unit (2 ds_read_b128) into the wave's NT accumulator tiles (3 NT MFMAs).  The units live in a register
    // ring of RING slots and are requested DEPTH stages before they are used -- across K steps, chunk seams and tile
    // seams alike -- so the wave holds 8 registers per stage in flight instead of a double-buffered set of all MT row
    // blocks (64 registers): that is what leaves room for the 128 accumulator registers of the 64-column wave tile.
    // The weight fragments of sub-step ks are reloaded (for the next K step) right after their last use in this one:
    // they have (NKS - 1) * MT stages (>= 768 matrix-pipe cycles) to return from L2.  All waits are the compiler's
    // (plain loads: exact counted s_waitcnt); the sched_group_barriers only pin the issue order.
    // M16 (16x16x32 MFMA): a K step is ONE MFMA deep, so its stages run column block by column block over ALL row blocks:
    // stage (cb, rb) = 3 MFMAs.  The RB input units of the step stay in registers (64 for the 128-row wave tile); unit rb
    // is reloaded for the NEXT step right after its last use (column block CBN - 1), the weights of column block cb right
    // after stage (cb, RB - 1): every reload has RB - 1 .. RB stages (>= 336 matrix-pipe cycles) to land, with single
    // buffers and compile-time register indices.
    constexpr int STAGES = M16 ? RB * CBN : NKS * MT;    // per K step
    constexpr int RING = M16 ? RB : (STAGES >= 4 ? 4 : 2);   // STAGES % RING == 0: a K step always starts at ring slot 0
    constexpr int DEPTH = RING - 1;
    static_assert(STAGES % RING == 0, "ring position must be a compile-time constant inside a K step");
    std::conditional_t<M16, typename Frag16<PREC, 1>::AU, typename FragT::AU> ring[RING];
    typename FragT::B fb[NKS];
    typename Frag16<PREC, 1>::B fb16[CBN];         // M16: weights of one 16-column block each
    const int rowstep = g.hw * LDA;                // LDS floats between halo rows

    auto tap_off = [&](int tap) { return CONV ? (tap / 3) * rowstep + (tap % 3) * LDA : tap * (BM * LDA); };
    // one K step: `acur` holds this step's chunk, `anext` the chunk the prefetches run into when `seam` (last tap)
    auto do_step = [&](const float* acur, const float* anext, auto tapc, auto parc, const char* wnext) {
        constexpr int tap = decltype(tapc)::value;
        constexpr int PAR = decltype(parc)::value;
        constexpr bool seam = tap + 1 == TAPS;
        static_for<STAGES>([&](auto stc) {
            constexpr int st = decltype(stc)::value;
            constexpr int ks = M16 ? 0 : st / MT, mt = M16 ? st % RB : st % MT;
            // the unit of stage st + DEPTH: this step, or the first stages of the next one (next tap / next chunk)
            constexpr int pst = (st + DEPTH) % STAGES;
            constexpr bool wrap = st + DEPTH >= STAGES;
            const float* pbase = (wrap && seam) ? anext : acur;
            constexpr int ptap = wrap ? (seam ? 0 : tap + 1) : tap;
            if constexpr (M16) {
                constexpr int cbi = st / RB;
                const float* nbase = seam ? anext : acur;                  // next step: next tap of this chunk / next chunk
                constexpr int ntap = seam ? 0 : tap + 1;
                if (!ABL(16)) acc[mt][cbi] = Frag16<PREC, 1>::mma1(acc[mt][cbi], ring[mt], fb16[cbi]);
                if (cbi == CBN - 1 && !ABL(64)) ring[mt].load(nbase + tap_off(ntap) + aoff[mt], lane >> 4);
                if (mt == RB - 1 && !ABL(32)) fb16[cbi].load(wnext + (cbi >> 1) * WUNIT + (cbi & 1) * 256);
                (void)pbase; (void)ptap; (void)pst; (void)wrap;
            } else {
                if (!ABL(64)) ring[(st + DEPTH) % RING].load(pbase + tap_off(ptap) + aoff[pst % MT], pst / MT, lh);
                if (!ABL(16)) FragT::mma(acc[mt], ring[st % RING], fb[ks]);
                if (mt == MT - 1 && !ABL(32)) fb[ks].load(wnext, ks);
            }
            // issue order: one MFMA, ONE LDS read, ... then one MFMA, ONE weight load, ...; the remaining MFMAs last
            // (a burst of reads in front of the MFMAs fills the LDS queue and the in-order wave cannot issue its MFMAs
            // until they are accepted)
            {
                constexpr int NM = M16 ? 3 : FragT::NMMA;
                constexpr int NR = M16 ? (st / RB == CBN - 1 ? 2 : 0) : FragT::NREADS;
                constexpr int P1 = NR < NM ? NR : NM;
                constexpr int nw = M16 ? (st % RB == RB - 1 ? 2 : 0) : ((mt == MT - 1) ? FragT::NWLOADS : 0);
                constexpr int P2 = nw < NM - P1 ? nw : NM - P1;
#pragma unroll
                for (int i = 0; i < P1; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < P2; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                if constexpr (NM - P1 - P2 > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - P1 - P2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (seam) SYNC();
    };

    // first K step of the block's k-th tile (the last tile may start in the middle of its K range: balanced tail)
    auto wstart_of = [&](int k) { return wtile_of(k) + (size_t)cbeg(k) * TAPS * wstep; };
    const char* wp = wstart_of(0);                 // weight fragments of the CURRENT step
    if constexpr (M16) {
#pragma unroll
        for (int cb = 0; cb < CBN; ++cb) fb16[cb].load(wp + (cb >> 1) * WUNIT + (cb & 1) * 256);
    } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) fb[ks].load(wp, ks);
    }
    SYNC();                               // pairs with the loaders' prologue barrier: chunks 0 AND 1 are staged
#pragma unroll
    for (int st = 0; st < (M16 ? RB : DEPTH); ++st) {
        if constexpr (M16) ring[st].load(As + aoff[st % RB], lane >> 4);
        else ring[st].load(As + aoff[st % MT], st / MT, lh);
    }
    int aslot = 0;                                 // ring position of the current chunk
    // CONV: ONE barrier per 32-channel chunk (9 K steps).  The weights never pass through LDS and the input tile of a
    // chunk is immutable while its 9 taps run, so nothing inside a chunk needs the loaders: period q (between barriers
    // q and q+1) computes chunk q from ring slot q % 3 and, in its last DEPTH stages, prefetches the first fragments of
    // chunk q+1 (slot complete since barrier q: the loaders run two chunks ahead, from the prologue on), while the
    // loaders fill slot (q+2) % 3 (last read in period q-1).
    for (int k = 0; k < ntiles; ++k) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < CBN; ++j)
#pragma unroll
                for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;
        const char* const wseam = k + 1 < ntiles ? wstart_of(k + 1) : nullptr;   // first step of the next tile
        const int c_end = cend(k);
        for (int chunk = cbeg(k); chunk < c_end; ++chunk) {
            const int naslot = aslot + 1 == NA ? 0 : aslot + 1;
            const float* acur = As + (size_t)aslot * a_floats;
            const float* anext = As + (size_t)naslot * a_floats;
            // step after this chunk's last one: next chunk, next tile, or (end of the stream) the same slice again
            const char* const wlast = chunk + 1 == c_end ? (wseam ? wseam : wp + (TAPS - 1) * wstep) : nullptr;
            // taps fully unrolled: tap offsets are compile-time, so no scalar index math sits between the MFMA blocks
            static_for<TAPS>([&](auto tapc) {
                constexpr int tap = decltype(tapc)::value;
                const char* wnext = wp + wstep;
                if (tap + 1 == TAPS && wlast) wnext = wlast;
                do_step(acur, anext, tapc, std::integral_constant<int, tap & 1>(), wnext);
                wp = wnext;
            });
            aslot = naslot;
        }
        if constexpr (DEFER) {
            if (k + 1 < ntiles) {
                // Every tile but the block's last leaves through LDS: the accumulators go to the staging tile as they are
                // (tile-local row, channel) and the LOADER waves run the epilogue during the next tile's K loop (drain,
                // above).  This wave is past barrier qe (the last chunk's), the loaders finished reading the previous
                // staged tile before it, and they read this one after the next barrier.
                int plie = pli;
                asm volatile("" : "+v"(plie));             // recompute the staging addresses per tile (see the epilogue)
#pragma unroll
                for (int mt = 0; mt < RB; ++mt) {
                    float* sp = stg + (size_t)(wm * WM + mt * RBH + plie) * STG_LD + wn * WN + 4 * lh;
#pragma unroll
                    for (int q = 0; q < QPB; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[mt][0][q * 4 + j];
                        *reinterpret_cast<f32x4*>(sp + q * 8) = v;
                    }
                }
                continue;
            }
        }

        // ---- balanced tail: this tile's K range is shared with other blocks (see the schedule at the top) ----
        const char* part_base = nullptr;                               // finisher: this thread's quads in the producers' slabs
        int nparts = 0;
        float wsk = wsi;                                               // this tile's output scale (NaN: see the finisher's poll)
        int* part_cnt = nullptr;
        if (rem_lin >= 0 && k == ntiles - 1) {
            // workspace: [WORK_TILES] {arrived, consumed} counters, then one slab per (split tile, producing part);
            // slab element (mt, nt, quad q) of compute thread tid at (((mt * NT + nt) * 4 + q) * NCOMP + tid) * 16 bytes:
            // every store / load instruction of a wave moves 1 KiB of consecutive bytes
            const int gi = tail_counter(xcd, loc, nloc, split);
            int* const cnt = reinterpret_cast<int*>(a.work) + gi * 2;
            char* const slab0 = reinterpret_cast<char*>(a.work) + WORK_HEAD + (size_t)tail_slab(xcd, loc, nloc, split) * SLAB;
            if (rem_part + 1 < split) {
                // PRODUCER (R1 of the guide's publish recipe): write-through (sc1) 16-byte stores -- no release fence, no
                // write-back of the L2's other dirty lines --, every storing wave drains its stores, then signals for itself
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab0 + (size_t)rem_part * SLAB, 0, (int)SLAB, 0x00020000);
#pragma unroll
                for (int mt = 0; mt < RB; ++mt)
#pragma unroll
                    for (int nt = 0; nt < CBN; ++nt)
#pragma unroll
                        for (int q = 0; q < QPB; ++q)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, accq(acc, mt, nt, q)), rs,
                                                                   (((mt * CBN + nt) * QPB + q) * NCOMP + tid) * 16, 0, 16 /* sc1 */);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;                                                     // no epilogue; the split tile is the block's last
            }
            // FINISHER (last K range): every wave polls for itself (one lane, relaxed, sleeping), ONE agent-scope acquire
            // after the match, then plain loads of the producers' slabs (in the epilogue), added in part order
            int late = 0;
            if (lane == 0) {
                const int need = (split - 1) * (NCOMP / 64);
                int polls = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++polls >= FINISH_POLL_MAX) { late = 1; break; }
                }
                int* const status = reinterpret_cast<int*>(a.work) + WORK_STATUS_INT;
                if (late) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // A workspace that has timed out once is not trusted again until the host has zeroed it: producers of the
                // failed launch may still arrive and push a LATER launch's counters over their target early -- stale slabs,
                // wrong values that are not NaN.  While the health word is up every split tile is poisoned (loud), and the
                // host paths that own a workspace check the word at their sync points (unet._Engine.check_health).
                else if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) late = 2;
            }
            if (__builtin_amdgcn_readfirstlane(late)) wsk = __builtin_nanf("");      // bounded poll expired: poison, do not hang
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the producers' slabs are read by the epilogue, next to the residual (the accumulator registers themselves are
            // not touched here: modifying 64..128 live registers under a branch costs a copy / spill storm at the join)
            {
                int lane_p;                                  // (per tile: not a 64-bit lane pointer carried through the K loop)
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
                part_base = slab0 + (size_t)(wave * 64 + lane_p) * 16;
            }
            nparts = split - 1;
            part_cnt = cnt;
        }

        // ---- epilogue of tile k (the loaders are already staging tile k+1) ---------------------
        // The MFMAs are issued with the WEIGHT fragment as the A operand, so an accumulator tile is
        // [32 channels x 32 pixels]: a lane owns ONE pixel (lane & 31) and its 16 registers are four runs
        // of 4 consecutive channels (8g + 4*(lane>>5) + 0..3).  => 16-byte residual loads / stores, and the
        // row index math runs twice per lane instead of 32 times.
        const Tile T = tile_at(g, lin_of(k), BN, TW, TH);
        // the lane index re-read from the hardware (all lanes are active here): what the epilogue derives from the lane is
        // computed per tile -- neither it nor `lane` itself has to survive the K loop in a register
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int cb = T.n0c + wn * WN + 4 * (M16 ? lane_e >> 4 : lane_e >> 5);   // first channel of this lane's first quad (nt = 0)
        auto keep_acc = [&]() {                     // ablation builds: the accumulators stay live without an epilogue
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CBN; ++j) KEEP_LIVE(acc[i][j]);
        };
        auto run_epilogue = [&](auto partc) {
            constexpr bool PARTV = decltype(partc)::value;
            if (DBG(16) || ABL(8)) keep_acc();
            else if (!a.res) epilogue<0, PARTV>(acc, T, wm, wn, lane_e, cb, wsk, part_base, nparts);
            else if (a.res_mode == SGD_RS_NONE) epilogue<1, PARTV>(acc, T, wm, wn, lane_e, cb, wsk, part_base, nparts);
            else if (NT > 1) __builtin_trap();      // resampled residuals: 128-column tiles only (sgd_igemm picks the tile)
            else if (a.res_mode == SGD_RS_AVGPOOL2) epilogue<2, PARTV>(acc, T, wm, wn, lane_e, cb, wsk, part_base, nparts);
            else epilogue<3, PARTV>(acc, T, wm, wn, lane_e, cb, wsk, part_base, nparts);
        };
        if (nparts) {
            run_epilogue(std::true_type());
            // the counters reset themselves: the last of the finisher's waves to have read the slabs zeroes both (the next
            // launch that uses this workspace is ordered behind this one on the stream)
            if (lane == 0) {
                if (__hip_atomic_fetch_add(part_cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NCOMP / 64 - 1) {
                    __hip_atomic_store(part_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(part_cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            PROBE_EPI(run_epilogue(std::false_type()));
        }
        if constexpr (NT > 1 || M16) {
            // 64-column wave tile (128 accumulator registers) or 16x16x32 MFMAs (all RB input units resident: 80 operand
            // registers).  The operands prefetched for the next tile (weights of its first step, the first input units)
            // are NOT carried across the epilogue -- registers the epilogue needs -- but requested again here: one L2
            // round trip per tile (~1 % of a tile's K loop).
            if (k + 1 < ntiles) {
                const float* a0 = As + (size_t)aslot * a_floats;
                if constexpr (M16) {
                    // the WEIGHTS of the next tile's first step (requested by the last step, 16 registers) are carried: asked
                    // for again here they would sit behind the acknowledgement of the epilogue's 16 stores (vmcnt is in
                    // order); the input units come from LDS
#pragma unroll
                    for (int st = 0; st < RB; ++st) ring[st].load(a0 + aoff[st % RB], lane >> 4);
                } else {
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) fb[ks].load(wp, ks);
#pragma unroll
                    for (int st = 0; st < DEPTH; ++st) ring[st].load(a0 + aoff[st % MT], st / MT, lh);
                }
            }
        }
    }
    PROBE_END(0);
}

template <int BN, int PREC, bool VEC, int TAPS, bool DEFER = false>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(const KArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    IgemmBlock<BN, PREC, VEC, TAPS, DEFER> blk{ka.a, ka.g};
    if (!blk.init(smem)) return;
    blk.build_pixtab(0, blk.tid, NTHREADS);
    blk.build_pixtab(1, blk.tid, NTHREADS);
    blk.build_pixtab(2, blk.tid, NTHREADS);
    blk.build_pixtab(3, blk.tid, NTHREADS);
    // bias of the whole layer (zeros without one): the epilogue reads it from LDS
    if (blk.bias_lds)
        for (int i = blk.tid; i < ka.a.cout_p; i += NTHREADS) blk.bias_s[i] = (ka.a.bias && i < ka.a.cout && !ABL(2)) ? ka.a.bias[i] : 0.f;
    __syncthreads();
    if (blk.tid >= NCOMP) blk.loader_role();
    else blk.compute_role();
}

template <int BN, int PREC, bool VEC, int TAPS, bool DEFER = false>
int launch1(const KArgs& ka, size_t smem, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<BN, PREC, VEC, TAPS, DEFER>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const int total = ka.g.mt * ka.g.nt;
    // persistent: one block per CU walks its tiles.  The device's CU count (256 on MI355X), less what the caller keeps
    // free for another stream (args.grid_cap), in whole groups of 8 (one block per XCD and group)
    int full = device_cus();
    if (ka.a.grid_cap > 0 && ka.a.grid_cap < full) full = ka.a.grid_cap >= 8 ? (ka.a.grid_cap & ~7) : 8;
    // (tests cap the grid the same way so that small problems walk many tiles per block)
    int grid = ((total + 7) / 8) * 8;
    if (grid > full) grid = full;
    if (ka.a.work && total >= 8) grid = full;     // balanced tail: blocks without a whole tile take K parts of the last ones
    hipLaunchKernelGGL((igemm_kernel<BN, PREC, VEC, TAPS, DEFER>), dim3(grid), dim3(NTHREADS), smem, st, ka);
    const int rc = sgd_check_launch();
    // a launch that did not start leaves nothing behind, but one that faulted may have left arrival counters half-way:
    // never hand such a head to the next launch
    if (rc != SGD_OK && ka.a.work) (void)hipMemsetAsync(ka.a.work, 0, WORK_HEAD, st);
    return rc;
}

// vec: 0 scalar inputs, 1 16-byte inputs, 2 16-byte inputs + the loader-side epilogue (DEFER: 3x3, 128-column tiles, split modes)
// taps: 9 conv, 1 flat (one 32-channel plane per barrier), 2 flat with two planes per chunk (lean 16-byte launches)
template <int BN, int PREC>
int launch(const KArgs& ka, int vec, int taps, size_t smem, hipStream_t st) {
    const bool conv = taps == 9;
#ifdef SGDM_IGEMM_NOPK
    // the unit compiled without packed-f32 code generation (see the end of this file) serves the 1x1 / linear launches with
    // the LayerNorm-row prologue only: no 3x3 instance is instantiated here
    if (conv) return SGD_ERR_ARG;
    if constexpr (BN == 256) {
        return taps == 2 ? launch1<BN, PREC, true, 2>(ka, smem, st) : launch1<BN, PREC, true, 1>(ka, smem, st);
    } else {
        if constexpr (BN == 128) {
            if (taps == 2) return launch1<BN, PREC, true, 2>(ka, smem, st);
        }
        return vec ? launch1<BN, PREC, true, 1>(ka, smem, st) : launch1<BN, PREC, false, 1>(ka, smem, st);
    }
#else
    if constexpr (BN == 256) {                    // chosen for 16-byte launches only
        if (conv) return launch1<BN, PREC, true, 9>(ka, smem, st);
        return taps == 2 ? launch1<BN, PREC, true, 2>(ka, smem, st) : launch1<BN, PREC, true, 1>(ka, smem, st);
    } else {
        if constexpr (BN == 128 && PREC != SGD_PREC_F32) {
            if (conv && vec == 2) return launch1<BN, PREC, true, 9, true>(ka, smem, st);
        }
        if (conv) return vec ? launch1<BN, PREC, true, 9>(ka, smem, st) : launch1<BN, PREC, false, 9>(ka, smem, st);
        if constexpr (BN == 128) {
            if (taps == 2) return launch1<BN, PREC, true, 2>(ka, smem, st);
        }
        return vec ? launch1<BN, PREC, true, 1>(ka, smem, st) : launch1<BN, PREC, false, 1>(ka, smem, st);
    }
#endif
}

}  // namespace

#ifndef SGDM_IGEMM_PREC
#error "igemm.hip is compiled once per arithmetic mode: -DSGDM_IGEMM_PREC=0|1|2 (build.py); geometry and entry points: igemm_host.hip"
#endif
#ifdef SGDM_DEV_ONE      /* development: compile ONE kernel instance (register / asm inspection), never linked */
#define SGD_DISPATCH_BODY(P)                                                                                        \
    const KArgs& ka = *reinterpret_cast<const KArgs*>(kap); (void)bn; (void)vec; (void)taps;                         \
    return launch1<SGDM_DEV_BN, P, true, SGDM_DEV_ONE, SGDM_DEV_DEFER>(ka, smem, st);
#else
#define SGD_DISPATCH_BODY(P)                                                                                        \
    const KArgs& ka = *reinterpret_cast<const KArgs*>(kap);                                                         \
    return bn == 256 ? launch<256, P>(ka, 1, taps, smem, st)                                                         \
                     : (bn == 128 ? launch<128, P>(ka, vec, taps, smem, st) : launch<32, P>(ka, vec, taps, smem, st));
#endif
#if SGDM_IGEMM_PREC == 0
int sgd_igemm_dispatch_f32(const void* kap, int bn, int vec, int taps, size_t smem, hipStream_t st) { SGD_DISPATCH_BODY(SGD_PREC_F32) }
#elif SGDM_IGEMM_PREC == 1 && defined(SGDM_IGEMM_NOPK)
int sgd_igemm_dispatch_f16x3_nopk(const void* kap, int bn, int vec, int taps, size_t smem, hipStream_t st) { SGD_DISPATCH_BODY(SGD_PREC_F16X3) }
#elif SGDM_IGEMM_PREC == 1
int sgd_igemm_dispatch_f16x3(const void* kap, int bn, int vec, int taps, size_t smem, hipStream_t st) { SGD_DISPATCH_BODY(SGD_PREC_F16X3) }
#elif defined(SGDM_IGEMM_NOPK)
int sgd_igemm_dispatch_bf16x3_nopk(const void* kap, int bn, int vec, int taps, size_t smem, hipStream_t st) { SGD_DISPATCH_BODY(SGD_PREC_BF16X3) }
#else
int sgd_igemm_dispatch_bf16x3(const void* kap, int bn, int vec, int taps, size_t smem, hipStream_t st) { SGD_DISPATCH_BODY(SGD_PREC_BF16X3) }
#endif
