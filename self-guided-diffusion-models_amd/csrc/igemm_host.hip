// Host side of the fused implicit-GEMM conv / linear kernel (csrc/igemm.hip): launch geometry, tile / instance rules, the balanced
// tail's workspace layout hook and the extern "C" entry points sgd_igemm / sgd_igemm_stats_parts / sgd_igemm_work_* of
// include/sgdm_hip.h.  The kernel instances live in one translation unit per arithmetic mode (igemm.hip) and are reached through
// the sgd_igemm_dispatch_* functions declared in igemm_shared.h; the argument block crosses as bytes.
#include <stdint.h>

#include "igemm_shared.h"

extern "C" int sgd_abi_version(void) { return SGD_ABI_VERSION; }

static inline int pick_bn(int cout) { return (cout % 128 == 0) ? 128 : 32; }

// Column tile of a launch.  Layers of a multiple of 128 output channels take the 128-column tile -- unless the launch is SMALL
// (round 6): a persistent grid is one block per compute unit, and at small batch the low-resolution
// layers have a handful of 128-column tiles (C1: 256 -> 256 at 8 x 8, UNet batch 16: 8 tiles of 128 x 256 on 256 compute
// units, 65 us per launch under rocprofv3).  With at most a quarter of the device's compute units in 128-column tiles the launch
// runs the 32-column instance instead (four times the tiles, four waves of 32 x 32 per block; the input tile is staged once per
// 32 columns, which an idle device does not notice): measured 1.2 .. 2.2 x on 3x3 launches of 8 .. 64 such tiles and 1.1 .. 1.5 x on
// 1x1 launches, slower from 96 .. 128 tiles on (profiles/r6_ab_small_launches.txt).  The packed weights do not depend on the tile;
// the statistics slots do (sgd_igemm_stats_parts applies the same rule).  SGD_TUNE_NO_SMALL: off.
static inline int column_tile(const sgd_igemm_args& a) {
    const int bn = pick_bn(a.cout);
    if (bn != 128 || (a.tune & (SGD_TUNE_NO_SMALL | SGD_TUNE_BN128 | SGD_TUNE_BN256))) return bn;
    const long rows = a.mode == SGD_MODE_CONV3 ? (long)a.n * a.ho * a.wo : (long)a.m;
    const long t128 = ((rows + BM - 1) / BM) * ((a.cout + 127) / 128);
    return t128 * 4 <= device_cus() ? 32 : 128;
}

static int make_geo(const sgd_igemm_args& a, Geo& g, int bn, int& na, int fg = 1) {
    if (a.c0 <= 0 || a.c1 < 0 || a.cout <= 0 || a.y_ld < a.cout) return SGD_ERR_ARG;
    if (a.mode == SGD_MODE_CONV3) {
        if (a.n <= 0 || a.hi <= 0 || a.wi <= 0 || (a.stride != 1 && a.stride != 2)) return SGD_ERR_ARG;
        if (a.stride == 2 && a.resample != SGD_RS_NONE) return SGD_ERR_ARG;
        const bool up = a.resample == SGD_RS_UP2 || a.resample == SGD_RS_ZEROUP2;
        g.hc = a.resample == SGD_RS_AVGPOOL2 ? a.hi / 2 : (up ? a.hi * 2 : a.hi);
        g.wc = a.resample == SGD_RS_AVGPOOL2 ? a.wi / 2 : (up ? a.wi * 2 : a.wi);
        if (a.resample == SGD_RS_AVGPOOL2 && ((a.hi | a.wi) & 1)) return SGD_ERR_ARG;
        const int ho = a.stride == 2 ? (g.hc + 1) / 2 : g.hc, wo = a.stride == 2 ? (g.wc + 1) / 2 : g.wc;
        if (a.ho != ho || a.wo != wo) return SGD_ERR_ARG;
        if (!is_pow2(a.ho) || !is_pow2(a.wo) || a.ho < 2 || a.wo < 2) return SGD_ERR_ARG;
        if (a.res && a.res_mode == SGD_RS_UP2 && ((a.ho | a.wo) & 1)) return SGD_ERR_ARG;
        if ((long)a.n * a.hi * a.wi >= (1L << 31)) return SGD_ERR_ARG;      // source rows are 32-bit in the tile table
        int tw = a.wo < 16 ? a.wo : 16;
        int th = BM / tw; if (th > a.ho) th = a.ho;
        while (th * tw > BM) th >>= 1;
        if (a.stride == 2) { if (tw > 8) tw = 8; if (th > 8) th = 8; }   // big input halos: smaller spatial tile
        int nb = BM / (th * tw);
        g.hh = a.stride == 2 ? 2 * th + 1 : th + 2;
        g.hw = a.stride == 2 ? 2 * tw + 1 : tw + 2;
        // the double-buffered halo tile must fit LDS; rows of images beyond nb are computed on don't-care data and
        // masked in the epilogue
        while (nb > 1 && 3 * (size_t)nb * g.hh * g.hw * LDA * 4 + (size_t)nb * g.hh * g.hw * 32 > 150 * 1024)
            nb >>= 1;
        g.tw_l2 = ilog2(tw); g.th_l2 = ilog2(th); g.nb = nb;
        g.tiles_x = a.wo / tw; g.tiles_y = a.ho / th;
        g.pix = nb * g.hh * g.hw;
        g.mt = ((a.n + nb - 1) / nb) * g.tiles_x * g.tiles_y;
        g.fast_a = g.pix <= FAST_PIX ? 1 : 0;
        na = 3;
    } else if (a.mode == SGD_MODE_FLAT) {
        if (a.m <= 0) return SGD_ERR_ARG;
        if (a.pro == SGD_PRO_AFFINE_NC && a.rows_per_n <= 0) return SGD_ERR_ARG;
        if (a.res && a.res_mode != SGD_RS_NONE) return SGD_ERR_ARG;
        g.tw_l2 = g.th_l2 = 0; g.nb = 1; g.tiles_x = g.tiles_y = 1; g.hh = g.hw = 1; g.hc = g.wc = 1;
        g.pix = BM * fg;          // fg 32-channel planes of the 128 rows side by side in one ring slot
        g.mt = (a.m + BM - 1) / BM;
        g.fast_a = 1;
        na = 3;
    } else {
        return SGD_ERR_ARG;
    }
    // epilogue statistics (args.stats): whole 128-row tiles of ONE image, 16-byte stores
    g.sparts = 0;
    const int ppt = bn >= 128 ? 1 : 4;                        // M slices per tile = compute-wave rows
    if (((a.cout | a.y_ld) & 3) == 0 && a.orows_in == 0) {
        if (a.mode == SGD_MODE_CONV3) {
            if (g.nb == 1 && (1 << (g.tw_l2 + g.th_l2)) == BM) g.sparts = g.tiles_x * g.tiles_y * ppt;
        } else if (a.rows_per_n > 0 && a.rows_per_n % BM == 0 && a.m % a.rows_per_n == 0) {
            g.sparts = a.rows_per_n / BM * ppt;
        }
    }
    return SGD_OK;
}

// args.tune: SGD_TUNE_BN128 never, SGD_TUNE_BN256 whenever the shape allows; default: the rule below
static bool want_bn256(const sgd_igemm_args& a) {
    if (a.tune & SGD_TUNE_BN128) return false;
    if (a.tune & SGD_TUNE_BN256) return true;
    const long rows = a.mode == SGD_MODE_CONV3 ? (long)a.n * a.ho * a.wo : (long)a.m;
    const long mt = (rows + BM - 1) / BM;
    const long t256 = mt * (a.cout_p / 256), t128 = mt * (a.cout_p / 128);
    // 3x3 launches in a split mode: the 128-column tile runs the 16x16x32 MFMA form, which the 64-column wave tile has no
    // registers for, and wins wherever it fills the chip (measured, round 3: +4..14 % on every layer of more than one round).
    // Up to one round of 128-column tiles the bigger tile was the rule since round 3 (8x8 maps at UNet batch 80: 160 tiles,
    // +6..21 %); round 6 re-measured the zone (profiles/r6_ab_small_launches.txt): the bigger tile wins only where the
    // balanced tail can cut it along K and two images share a tile -- 8x8 maps of >= 512 input channels (+7..21 %) -- and loses
    // 10..37 % elsewhere (16 x 256 -> 256 @32^2: 0.075 vs 0.055 ms; 32 x 512 -> 512 @16^2: 0.114 vs 0.092), as does "one round of
    // bigger tiles instead of a second, partly empty one" (80 x 1024 -> 1024 @8^2: 0.321 vs 0.277; 40 x 512 -> 512 @16^2: 0.141 vs 0.135)
    if (a.mode == SGD_MODE_CONV3 && a.prec != SGD_PREC_F32) return t128 <= 256 && a.ho * a.wo <= 64 && a.c0 + a.c1 >= 512;
    // whole rounds of 256 persistent blocks: a 128 x 256 tile costs 2 / 1.07 of a 128 x 128 one (measured, tools/ab_conv.py:
    // +5..9 % where both shapes fill the chip evenly), so it wins unless the coarser tiles quantise worse
    if (t256 <= 256 && t128 > 256) return true;    // one round of bigger tiles instead of a second, partly empty one
    const long r256 = (t256 + 255) / 256, r128 = (t128 + 255) / 256;
    return r256 * 2.0 < r128 * 1.07;
}

// Test hook (tests/test_boundary_cpu.py, no GPU): the balanced-tail workspace layout of a launch of `total_tiles` tiles with
// `nchunks` channel chunks per tile on `grid` blocks -- per block {split, counter index, first slab, slabs} (split 0: the
// block has no split tile) -- from the same functions the kernel uses.
extern "C" int sgd_igemm_tail_layout(int32_t total_tiles, int32_t nchunks, int32_t taps, int32_t grid, int32_t* out) {
    if (!out || grid < 8 || (grid & 7) || total_tiles < 0) return SGD_ERR_ARG;
    const int xchunk = (total_tiles + 7) >> 3, nloc = grid >> 3;
    for (int b = 0; b < grid; ++b) {
        const int xcd = b & 7, loc = b >> 3;
        const int xbeg = xcd * xchunk, xend = (xbeg + xchunk < total_tiles) ? xbeg + xchunk : total_tiles;
        const int xtiles = xend > xbeg ? xend - xbeg : 0;
        const int nfull = xtiles / nloc, xrem = xtiles - nfull * nloc;
        const int split = tail_split(xrem, nloc, nchunks, taps);
        int32_t* o = out + 4 * b;
        o[0] = o[1] = o[2] = o[3] = 0;
        if (split && loc < xrem * split) {
            o[0] = split;
            o[1] = tail_counter(xcd, loc, nloc, split);
            o[2] = tail_slab(xcd, loc, nloc, split);
            o[3] = split - 1;
        }
    }
    return SGD_OK;
}

extern "C" int64_t sgd_igemm_work_bytes(void) {
    // counters + the most slabs one launch can need: 8 XCDs x floor(32 / split) split tiles x (split - 1) producers, at the
    // 128 x 256 tile (128 KiB of partial accumulators per slab): split = 4 -> 192 slabs
    return (int64_t)WORK_HEAD + 192 * (int64_t)(BM * 256 * 4);
}

extern "C" int64_t sgd_igemm_work_status_offset(void) { return (int64_t)WORK_STATUS_INT * 4; }

extern "C" int sgd_igemm_stats_parts(const sgd_igemm_args* args) {
    if (!args) return 0;
    Geo g;
    int na;
    if (make_geo(*args, g, column_tile(*args), na) != SGD_OK) return 0;
    return g.sparts;
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
    int bn = column_tile(a);
    if (a.cout_p % pick_bn(a.cout) != 0 || a.cout_p < a.cout || a.cin_p % KC != 0 || a.cin_p < cin) return SGD_ERR_ARG;
    const bool vec = (a.c0 % 4 == 0) && (a.c1 % 4 == 0);
    // 128 x 256 tile (64 columns per compute wave): launches whose output channels allow it.  The packed-weight layout does
    // not depend on the tile (units of 32 output channels), so this is a launch-time choice.
    if (bn == 128 && a.cout_p % 256 == 0 && vec && (!a.res || a.res_mode == SGD_RS_NONE) && want_bn256(a)) bn = 256;
    int na;
    const bool conv = a.mode == SGD_MODE_CONV3;
    // Two planes per chunk (igemm_kernel<.., TAPS = 2>): OPT-IN, args.tune & SGD_TUNE_FLAT2.  Flat launches the lean loaders serve
    // (16-byte rows, no / per-image GroupNorm prologue, no dropout, whole 32-channel planes per source) with an even
    // number of planes.  Measured (round 4, tools/ab_conv.py, UNet batch 80): bit-identical to the one-plane instance,
    // +3..6 % on proj_out / decoder skips with a prologue, 0..3 % on plain ones, 0 on qkv and the HBM-bound 64x64 skips
    // -- the barrier per K step was never these launches' cost.  With the LayerNorm-row prologue (Attention_LR's to_q /
    // to_kv) in a split mode the same instance returns wrong rows -- always tile rows 6, 7 mod 8, i.e. lanes 48..63 of a
    // loader wave, a different subset on every launch -- while exact f32 and every other prologue stay bit-identical.
    // DESIGN.md section 4 (round 4) and profiles/r4_ln_hazard.txt hold what tools/ln_hazard.py established: the wrong
    // cells hold exactly beta in the low lane of a packed-f32 pair (the LayerNorm value with a zero product), in three of
    // the six unrolled copies of the staging code only; no wait or idle cycle around the loads or the LDS stores changes
    // it, moving the surrounding code does.  Round 6: with packed-f32 code generation off the same two-plane instance passes
    // 240 of 240 launches that fail 238 of 240 with it on (tools/ln_hazard.py, profiles/r6_ln_hazard.txt), so every
    // LayerNorm launch of a split mode runs on the no-packed-f32 unit (ln_nopk below); the two-plane instance stays opt-in.
    int taps = conv ? 9 : 1;
    // LayerNorm-row prologue in a split mode: the unit without packed-f32 instructions (SGD_TUNE_LN_PACKED: the regular unit --
    // tools/ln_hazard.py reproduces the round-4 fault with it); only there may the two-plane instance serve that prologue
    const bool ln_nopk = !conv && a.pro == SGD_PRO_LN_ROW && a.prec != SGD_PREC_F32 && !(a.tune & SGD_TUNE_LN_PACKED);
    const bool ln_any = !conv && a.pro == SGD_PRO_LN_ROW && (ln_nopk || (a.tune & SGD_TUNE_LN_PACKED));
    if (!conv && vec && bn >= 128 && a.drop_p == 0.f && cin % (2 * KC) == 0 && (a.c1 == 0 || a.c0 % KC == 0)
        && (a.pro == SGD_PRO_NONE || (a.pro == SGD_PRO_AFFINE_NC && a.rows_per_n > 0 && a.rows_per_n % BM == 0) || ln_any)
        && (a.tune & SGD_TUNE_FLAT2))
        taps = 2;
    {
        const int rc = make_geo(a, g, bn, na, conv ? 1 : taps);
        if (rc != SGD_OK) return rc;
    }
    if (a.stats && g.sparts == 0) return SGD_ERR_ARG;
    g.nt = a.cout_p / bn;
    if ((a.work && a.work_bytes < sgd_igemm_work_bytes()) || (a.tune & SGD_TUNE_PLAIN_SCHEDULE)) a.work = nullptr;
    {
        // epilogue uses 32-bit row indices
        const long rows_out = a.mode == SGD_MODE_CONV3 ? (long)a.n * a.ho * a.wo : (long)a.m;
        const long rows_res = a.res_mode == SGD_RS_AVGPOOL2 ? rows_out * 4 : rows_out;
        if (rows_out >= (1L << 31) || rows_res >= (1L << 31)) return SGD_ERR_ARG;
    }
#ifdef SGDM_PROBE
    {
        const char* e = getenv("SGDM_DBG");
        g.dbg = e ? atoi(e) : 0;
        const char* sp = getenv("SGDM_STAMP_PTR");       // device buffer of gridDim * 12 * 4 u64 (tools/probe_conv.py)
        g.stamp = sp ? reinterpret_cast<unsigned long long*>(strtoull(sp, nullptr, 0)) : nullptr;
        const char* tp = getenv("SGDM_TRACE_PTR");
        g.trace = tp ? reinterpret_cast<unsigned long long*>(strtoull(tp, nullptr, 0)) : nullptr;
    }
#endif
    const size_t smem = (size_t)na * g.pix * LDA * sizeof(float) + (size_t)g.pix * 32
                        + (a.cout_p <= BIAS_LDS_MAX ? (size_t)a.cout_p * sizeof(float) : 0);
    if (smem > 160 * 1024) return SGD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // Loader-side epilogue (igemm_kernel<.., DEFER>): 3x3 launches with 128-column tiles in a split mode, 16-byte inputs
    // and outputs, no or same-row residual, bias in LDS, at least 3 chunks per tile (the slices of a tile's epilogue ride
    // on the periods of the next one), buffers addressable with 32-bit byte offsets, and room for the staging tile
    int variant = vec ? 1 : 0;
    size_t smem_launch = smem;
    {
        // Opt-in (args.tune & SGD_TUNE_DEFER).  Measured (round 3, UNet batch 80): correct, the compute waves' epilogue time drops from
        // 8..10 us per tile to 0.3 us -- and the launches are 3..7 % SLOWER: the chip is power-limited, the idle wait cost
        // little energy, and the staging copy plus the loaders' extra instructions cost more than the wait saved.
        const int nchunks = (cin + KC - 1) / KC;
        const long rows_out = (long)a.n * a.ho * a.wo;
        const size_t smem_defer = smem + (size_t)BM * (128 + 4) * sizeof(float);
        if (conv && vec && bn == 128 && a.prec != SGD_PREC_F32 && ((a.cout | a.y_ld) & 3) == 0
            && (!a.res || a.res_mode == SGD_RS_NONE) && a.resample != SGD_RS_AVGPOOL2 && a.cout_p <= BIAS_LDS_MAX && nchunks >= 3
            && rows_out * a.y_ld * 4 < (1L << 32) && (!a.stats || (long)a.n * g.sparts * 2 * a.cout * 4 < (1L << 32))
            && smem_defer <= 160 * 1024 && (a.tune & SGD_TUNE_DEFER)) {
            variant = 2;
            smem_launch = smem_defer;
        }
    }
    switch (a.prec) {
        case SGD_PREC_F32: return sgd_igemm_dispatch_f32(&ka, bn, variant, taps, smem_launch, st);
        case SGD_PREC_F16X3: return ln_nopk ? sgd_igemm_dispatch_f16x3_nopk(&ka, bn, variant, taps, smem_launch, st)
                                            : sgd_igemm_dispatch_f16x3(&ka, bn, variant, taps, smem_launch, st);
        case SGD_PREC_BF16X3: return ln_nopk ? sgd_igemm_dispatch_bf16x3_nopk(&ka, bn, variant, taps, smem_launch, st)
                                             : sgd_igemm_dispatch_bf16x3(&ka, bn, variant, taps, smem_launch, st);
        default: return SGD_ERR_ARG;
    }
}
