// Shared by the translation units of the fused implicit-GEMM conv / linear kernel: igemm.hip (the kernel, one unit per arithmetic
// mode), igemm_host.hip (geometry, launch rules, the extern "C" entry points) and pack.hip (weight packing).  Tile constants, the
// balanced tail's schedule arithmetic (used by the kernel AND by the host-side layout hook the CPU tests check), the launch geometry
// and the argument block that crosses between the units as bytes.
#pragma once
#include <stdlib.h>

#include "sgdm_common.h"
#include "../../include/sgdm_hip.h"

namespace {

constexpr int KC = 32;        // input channels per K chunk
constexpr int LDA = KC + 4;   // LDS row stride in floats (144 B): conflict-free b128 fragment reads
constexpr int BM = 128;
constexpr int NB_RING = 3;    // register ring depth of the 1x1 loaders (input rows requested 3 steps ahead)
constexpr int BIAS_LDS_MAX = 4096;   // layers up to this many (padded) output channels keep their bias in LDS (16 KB)
constexpr int A_THREADS = 256;  // 4 input-tile loader waves
constexpr int NCOMP = 256;      // 4 compute (MFMA) waves, one per SIMD
constexpr int NTHREADS = NCOMP + A_THREADS;
constexpr int WUNIT = 4096;   // bytes of one packed weight unit: 32 output x 32 input channels of one tap, fragment order
constexpr int MIN_PART_STEPS = 27;   // K steps (tap x 32 channels) of the smallest K part worth splitting off
constexpr int SPLIT_MAX = 4;   // K parts of a tile of the balanced tail (sgd_igemm_args.work)
constexpr int WORK_TILES = 256;                        // split tiles of one launch: < blocks
constexpr int WORK_HEAD = WORK_TILES * 8;              // bytes: per split tile {arrived, consumed} wave counters
// Health word of the workspace (sgd_igemm_work_status_offset): the last int of the head.  Counter pairs use indices
// < 8 * (nloc / 2) <= 128 of the 256 pairs, so the word is never a counter.
constexpr int WORK_STATUS_INT = WORK_HEAD / 4 - 1;
// Finisher poll bound: s_sleep 16 = 1024 cycles + one L2 round trip per poll, 2^20 polls ~ 1-2 s (measured 2.1 s at 2^22 sleeps
// only; a legitimate wait is a producer's K part: a few hundred microseconds).  Producers never wait and the launches that share a
// workspace are ordered on one stream, so a finisher that is still waiting then is waiting for a block that will never
// store (stale counters after a faulted launch, a second stream on the same workspace): it flags the workspace and
// poisons its outputs with NaN instead of hanging the device.
constexpr int FINISH_POLL_MAX = 1 << 20;

// Balanced-tail arithmetic shared by the kernel and sgd_igemm_tail_layout (the CPU test of the workspace layout).
// K parts of the `xrem` tiles an XCD has left after its whole rounds (0: no split): a part must be worth its hand-off
// (publish + poll + acquire + the slab reads of the finisher, ~10 us): at least MIN_PART_STEPS K steps.  Measured
// (tools/ab_conv.py, UNet batch 80): 3x3 convs of >= 256 input channels gain 4..7 %, 128-channel ones (2 chunks per
// part) and every 1x1 launch lose 5..15 %.
__host__ __device__ inline int tail_split(int xrem, int nloc, int nchunks, int taps) {
    if (xrem <= 0 || nchunks < 2) return 0;
    int split = nloc / xrem;
    if (split > SPLIT_MAX) split = SPLIT_MAX;
    while (split >= 2 && (nchunks / split) * taps < MIN_PART_STEPS) --split;
    return split < 2 ? 0 : split;
}
// Every XCD owns a FIXED range of counters and slabs: its split depends on ITS remainder (the last XCD of a launch usually
// has fewer tiles), and ranges sized by the XCD's own split overlapped between XCDs of different splits -- two split tiles
// on one counter: sums of the wrong tile, then a finisher polling forever (round 3: 1 evaluation in ~20 of the ch=224
// model at batch 1).  An XCD has at most nloc / 2 split tiles and (nloc / split) * (split - 1) <= 3 nloc / 4 producer
// slabs: 8 * 24 = 192 at 256 blocks (sgd_igemm_work_bytes).
__host__ __device__ inline int tail_counter(int xcd, int loc, int nloc, int split) { return xcd * (nloc >> 1) + loc / split; }
__host__ __device__ inline int tail_slab(int xcd, int loc, int nloc, int split) {
    return xcd * ((nloc * 3) >> 2) + (loc / split) * (split - 1);
}
constexpr int FAST_PIX = 192;  // halo tiles up to this many pixels (16x8 outputs + halo = 180) use the split-phase A loader

struct Geo {
    int tw_l2, th_l2;         // log2 of the spatial tile (CONV3)
    int hh, hw;               // halo tile dims (rows, cols) in conv-input pixels
    int nb;                   // images per M tile
    int tiles_x, tiles_y;     // spatial tiles per image
    int pix;                  // nb*hh*hw (CONV3) or 128 (FLAT)
    int mt, nt;               // number of M / N tiles
    int hc, wc;               // conv-input dims (after resample)
    int sparts;               // statistics slots per image (args.stats), 0: unsupported geometry
    int fast_a;               // 1: split-phase A loader (stride 1, no avg-pool, pix*8 <= JMAX*256)
#ifdef SGDM_PROBE
    int dbg;                  // SGDM_DBG bits: 1 skip A staging, 2 skip B staging, 4 skip MFMA, 8 skip stores, 16 skip the epilogue, 256 skip its statistics
    unsigned long long* stamp;   // [block][wave][4]: total cycles, cycles inside barriers, barriers, epilogue cycles
    unsigned long long* trace;   // [16 blocks][wave][512 barriers][2]: arrival / release time of every barrier
#endif
};

struct KArgs {
    sgd_igemm_args a;
    Geo g;
};

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
// compute units of the current device in whole groups of 8, at most 256 (the balanced tail's workspace layout is sized for
// 32 blocks per XCD); read once per process and translation unit
inline int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        n &= ~7;
        cus = n < 8 ? 8 : (n > 256 ? 256 : n);
    }
    return cus;
}
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

// One translation unit per arithmetic mode (build.py compiles igemm.hip with -DSGDM_IGEMM_PREC=0 / 1 / 2, in parallel: the kernel
// template has 12 instances per mode and eight epilogue variants each): the mode's launch dispatcher has external linkage, the
// geometry and the entry points live in igemm_host.hip.  The argument block crosses as bytes.
int sgd_igemm_dispatch_f32(const void* ka, int bn, int vec, int taps, size_t smem, hipStream_t st);
int sgd_igemm_dispatch_f16x3(const void* ka, int bn, int vec, int taps, size_t smem, hipStream_t st);
int sgd_igemm_dispatch_bf16x3(const void* ka, int bn, int vec, int taps, size_t smem, hipStream_t st);
// The LayerNorm-row prologue (Attention_LR's to_q / to_kv, crossattetion_lr.py:81-88) in a split mode runs on instances compiled
// with packed-f32 code generation OFF (build.py: -DSGDM_IGEMM_NOPK -Xclang -target-feature -Xclang -packed-fp32-ops; 1x1 / linear
// instances only).  Round 4 found the two-plane instance of that prologue returning exactly beta -- the LayerNorm value with a zero
// product -- in the low lane of v_pk_{mul,fma}_f32 for lanes 48..63 of a loader wave, in specific unrolled copies, on every launch;
// with packed-f32 code generation off the same source passes (profiles/r5_ln_hazard_isa.txt).  The one-plane
// instance the product uses has never shown it (canary + bit-repeatability tests), but it issues the same instructions: since round 6
// no LayerNorm launch executes a packed-f32 instruction at all.  The whole library built that way cost 7.6 % of a sampling step (76
// spilled registers in the 3x3 instance); confined to these launches it costs C2 nothing and C4 / C5 the difference on ~20 launches.
int sgd_igemm_dispatch_f16x3_nopk(const void* ka, int bn, int vec, int taps, size_t smem, hipStream_t st);
int sgd_igemm_dispatch_bf16x3_nopk(const void* ka, int bn, int vec, int taps, size_t smem, hipStream_t st);
