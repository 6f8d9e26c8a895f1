// In-run calibration of the device (bench.py: roofline.device_mfma_tflops / device_copy_tbps).  gfx950 only.
//
// The boxes of the pool differ by ~5-7 % for one binary and the chip trades clock for matrix-pipe duty on random data
// (DESIGN.md section 4), so a bench line alone cannot tell a slower box from slower code.  These two kernels measure, in
// the same process and thermal state as the timed steps, what THIS device delivers at that moment:
//   * sgd_debug_mfma_probe: nothing but v_mfma_f32_16x16x32_f16 on random register operands, one wave per SIMD on every
//     compute unit, eight independent accumulator tiles per wave (the pipe never waits on a dependency), no memory
//     traffic inside the loop -- the matrix pipe's sustained rate under its own power draw;
//   * sgd_debug_copy_probe: a 16-byte-per-lane grid-stride copy (four loads in flight per lane) -- the practical HBM rate
//     for one read + one write stream.
// Diagnostics (libsgdm_hip_tools.so, include/sgdm_hip_tools.h): never on the product path, not part of libsgdm_hip.so.
#include "../sgdm_common.h"
#include "../../../include/sgdm_hip_tools.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// a random f16 in [-1, 1) with a full random mantissa (the data the conv kernels see is dense in toggling bits)
__device__ __forceinline__ _Float16 rnd_f16(uint32_t h) {
    const float f = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    return (_Float16)f;
}

// VARIANT 0: 16x16x32 on random operands (the figure bench.py reports); 1: the same on all-zero operands (the issue rate at
// the clock an idle-data loop holds: separates "issue-limited" from "power-limited"); 2: 32x32x16 on random operands
template <int VARIANT>
__global__ __launch_bounds__(256) void mfma_probe_kernel(uint32_t seed, long iters, float* __restrict__ out) {
    extern __shared__ float hold[];                  // 150 KB of dynamic LDS: one block per compute unit, nothing beside it
    (void)hold;
    const uint32_t id = (blockIdx.x * 256u + threadIdx.x) * 64u + seed;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[i][j] = VARIANT == 1 ? (_Float16)0.f : rnd_f16(mix32(id + (uint32_t)(i * 8 + j)));
            b[i][j] = VARIANT == 1 ? (_Float16)0.f : rnd_f16(mix32(id + 32u + (uint32_t)(i * 8 + j)));
        }
    float total = 0.f;
    if constexpr (VARIANT == 2) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (long it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)       // (inline asm: in-place accumulation, see below)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i]), "v"(b[(i + 1) & 3]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) total += acc[i][r];
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (long it = 0; it < iters; ++it) {
#pragma unroll
            // inline asm with the accumulator as an in-place AGPR operand: through the builtin the register allocator rotated
            // the eight tiles through one another every iteration (~40 v_accvgpr moves per 8 MFMAs: the loop measured the
            // moves, 883 TF on a device whose conv kernel sustains 1,170)
            for (int i = 0; i < 8; ++i)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 3]), "v"(b[(i + (i >> 2)) & 3]));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    if (out) out[(size_t)blockIdx.x * 256 + threadIdx.x] = total;
}

// VARIANT 0: grid-stride, four 16-byte loads in flight per lane (the round-5 probe: 4.6-4.8 TB/s of read + written bytes);
// 1: the same with eight in flight; 2: four in flight, non-temporal loads and stores; 3: every wave walks CONTIGUOUS 8 KiB pieces
// (eight consecutive 1 KiB wave-instructions), pieces dealt round-robin over the waves of the grid; 4: as 3 with non-temporal
// accesses; 5: read only (the sum lands in dst[wave]); 6: write only (fill).  tools/hbm_probe_sweep.py holds the sweep.
template <int VARIANT>
__global__ __launch_bounds__(256) void copy_probe_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if constexpr (VARIANT <= 2) {
        constexpr int U = VARIANT == 1 ? 8 : 4;
        for (; i + (U - 1) * stride < n; i += U * stride) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = VARIANT == 2 ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (VARIANT == 2) __builtin_nontemporal_store(v[u], dst + i + u * stride);
                else dst[i + u * stride] = v[u];
            }
        }
        for (; i < n; i += stride) dst[i] = src[i];
    } else {
        constexpr int U = 8;                                    // 8 KiB per wave and piece
        const long wave = i >> 6, nwaves = stride >> 6;
        const int lane = threadIdx.x & 63;
        const long pieces = n / (64 * U);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (long p = wave; p < pieces; p += nwaves) {
            const long b = p * (64 * U) + lane;
            f32x4 v[U];
            if constexpr (VARIANT != 6) {
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = VARIANT == 4 ? __builtin_nontemporal_load(src + b + u * 64) : src[b + u * 64];
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = f32x4{(float)p, (float)u, 0.f, 1.f};
            }
            if constexpr (VARIANT == 5) {
#pragma unroll
                for (int u = 0; u < U; ++u) acc += v[u];
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if constexpr (VARIANT == 4) __builtin_nontemporal_store(v[u], dst + b + u * 64);
                    else dst[b + u * 64] = v[u];
                }
            }
        }
        if constexpr (VARIANT == 5) dst[i] = acc;
        else for (long t = pieces * (64 * U) + i; t < n; t += stride) dst[t] = VARIANT == 6 ? f32x4{0.f, 0.f, 0.f, 1.f} : src[t];
    }
}

// What does the conv kernel's compute-wave stream cost the matrix pipe, and would a second MFMA wave per SIMD give it back?
// One "K step" of a wave = MT row blocks x NT column blocks of 16 x 16 x 32 products in three split-precision MFMAs each (MT * NT =
// 16: 48 MFMAs, 64 accumulator registers), with the row-block operands (hi + lo: two 16-byte LDS reads per row block, 2 * MT per
// step) re-read from LDS for every step right after their last use, as igemm_kernel does; the column-block operands stay in
// registers (the conv kernel streams them from L2).  MT = 8, NT = 2 is the shipped 128 x 32 wave tile (16 reads per 48 MFMAs),
// MT = 4, NT = 4 the 64 x 64 one (8 reads), MT = 0 no reads at all.  WPS = MFMA waves per SIMD (block = 256 * WPS threads).
template <int MT, int NT, int WPS>
__global__ __launch_bounds__(256 * WPS) void mfma_lds_probe_kernel(uint32_t seed, long iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) _Float16 plds[];          // 150 KB: one block per CU; the first 64 KB are read
    constexpr int RMT = MT > 0 ? MT : 8;                 // row blocks (MT == 0: operands never reloaded)
    constexpr int RNT = 16 / RMT;
    static_assert(MT == 0 || MT * NT == 16, "48 MFMAs per step");
    for (int i = threadIdx.x; i < 32768; i += 256 * WPS) plds[i] = rnd_f16(mix32(seed + 17u * (uint32_t)i + blockIdx.x));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t id = (blockIdx.x * 256u * WPS + threadIdx.x) * 64u + seed;
    f16x8 bh[RNT], bl[RNT], ah[RMT], al[RMT];
#pragma unroll
    for (int i = 0; i < RNT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { bh[i][j] = rnd_f16(mix32(id + (uint32_t)(i * 8 + j))); bl[i][j] = rnd_f16(mix32(id + 64u + (uint32_t)(i * 8 + j))); }
    // lane l reads 16 consecutive bytes at 16 l: every ds_read_b128 is conflict-free; (step & 3, row block, plane) select the 1 KB slice
    const _Float16* base = plds + lane * 8;
    auto slice = [&](long step, int mt, int plane) { return base + ((int)(step & 3) * 16 + mt * 2 + plane) * 512; };
#pragma unroll
    for (int mt = 0; mt < RMT; ++mt) { ah[mt] = *reinterpret_cast<const f16x8*>(slice(0, mt, 0)); al[mt] = *reinterpret_cast<const f16x8*>(slice(0, mt, 1)); }
    f32x4 acc[RMT][RNT];
#pragma unroll
    for (int i = 0; i < RMT; ++i)
#pragma unroll
        for (int j = 0; j < RNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int nt = 0; nt < RNT; ++nt)
#pragma unroll
            for (int mt = 0; mt < RMT; ++mt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                if (MT > 0 && nt == RNT - 1) {           // last use of row block mt in this step: its operands of the next step
                    ah[mt] = *reinterpret_cast<const f16x8*>(slice(it + 1, mt, 0));
                    al[mt] = *reinterpret_cast<const f16x8*>(slice(it + 1, mt, 1));
                }
            }
    }
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < RMT; ++i)
#pragma unroll
        for (int j = 0; j < RNT; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (out) out[(size_t)blockIdx.x * 256 * WPS + threadIdx.x] = total;
}

// The same stream with the rest of the conv kernel's per-step traffic added one piece at a time (EX bits), on the shipped wave tile
// (8 row blocks x 2 column blocks, one MFMA wave per SIMD):
//   1  the column-block (weight) fragments are re-loaded from global memory every step -- 4 x 16 bytes per lane from a 1 MB buffer
//      that stays in L2, requested right after the last use of the column block, as igemm_kernel does;
//   2  four LOADER waves share the SIMDs: per 9 steps ("a chunk") each loader thread moves six 16-byte rows from a streaming global
//      buffer through a GroupNorm-affine + SiLU + hi / lo split (the lean conv loader's arithmetic) into LDS (2 x 8 bytes);
//   4  compute and loader waves meet at one s_barrier per chunk;
//   8  (instead of 4) producer / consumer counters in LDS: a wave waits only when the other role is really behind.
// WIDE: the 8-wave variant -- eight MFMA waves (two per SIMD) of 64 x 64 (4 row blocks x 4 column blocks: 8 LDS reads and 8 weight
// loads per step and wave; pairs of waves load the same weight fragments) next to the same four loader waves, whose chunk now feeds
// twice the MFMAs
template <int EX, bool WIDE>
__global__ __launch_bounds__((WIDE ? 512 : 256) + ((EX & 2) ? 256 : 0)) void mfma_stream_probe_kernel(uint32_t seed, long iters, const f16x8* __restrict__ wbuf,
                                                                              const f32x4* __restrict__ abuf, long arows,
                                                                              float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) _Float16 plds[];
    constexpr int CTHR = WIDE ? 512 : 256, NTHR = CTHR + ((EX & 2) ? 256 : 0);
    constexpr int CW = CTHR / 64, RMT = WIDE ? 4 : 8, RNT = WIDE ? 4 : 2;
    __shared__ int flags[2];                             // (EX & 8) ready: chunks staged x 4 loader waves; done: chunks consumed x 4
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 32768; i += NTHR) plds[i] = rnd_f16(mix32(seed + 17u * (uint32_t)i + blockIdx.x));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long chunks = iters / 9;
    if (wave >= CW) {
        // ---- loader role
        if constexpr ((EX & 2) != 0) {
            const int lt = threadIdx.x - CTHR;
            _Float16* dst = plds + 40960 + lt * 8;                       // beyond the slices the compute waves read
            const f32x4 ka = {1.01f, 0.99f, 1.02f, 0.98f}, kb = {0.01f, -0.02f, 0.03f, -0.01f};
            f32x4 raw[6];
            long row = ((long)blockIdx.x * 256 + lt) * 6 % arows;
#pragma unroll
            for (int j = 0; j < 6; ++j) raw[j] = abuf[(row + j) % arows];
            for (long c = 0; c < chunks; ++c) {
                row = (row + 6L * 256 * gridDim.x) % arows;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x4 v = raw[j] * ka + kb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[e]));
                    f16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = (_Float16)v[e]; l[e] = (_Float16)(v[e] - (float)h[e]); }
                    *reinterpret_cast<f16x4*>(dst + ((c & 1) * 6 + j) * 2048) = h;
                    *reinterpret_cast<f16x4*>(dst + ((c & 1) * 6 + j) * 2048 + 4) = l;
                    raw[j] = abuf[(row + j) % arows];
                }
                if constexpr ((EX & 4) != 0) __syncthreads();
                if constexpr ((EX & 8) != 0) {
                    // producer / consumer counters in LDS instead of the rendezvous: this wave's stores of chunk c are done ->
                    // ready += 1; before it overwrites the slot two chunks later the compute waves must be past chunk c - 1
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add(&flags[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (c >= 1) {
                        const int need = CW * (int)c;                            // every compute wave finished chunk c - 1
                        while (__hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
        }
        return;
    }
    const uint32_t id = (blockIdx.x * (uint32_t)CTHR + threadIdx.x) * 64u + seed;
    f16x8 bh[RNT], bl[RNT], ah[RMT], al[RMT];
    // [step & 63][column group][nt][plane][lane]: 4 groups of 2 blocks (one per wave) or, WIDE, 4 groups of 4 (shared by two waves)
    const f16x8* wl = wbuf + (size_t)(WIDE ? wave & 3 : wave) * (RNT * 2) * 64 + lane;
    auto wslice = [&](long step, int nt, int plane) { return wl + ((size_t)(step & 63) * (4 * RNT * 2) + nt * 2 + plane) * 64; };
#pragma unroll
    for (int i = 0; i < RNT; ++i) {
        if constexpr ((EX & 1) != 0) { bh[i] = *wslice(0, i, 0); bl[i] = *wslice(0, i, 1); }
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { bh[i][j] = rnd_f16(mix32(id + (uint32_t)(i * 8 + j))); bl[i][j] = rnd_f16(mix32(id + 64u + (uint32_t)(i * 8 + j))); }
        }
    }
    const _Float16* base = plds + lane * 8;
    auto slice = [&](long step, int mt, int plane) { return base + ((int)(step & 3) * 16 + mt * 2 + plane) * 512; };
#pragma unroll
    for (int mt = 0; mt < RMT; ++mt) { ah[mt] = *reinterpret_cast<const f16x8*>(slice(0, mt, 0)); al[mt] = *reinterpret_cast<const f16x8*>(slice(0, mt, 1)); }
    f32x4 acc[RMT][RNT];
#pragma unroll
    for (int i = 0; i < RMT; ++i)
#pragma unroll
        for (int j = 0; j < RNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    long it = 0;
    for (long c = 0; c < chunks; ++c) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++it) {
#pragma unroll
            for (int nt = 0; nt < RNT; ++nt)
#pragma unroll
                for (int mt = 0; mt < RMT; ++mt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    if (nt == RNT - 1) {
                        ah[mt] = *reinterpret_cast<const f16x8*>(slice(it + 1, mt, 0));
                        al[mt] = *reinterpret_cast<const f16x8*>(slice(it + 1, mt, 1));
                    }
                    if ((EX & 1) && mt == RMT - 1) { bh[nt] = *wslice(it + 1, nt, 0); bl[nt] = *wslice(it + 1, nt, 1); }
                }
        }
        if constexpr ((EX & 4) != 0 && (EX & 2) != 0) __syncthreads();
        if constexpr ((EX & 8) != 0) {
            // done with chunk c (its last fragment reads were issued for step it, which belong to the next chunk's slot in the real
            // kernel: the counter goes up once per chunk and wave); chunk c + 1 must be staged: four loader waves x (c + 1) chunks
            if (lane == 0) __hip_atomic_fetch_add(&flags[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int need = 4 * ((int)c + 1);                               // four loader waves staged chunk c + 1
            while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
        }
    }
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < RMT; ++i)
#pragma unroll
        for (int j = 0; j < RNT; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (out) out[(size_t)blockIdx.x * CTHR + threadIdx.x] = total;
}

}  // namespace

template <int EX, bool WIDE>
static int launch_stream_probe(int32_t blocks, int64_t iters, uint32_t seed, const void* wbuf, const void* abuf, int64_t arows,
                               float* out, void* stream) {
    const size_t LDS = 150 * 1024;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)mfma_stream_probe_kernel<EX, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        attr = true;
    }
    hipLaunchKernelGGL((mfma_stream_probe_kernel<EX, WIDE>), dim3(blocks), dim3((WIDE ? 512 : 256) + ((EX & 2) ? 256 : 0)), LDS,
                       (hipStream_t)stream, seed, (long)iters, reinterpret_cast<const f16x8*>(wbuf), reinterpret_cast<const f32x4*>(abuf),
                       (long)arows, out);
    return sgd_check_launch();
}

// extras: bit 0 weight fragments from global (wbuf: >= 2 MiB), bit 1 loader waves (abuf: arows rows of 16 bytes), bit 2 a barrier per
// 9 steps (only with bit 1), bit 3 LDS counters instead of it, bit 4 the 8-wave variant (two MFMA waves of 64 x 64 per SIMD).  iters
// is rounded down to a multiple of 9; flops = blocks * (4 or 8) * (iters / 9 * 9) * 48 * 16384.
extern "C" int sgd_debug_mfma_stream_probe(int32_t blocks, int64_t iters, uint32_t seed, int32_t extras, const void* wbuf,
                                           const void* abuf, int64_t arows, float* out, void* stream) {
    SGD_CLEAR_ERR();
    const int wide = extras & 16;
    extras &= 15;
    if (blocks <= 0 || blocks > 4096 || iters < 9 || (extras & 12) == 12 || ((extras & 1) && !wbuf) || ((extras & 2) && (!abuf || arows < 4096)))
        return SGD_ERR_ARG;
    if ((extras & 12) && !(extras & 2)) return SGD_ERR_ARG;
#define SGD_SP(E) case E: return wide ? launch_stream_probe<E, true>(blocks, iters, seed, wbuf, abuf, arows, out, stream) \
                                      : launch_stream_probe<E, false>(blocks, iters, seed, wbuf, abuf, arows, out, stream)
    switch (extras) {
        SGD_SP(0); SGD_SP(1); SGD_SP(2); SGD_SP(3); SGD_SP(6); SGD_SP(7); SGD_SP(10); SGD_SP(11);
        default: return SGD_ERR_ARG;
    }
#undef SGD_SP
}

template <int MT, int NT, int WPS>
static int launch_mfma_lds_probe(int32_t blocks, int64_t iters, uint32_t seed, float* out, void* stream) {
    const size_t LDS = 150 * 1024;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)mfma_lds_probe_kernel<MT, NT, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        attr = true;
    }
    hipLaunchKernelGGL((mfma_lds_probe_kernel<MT, NT, WPS>), dim3(blocks), dim3(256 * WPS), LDS, (hipStream_t)stream, seed, (long)iters, out);
    return sgd_check_launch();
}

// row_blocks: 8 (the shipped 128 x 32 wave tile), 4 (64 x 64) or 0 (no LDS reads); waves_per_simd: 1 or 2.  Flops of a launch:
// blocks * 4 * waves_per_simd * iters * 48 * 16384.
extern "C" int sgd_debug_mfma_lds_probe(int32_t blocks, int64_t iters, uint32_t seed, int32_t row_blocks, int32_t waves_per_simd,
                                        float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (blocks <= 0 || blocks > 4096 || iters <= 0 || (waves_per_simd != 1 && waves_per_simd != 2)) return SGD_ERR_ARG;
#define SGD_LP(MT, NT) return waves_per_simd == 1 ? launch_mfma_lds_probe<MT, NT, 1>(blocks, iters, seed, out, stream) \
                                                  : launch_mfma_lds_probe<MT, NT, 2>(blocks, iters, seed, out, stream)
    if (row_blocks == 8) { SGD_LP(8, 2); }
    if (row_blocks == 4) { SGD_LP(4, 4); }
    if (row_blocks == 0) { SGD_LP(0, 0); }
#undef SGD_LP
    return SGD_ERR_ARG;
}

template <int VARIANT>
static int launch_mfma_probe(int32_t blocks, int64_t iters, uint32_t seed, float* out, void* stream) {
    const size_t LDS = 150 * 1024;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)mfma_probe_kernel<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        attr = true;
    }
    hipLaunchKernelGGL((mfma_probe_kernel<VARIANT>), dim3(blocks), dim3(256), LDS, (hipStream_t)stream, seed, (long)iters, out);
    return sgd_check_launch();
}

extern "C" int sgd_debug_mfma_probe(int32_t blocks, int64_t iters, uint32_t seed, int32_t variant, float* out, void* stream) {
    SGD_CLEAR_ERR();
    if (blocks <= 0 || blocks > 4096 || iters <= 0 || variant < 0 || variant > 2) return SGD_ERR_ARG;
    if (variant == 1) return launch_mfma_probe<1>(blocks, iters, seed, out, stream);
    if (variant == 2) return launch_mfma_probe<2>(blocks, iters, seed, out, stream);
    return launch_mfma_probe<0>(blocks, iters, seed, out, stream);
}

extern "C" int64_t sgd_debug_mfma_probe_flops(int32_t blocks, int64_t iters, int32_t variant) {
    // blocks x 4 waves x iters x (8 MFMAs of 16 x 16 x 32, or 4 of 32 x 32 x 16) x 2 flop per multiply-add
    return (int64_t)blocks * 4 * iters * (variant == 2 ? 4 * 32768 : 8 * 16384);
}

extern "C" int sgd_debug_copy_probe(const float* src, float* dst, int64_t count, int32_t variant, int32_t blocks, void* stream) {
    SGD_CLEAR_ERR();
    if (!src || !dst || count <= 0 || (count & 3) || (((uintptr_t)src | (uintptr_t)dst) & 15) || variant < 0 || variant > 6) return SGD_ERR_ARG;
    if (blocks <= 0) blocks = 256 * 8;
    if (variant == 5 && (long)blocks * 256 * 4 > count) return SGD_ERR_ARG;          // one result quad per thread
    const f32x4* s = reinterpret_cast<const f32x4*>(src);
    f32x4* d = reinterpret_cast<f32x4*>(dst);
    const long n = (long)(count >> 2);
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 0: hipLaunchKernelGGL(copy_probe_kernel<0>, dim3(blocks), dim3(256), 0, st, s, d, n); break;
        case 1: hipLaunchKernelGGL(copy_probe_kernel<1>, dim3(blocks), dim3(256), 0, st, s, d, n); break;
        case 2: hipLaunchKernelGGL(copy_probe_kernel<2>, dim3(blocks), dim3(256), 0, st, s, d, n); break;
        case 3: hipLaunchKernelGGL(copy_probe_kernel<3>, dim3(blocks), dim3(256), 0, st, s, d, n); break;
        case 4: hipLaunchKernelGGL(copy_probe_kernel<4>, dim3(blocks), dim3(256), 0, st, s, d, n); break;
        case 5: hipLaunchKernelGGL(copy_probe_kernel<5>, dim3(blocks), dim3(256), 0, st, s, d, n); break;
        default: hipLaunchKernelGGL(copy_probe_kernel<6>, dim3(blocks), dim3(256), 0, st, s, d, n); break;
    }
    return sgd_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// Diagnostic: hold `blocks` compute units for a wall-clock interval (sgd_debug_occupy).  512 threads + 150 KB of LDS per
// block: nothing of the persistent conv kernel (one block per CU) fits beside it, which is what a collective's kernels on
// a side stream do to the backward's launches (tests/test_hip_contention.py).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(512) void occupy_kernel(unsigned long long ticks) {
    extern __shared__ float occ_lds[];
    const unsigned long long t0 = wall_clock64();                 // constant 100 MHz counter
    occ_lds[threadIdx.x] = (float)threadIdx.x;                    // touch the allocation so it is real
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (occ_lds[threadIdx.x] < 0.f) __builtin_trap();
}
}  // namespace

extern "C" int sgd_debug_occupy(int32_t blocks, float milliseconds, void* stream) {
    SGD_CLEAR_ERR();
    if (blocks <= 0 || blocks > 256 || !(milliseconds > 0.f) || milliseconds > 1000.f) return SGD_ERR_ARG;
    static bool attr = false;
    constexpr size_t LDS = 150 * 1024;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        attr = true;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(512), LDS, (hipStream_t)stream,
                       (unsigned long long)(milliseconds * 1.0e5f));
    return sgd_check_launch();
}

