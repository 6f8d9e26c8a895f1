/* C-ABI of libsgdm_hip_tools.so: DIAGNOSTICS ONLY (bench.py's in-run device calibration, tests/test_hip_contention.py, tools/).
 * Built from self-guided-diffusion-models_amd/csrc/tools/ by build.py next to the product library; the product path
 * (sgdm_amd/, dropin/) never loads it and libsgdm_hip.so exports none of these symbols (tests/test_boundary_cpu.py).
 * gfx950 only.  Return codes as in sgdm_hip.h (0 ok, 1 invalid argument, 2 launch failure). */
#ifndef SGDM_HIP_TOOLS_H
#define SGDM_HIP_TOOLS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic (tests/test_hip_contention.py, tools/): occupy `blocks` compute units -- one 512-thread block with 150 KB of
 * LDS each, so nothing else fits beside it -- for `milliseconds` of wall-clock time.  Stands in for RCCL's kernels on a
 * side stream in the single-GPU contention tests. */
int sgd_debug_occupy(int32_t blocks, float milliseconds, void* stream);
/* Diagnostics (bench.py: the in-run device calibration beside every roofline figure; csrc/tools/probe.hip).
 * sgd_debug_mfma_probe: `blocks` blocks of 4 waves (150 KB of LDS each: one block per compute unit, one wave per SIMD) run
 * `iters` x 8 independent v_mfma_f32_16x16x32_f16 on random register operands and nothing else;
 * sgd_debug_mfma_probe_flops gives the flop count of such a launch.  out: NULL or blocks * 256 floats (keeps the work live).
 * sgd_debug_copy_probe: dst[0..count) = src[0..count), 16 bytes per lane, four loads in flight (count % 4 == 0, both
 * pointers 16-byte aligned): the practical HBM rate of one read and one write stream. */
int sgd_debug_mfma_probe(int32_t blocks, int64_t iters, uint32_t seed, int32_t variant /* 0: 16x16x32 on random operands,
                         1: on zeros, 2: 32x32x16 on random operands */, float* out, void* stream);
int64_t sgd_debug_mfma_probe_flops(int32_t blocks, int64_t iters, int32_t variant);
int sgd_debug_copy_probe(const float* src, float* dst, int64_t count, int32_t variant /* 0: grid-stride, 4 loads in flight; 1: 8 in
                         flight; 2: 4 in flight, non-temporal; 3: contiguous 8 KiB per wave and piece; 4: as 3, non-temporal; 5: read only
                         (sums land in dst); 6: write only */, int32_t blocks /* 0: 2048 */, void* stream);
/* diagnostic: the conv kernel's compute-wave stream in isolation -- 48 split-precision 16x16x32 MFMAs per step into 64 accumulator
 * registers, the row-block operands re-read from LDS every step (row_blocks = 8: 16 reads per step, the shipped 128 x 32 wave tile;
 * 4: 8 reads, a 64 x 64 tile; 0: none), on one or two MFMA waves per SIMD.  out: blocks * 256 * waves_per_simd floats or NULL.
 * flops = blocks * 4 * waves_per_simd * iters * 48 * 16384 (tools/mfma_probe_sweep.py --lds) */
int sgd_debug_mfma_lds_probe(int32_t blocks, int64_t iters, uint32_t seed, int32_t row_blocks, int32_t waves_per_simd, float* out,
                             void* stream);
/* diagnostic: that stream on the shipped wave tile with the rest of the conv kernel's per-step traffic added piece by piece --
 * extras bit 0: the weight fragments re-loaded from global memory every step (wbuf: >= 1 MiB, stays in L2); bit 1: four loader
 * waves per block moving six 16-byte rows per thread and chunk of 9 steps from abuf (arows rows, streamed) through affine + SiLU +
 * hi / lo split into LDS; bit 2 (with bit 1): one barrier per chunk; bit 3 (with bit 1, instead of bit 2): producer / consumer
 * counters in LDS; bit 4: eight MFMA waves of 64 x 64 (two per SIMD) instead of four of 128 x 32 (wbuf >= 2 MiB).  flops = blocks * (4 or 8) * (iters / 9 * 9) * 48 * 16384 */
int sgd_debug_mfma_stream_probe(int32_t blocks, int64_t iters, uint32_t seed, int32_t extras, const void* wbuf, const void* abuf,
                                int64_t arows, float* out, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* SGDM_HIP_TOOLS_H */
