/* sgdm_hip.h -- C-ABI of libsgdm_hip.so: the MI355X (gfx950) kernels behind the
 * self-guided-diffusion drop-in (UNet denoiser evaluation, CFG sampling step, train step).
 *
 * The reference has no FFI of its own: it is pure Python on ATen, and its
 * operator boundary is the pair of nn.Module classes instantiated through the Hydra
 * `target:` strings.  This header is the boundary a maintainer binds
 * *below* those classes (ctypes stub in INTEGRATION.md): every entry point replaces the
 * ATen op sequence of the cited reference lines.
 *
 * Conventions
 *   - plain pointers are DEVICE pointers unless stated; sizes are in elements;
 *   - activations are fp32 NHWC: [N, H, W, C] == rows of C channels;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launchers are
 *     asynchronous, allocate nothing, keep no global state, and are graph-capturable;
 *   - return value: 0 ok, 1 invalid argument (nothing launched), 2 launch failure.
 */
#ifndef SGDM_HIP_H
#define SGDM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGD_ABI_VERSION 21
int sgd_abi_version(void);
/* 16 hex digits identifying the sources and flags the library was compiled from (build.py: source_id()); static storage.
 * __graft_entry__.build() and tests/test_boundary_cpu.py compare it with the tree on disk. */
const char* sgd_build_id(void);

/* --------------------------------------------------------------------------------------
 * Fused implicit-GEMM convolution / linear layer.
 *   y[row, :Cout] = epilogue( W * prologue(x)[row-neighbourhood] )
 * replaces nn.Conv2d 3x3 s1/s2 p1, nn.Conv2d/Conv1d 1x1, nn.Linear together with the
 * element-wise ops the reference runs around them:
 *   ResBlock            openaimodel.py:300-320  (GN+SiLU -> conv, FiLM GN -> SiLU -> conv, +skip)
 *   Up/Downsample       openaimodel.py:151,200 ; openaimodel_ca.py:128,167-174
 *   skip concat         openaimodel.py:950 (two input pointers, no materialised cat)
 *   qkv / proj_out      openaimodel.py:349,357,368-371
 *   Attention_LR linears  crossattetion_lr.py:71-79,86-88,103,140
 *   time/cond MLPs, emb_layers  openaimodel.py:570-574,603-607,262-268
 * -------------------------------------------------------------------------------------- */
enum { SGD_MODE_FLAT = 0, SGD_MODE_CONV3 = 1 };
enum { SGD_RS_NONE = 0, SGD_RS_AVGPOOL2 = 1, SGD_RS_UP2 = 2,
       SGD_RS_ZEROUP2 = 3 /* x2 zero insertion: input of the adjoint of a stride-2 conv (Downsample backward) */ };
enum { SGD_PRO_NONE = 0, SGD_PRO_AFFINE_NC = 1, SGD_PRO_LN_ROW = 2 };
enum { SGD_PREC_F32 = 0, SGD_PREC_F16X3 = 1, SGD_PREC_BF16X3 = 2 };

typedef struct sgd_igemm_args {
    /* input: channels [0,c0) from x0, [c0,c0+c1) from x1 (x1 may be NULL with c1 = 0) */
    const float* x0;
    const float* x1;
    int32_t c0, c1;
    int32_t mode;          /* SGD_MODE_* */
    int32_t n, hi, wi;     /* CONV3: input dims of x0/x1 (before resample).  FLAT: unused */
    int32_t ho, wo;        /* CONV3: output spatial dims */
    int32_t m;             /* FLAT: number of rows.  CONV3: ignored (n*ho*wo) */
    int32_t rows_per_n;    /* FLAT + PRO_AFFINE_NC: rows per batch element (n = row / rows_per_n) */
    int32_t stride;        /* CONV3: 1 or 2 */
    int32_t resample;      /* CONV3: SGD_RS_* applied to the (activated) input before the conv */
    /* prologue on every input element */
    int32_t pro;           /* SGD_PRO_* */
    int32_t pro_silu;      /* 1: SiLU after the prologue affine (or alone) */
    const float* pa;       /* AFFINE_NC: a[n, C]      LN_ROW: stats[row, 2] = (mean, rstd) */
    const float* pb;       /* AFFINE_NC: b[n, C]      LN_ROW: gamma[C] */
    const float* pc;       /*                         LN_ROW: beta[C] or NULL */
    /* weights packed by sgd_pack_weight: MFMA fragment order, 4 KiB units [cin_p/32][taps][cout_p/32] of
       32 output x 32 input channels (csrc/igemm.hip: pack_weight_kernel); opaque to the caller */
    const void* w;
    int32_t cin_p, cout_p;
    const float* bias;     /* [cout] or NULL */
    /* epilogue */
    const float* res;      /* residual tensor (cout channels) or NULL */
    int32_t res_mode;      /* SGD_RS_NONE: same rows; AVGPOOL2: res is at 2x resolution; UP2: at 1/2 */
    float* y;
    int32_t cout;
    int32_t y_ld;          /* row stride of y in floats (>= cout) */
    /* output row remap: out_row = (row / orows_in) * orows_out + orow_off + row % orows_in
       (orows_in == 0: identity).  Lets to_kv / to_context write into the shared K/V buffer. */
    int32_t orows_in, orows_out, orow_off;
    int32_t prec;          /* SGD_PREC_* */
    /* train-time dropout on the activated input (nn.Dropout in ResBlock.out_layers, openaimodel.py:272):
     * element (row, channel) is kept iff sgd_drop_hash(seed, row*C + channel) >= drop_p, survivors scaled by
     * 1/(1-p).  Counter-based: forward, wgrad and the GroupNorm backward recompute the same mask, nothing stored. */
    float drop_p;          /* 0: off */
    uint32_t drop_seed;
    /* GroupNorm statistics of the OUTPUT, produced by the epilogue (the consumer's group_norm,
     * openaimodel.py:246-247, then needs no extra pass over y): per-(image, tile-part, channel) partial
     * (sum, sum of squares) of the stored values, layout [n, parts, 2, cout]; parts =
     * sgd_igemm_stats_parts(args) (0: this geometry cannot produce them; leave stats NULL).  Folded into
     * the sums[n, c, 2] layout of sgd_chan_stats by sgd_stats_reduce.  NULL: off. */
    float* stats;
    /* Per-tensor power-of-two scale of the packed weights (split-precision modes): the packed values are w * 2^k with k
     * chosen by sgd_pack_weight_scaled so that max|w| * 2^k lies in [1, 2) -- the 16-bit hi AND lo halves then sit in
     * fp16's normal range whatever the magnitude of the tensor (zero-initialised convs early in training are ~1e-4: their
     * lo halves would be fp16 subnormals with 2-3 significant bits).  The epilogue multiplies the accumulator by
     * *w_scale_inv = 2^-k (exact) before bias / residual.  DEVICE pointer to one float; NULL: 1. */
    const float* w_scale_inv;
    /* Optional scratch for the balanced tail of the persistent schedule.  A launch whose tile count is not a multiple of
     * the 256 resident blocks ends in a partial round; with a workspace the tiles of that round are split along K over the
     * otherwise idle blocks: the blocks of the first K parts store their fp32 partial accumulators here, the block of the
     * last part adds them (fixed order: deterministic) and runs the epilogue.  DEVICE buffer of sgd_igemm_work_bytes()
     * bytes, ZEROED ONCE by the caller (the arrival counters at its head reset themselves), reusable by any number of
     * launches that are ordered on one stream.  NULL (or too small): plain schedule, same results bit for bit except
     * for the summation order of the split tiles. */
    void* work;
    int64_t work_bytes;
    /* Persistent-grid cap (round 4).  The kernel runs one 512-thread block per compute unit (all of a CU's registers), so a
     * launch sized to the whole device cannot share it: while another stream's kernel holds CUs -- RCCL's all-reduce
     * during the overlapped backward (config/pl/default.yaml:2) -- the blocks that find no CU start only
     * when others exit, and with the static tile lists that is up to 2x the launch time.  grid_cap > 0 launches at most
     * grid_cap (rounded down to a multiple of 8, >= 8) blocks and leaves the remaining CUs to the other stream; 0 = every
     * CU of the device (hipDeviceProp.multiProcessorCount).  Results do not depend on it bit for bit, except for the
     * summation order of the balanced tail's split tiles. */
    int32_t grid_cap;
    /* Per-call schedule overrides, SGD_TUNE_* bits (0 = the launcher's own rules, which is what the product passes).  They
     * select among code paths that all return the same result (bit for bit, except where a flag changes the summation order
     * of split tiles or slabs): parity tests pin one path against another, A/B tools time them.  A field of the call --
     * the library reads NO environment variable and keeps no global state. */
    int32_t tune;
} sgd_igemm_args;
enum {
    SGD_TUNE_BN128 = 1,            /* sgd_igemm: the 128-column tile even where the launcher would take 128 x 256 */
    SGD_TUNE_BN256 = 2,            /* sgd_igemm: the 128 x 256 tile wherever the shape allows */
    SGD_TUNE_FLAT2 = 4,            /* sgd_igemm, 1x1 / linear: two 32-channel planes per barrier (never with PRO_LN_ROW) */
    SGD_TUNE_DEFER = 8,            /* sgd_igemm, 3x3 split modes: the epilogue on the loader waves */
    SGD_TUNE_PLAIN_SCHEDULE = 16,  /* sgd_igemm: no balanced tail even with a workspace */
    SGD_TUNE_NO_SMALL = 64,        /* sgd_igemm: the 128-column tile even for launches of fewer such tiles than an eighth of the device */
    SGD_TUNE_LN_PACKED = 32,       /* sgd_igemm, PRO_LN_ROW in a split mode: the regular instances (packed-f32 code generation on)
                                      instead of the no-packed-f32 ones the launcher takes for that prologue (tools/ln_hazard.py) */
    SGD_TUNE_WGRAD_GENERIC_NARROW = 256,  /* sgd_wgrad: stem / head on the generic kernels instead of wgrad_narrow_kernel */
    SGD_TUNE_WGRAD_NO_POOLED_PLANES = 512,/* sgd_wgrad: fused-average-pool convs on the per-tap kernel (no pooled planes) */
    SGD_TUNE_WGRAD_F32 = 1024,            /* sgd_wgrad: the exact-f32 per-tap kernel in every mode */
    SGD_TUNE_WGRAD_NO_WS = 2048,          /* sgd_wgrad: the round-2 all-taps kernel instead of the wave-specialised one */
    SGD_TUNE_WGRAD_NO_PLANES = 4096,      /* sgd_wgrad: no pre-split operand planes even with scratch */
    SGD_TUNE_WGRAD_NO_PIPE = 8192,        /* sgd_wgrad, 1x1 / linear: synchronous staging (no register pipelining) */
    SGD_TUNE_WGRAD_PLANES_ALWAYS = 16384  /* sgd_wgrad: pre-split planes also for a single 128-channel output tile */
};
int64_t sgd_igemm_work_bytes(void);
/* Balanced-tail health word (DEVICE int32 inside the workspace, byte offset sgd_igemm_work_status_offset()): 0 after a
 * clean run.  A finisher whose producers did not arrive within its bounded poll (~2 s; cannot happen when the launches
 * that share the workspace are ordered on one stream) sets it to 1 and multiplies its outputs by NaN instead of spinning
 * for ever: check it after a run that produced NaN, zero the whole workspace before using it again. */
int64_t sgd_igemm_work_status_offset(void);
/* Host-only test hook (no launch): the balanced-tail workspace layout of a launch of `total_tiles` tiles with `nchunks`
 * 32-channel chunks per tile and `taps` (9 / 1) K steps per chunk on `grid` persistent blocks.  out[4*b .. 4*b+3] =
 * {K split of block b's last tile (0: none), index of its arrival counter, first producer slab, producer slabs}.
 * Two split tiles of one launch must never share a counter or a slab (tests/test_boundary_cpu.py). */
int sgd_igemm_tail_layout(int32_t total_tiles, int32_t nchunks, int32_t taps, int32_t grid, int32_t* out);

/* the keep/drop hash, shared by device code and host tests (ABI 17: one hash per PAIR of elements):
 *   pair = index >> 1  (index = row * channels + channel);  lo/hi = the 32-bit halves of pair
 *   h = seed ^ (lo * 0x9E3779B1) ^ (hi * 0x632BE5AB); h ^= h>>16; h *= 0x85EBCA6B; h ^= h>>13; h *= 0xC2B2AE35; h ^= h>>16;
 *   keep  <=>  (index & 1 ? h >> 16 : h & 0xFFFF) >= (uint32_t)(p * 65536) */

int sgd_igemm(const sgd_igemm_args* args /* HOST pointer */, void* stream);
/* number of per-image partial-statistics slots the epilogue of this launch writes (see args->stats) */
int sgd_igemm_stats_parts(const sgd_igemm_args* args /* HOST pointer */);

/* bytes of the packed weight buffer for given dims; w_src is [cout, cin, k, k] (OIHW) or [cout, cin] */
int64_t sgd_packed_weight_bytes(int32_t cout, int32_t cin, int32_t ksize, int32_t prec);
int sgd_pack_weight(const float* w_src, void* w_dst, int32_t cout, int32_t cin, int32_t ksize,
                    int32_t prec, int32_t* cin_p, int32_t* cout_p /* HOST out */, void* stream);

/* sgd_pack_weight / sgd_pack_weight_dgrad with the per-tensor power-of-two scale described at sgd_igemm_args.w_scale_inv:
 * amax_bits is a DEVICE uint32 the caller zeroes; sgd_weight_amax (one or more calls: concatenated sources) folds
 * max|w| into it (bit pattern of a non-negative float, atomicMax); the packer derives 2^k from it, scales before the
 * hi/lo split and writes 2^-k to scale_inv_out (DEVICE float).  All on the stream, no host round trip. */
int sgd_weight_amax(const float* w, int64_t count, uint32_t* amax_bits, void* stream);
int sgd_pack_weight_scaled(const float* w_src, void* w_dst, int32_t cout, int32_t cin, int32_t ksize, int32_t prec,
                           int32_t transpose /* 1: dgrad operator, dims are the FORWARD weight's */,
                           const uint32_t* amax_bits, float* scale_inv_out, int32_t* cin_p, int32_t* cout_p /* HOST out */,
                           void* stream);

/* Every weight of a model in three launches (the per-step re-pack of a training loop: one zero, one amax, one pack kernel
 * instead of two launches per tensor).  jobs / the block tables are DEVICE arrays the caller builds once:
 *   jobs[j]          one weight tensor: src (OIHW fp32), dst (packed buffer of sgd_packed_weight_bytes), dims of the FORWARD
 *                    weight, transpose (1: the dgrad operator), amax_bits / scale_inv as in sgd_pack_weight_scaled (NULL in
 *                    f32 mode), own_amax (0: amax_bits belongs to another job of the same tensor and is only read)
 *   *_block_job[b]   the job block b works on;  *_first[j] the first block of job j (n_jobs + 1 entries, ascending);
 *                    a job with own_amax == 0 has no amax blocks.  sgd_pack_job_blocks gives the block counts per job
 *                    (what the single-tensor entry points would launch) and the padded dims the packed operator has. */
typedef struct sgd_pack_job {
    const float* src;
    void* dst;
    uint32_t* amax_bits;
    float* scale_inv;
    int32_t cout, cin, ksize, transpose;
    int32_t own_amax, reserved0;
} sgd_pack_job;
int sgd_pack_job_blocks(int32_t cout, int32_t cin, int32_t ksize, int32_t prec, int32_t transpose, int32_t* amax_blocks,
                        int32_t* pack_blocks, int32_t* cin_p, int32_t* cout_p /* HOST out */);
int sgd_pack_weights_batched(const sgd_pack_job* jobs, int32_t n_jobs, const int32_t* amax_block_job, const int32_t* amax_first,
                             int32_t n_amax_blocks, const int32_t* pack_block_job, const int32_t* pack_first,
                             int32_t n_pack_blocks, int32_t prec, void* stream);

/* Skinny linear with a long reduction (mlp_cond.0 of the cluster-k5000 config, openaimodel.py:597-607: [2B, 5000] x
 * [256, 5000]^T): exact fp32 FMA, the K range split over `ksplit` blocks per 64-column tile, partial sums in
 * work[ksplit, m, n] folded in fixed order (deterministic).  w is the nn.Linear weight as stored ([n, k], no packing).
 *   y[m, n] = x[m, :k] . w[n, :k] + bias[n]                                (bias may be NULL) */
int sgd_linear_splitk(const float* x, int32_t x_ld, const float* w, const float* bias, int32_t m, int32_t n, int32_t k,
                      float* work, int32_t ksplit, float* y, int32_t y_ld, void* stream);
/* y[m, n] = x[m, k] * w[k][n] (weight indexed [k][n], row stride w_ld; no bias): the INPUT gradient of a wide linear layer
 * with few rows -- x = the output gradient [m, k = out_features], w = the nn.Linear weight as stored ([out, in]; a column
 * window of it through w + offset).  Same split-K scheme and scratch as sgd_linear_splitk: work[ksplit, m, n], m <= 256. */
int sgd_linear_splitk_t(const float* x, int32_t x_ld, const float* w, int32_t w_ld, int32_t m, int32_t n, int32_t k,
                        float* work, int32_t ksplit, float* y, int32_t y_ld, void* stream);

/* --------------------------------------------------------------------------------------
 * GroupNorm(32) as statistics + per-(n,c) affine coefficients consumed by sgd_igemm's
 * prologue (util.py:199-216; openaimodel.py:246-247,270-271,312-316,348,831-832).
 *   sums[n, c, 2] = (sum, sum of squares) over the HW rows of x[n]   (chan offset for concat)
 *   a, b such that GN(x)*(1+scale)+shift == x*a + b
 * -------------------------------------------------------------------------------------- */
int sgd_chan_stats(const float* x, int32_t n, int32_t hw, int32_t c,
                   float* sums /* [n, c_total, 2] */, int32_t c_total, int32_t c_off, void* stream);
/* ResBlock without scale-shift norm (use_scale_shift_norm=False, openaimodel.py:317-319): x[n, hw, c] += e[n, c] in place
 * (e row stride e_ld; c, e_ld multiples of 4; 16-byte aligned).  The statistics of the sum are taken by sgd_chan_stats. */
int sgd_add_rows_nc(float* x, const float* e, int32_t e_ld, int32_t n, int64_t hw, int32_t c, void* stream);
/* sums[n, c_off + c, 2] = sum over parts of partial[n, parts, 2, c] (partials written by sgd_igemm's epilogue) */
int sgd_stats_reduce(const float* partial, int32_t n, int32_t parts, int32_t c,
                     float* sums /* [n, c_total, 2] */, int32_t c_total, int32_t c_off, void* stream);
/* sgd_stats_reduce of up to two concatenated sources (c = c0 + c1; partsX == 0: that source's sums are already in
 * `sums`, written by sgd_chan_stats) followed by sgd_gn_coef, in one launch.  `sums` is written for the sources with
 * partials (the training backward reads it); the coefficients are bit-identical to the two-launch route. */
int sgd_gn_coef_parts(const float* partial0, int32_t parts0, int32_t c0, const float* partial1, int32_t parts1, int32_t c1,
                      float* sums, const float* gamma, const float* beta, const float* film, int32_t film_ld,
                      int32_t n, int32_t groups, int32_t hw, float eps, float* a, float* b, void* stream);
int sgd_gn_coef(const float* sums, const float* gamma, const float* beta,
                const float* film /* [n, film_ld] scale at +0, shift at +c; or NULL */, int32_t film_ld,
                int32_t n, int32_t c, int32_t groups, int32_t hw, float eps,
                float* a, float* b, void* stream);

/* LayerNorm over the channel dim of rows (crossattetion_lr.py:36-43; openaimodel_ca.py:583,1017) */
int sgd_ln_stats(const float* x, int32_t rows, int32_t c, float eps, float* stats /* [rows,2] */, void* stream);
/* out = (res ? res : 0) + LN(x) * gamma + (beta ? beta : 0) */
int sgd_ln_apply(const float* x, const float* gamma, const float* beta, const float* res,
                 int32_t rows, int32_t c, float eps, float* out, void* stream);

/* --------------------------------------------------------------------------------------
 * Attention cores.
 *  legacy QKV self-attention  (QKVAttentionLegacy, openaimodel.py:403-420)
 *  multi-query attention over [context | null | self] keys (crossattetion_lr.py:90-139)
 * q rows: [b, tq, *] with row stride q_ld, head h at +h*q_hs; k/v rows: [b, tk, *] with row
 * stride kv_ld, head h at +h*kv_hs (0 for multi-query).  out[b, tq, heads*d] head-major.
 * softmax(scale * q.k) ; d in {16, 32, 64, 128}.
 * sgd_attention: exact fp32 MFMA.  sgd_attention_split: the same contract in split precision (every fp32 operand as
 * hi + lo f16, three products, fp32 accumulate -- the arithmetic of SGD_PREC_F16X3); out must be 16-byte aligned.
 * -------------------------------------------------------------------------------------- */
int sgd_attention(const float* q, int32_t q_ld, int32_t q_hs,
                  const float* k, const float* v, int32_t kv_ld, int32_t kv_hs,
                  int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d, float scale,
                  float* out, int32_t out_ld,
                  float* lse /* [batch, heads, tq] log-sum-exp of the scaled logits for the backward, or NULL */,
                  void* stream);
int sgd_attention_split(const float* q, int32_t q_ld, int32_t q_hs,
                        const float* k, const float* v, int32_t kv_ld, int32_t kv_hs,
                        int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d, float scale,
                        float* out, int32_t out_ld, float* lse, void* stream);

/* --------------------------------------------------------------------------------------
 * Small glue kernels on the UNet boundary.
 * -------------------------------------------------------------------------------------- */
/* timestep_embedding (util.py:151-171): t int64 [n_src], row r uses t[r % n_src] -> out[n, dim];
 * freqs[dim/2] = exp(-ln(1e4) * arange(dim/2) / (dim/2)) as fp32 (host-built table, util.py:162-164) */
int sgd_timestep_embedding(const int64_t* t, const float* freqs, int32_t n_src, int32_t n, int32_t dim,
                           float* out, void* stream);
/* cond select (openaimodel.py:929-931): out[r,:] = mask[r] ? null_row : cond[r % n_src,:]; cond is
 * float32 (is_i64 = 0), int64 one-hot (is_i64 = 1, cast as openaimodel.py:911) or -- is_i64 = 2 -- the int64 class /
 * cluster IDS [n_src] themselves, expanded to the one-hot row here (F.one_hot of dataset/ds_utils/
 * unsupervised_cluster.py:33-46, supervised_label.py:31-40 moved onto the device: 8 bytes per sample cross PCIe
 * instead of 8*k) */
int sgd_cond_select(const void* cond, int32_t is_i64, const uint8_t* mask, const float* null_row,
                    int32_t n_src, int32_t n, int32_t k, float* out, void* stream);
/* x NCHW [n_src,cx,h,w] (+ layout NCHW [n_src,cl,h,w] masked by mask -> null_layout[h*w]) -> NHWC
 * [n, h, w, cx+cl]  (openaimodel.py:933-939, 944; batch doubling of :886-891 via r % n_src) */
int sgd_pack_input(const float* x, const float* layout, const uint8_t* mask, const float* null_layout,
                   int32_t n_src, int32_t n, int32_t cx, int32_t cl, int32_t h, int32_t w,
                   float* out, void* stream);
/* NHWC [n,h,w,c] -> NCHW [n,c,h,w] */
/* The same with the layout in its compact on-disk form, expanded on the device exactly as the reference's data
 * workers expand it on the CPU (dataset/transforms/complex_ds_common_util.py):
 *   layout_fmt 1: uint8 label map [n_src,h,w] -> one-hot over cl channels, label 255 -> class 0 (stego_to_onehotmask :118-123)
 *   layout_fmt 2: int32 boxes [n_src,4] = (x0,y0,x1,y1) in the h x w frame -> mask[y0:y1, x0:x1] = 1, cl == 1
 *                 (get_lostbboxmask :151-162) */
int sgd_pack_input_compact(const float* x, const void* layout, int32_t layout_fmt, const uint8_t* mask,
                           const float* null_layout, int32_t n_src, int32_t n, int32_t cx, int32_t cl, int32_t h,
                           int32_t w, float* out, void* stream);
/* n-hot attribute vector of a label map (stegomask_to_attr_nhot, complex_ds_common_util.py:126-133): out[b, j] = 1 if
 * label j occurs in labels[b, :], k <= 256 */
int sgd_labelmap_nhot(const uint8_t* labels, int32_t b, int32_t hw, int32_t k, float* out, void* stream);
/* y[r, c] = mask[r] ? nullproj[c] : w[c, ids[r % n_src]] + bias[c] -- the first Linear of mlp_cond (openaimodel.py:
 * 597-607) on a one-hot input as a column gather; bit-identical to the dense product (the other k-1 terms are exact
 * zeros).  w is the nn.Linear weight as stored [nout, k]; nullproj = w . null_cond_emb + bias (computed by the caller
 * with sgd_linear_splitk when the weights change) */
int sgd_linear_gather(const int64_t* ids, const uint8_t* mask, const float* w, const float* bias, const float* nullproj,
                      int32_t n_src, int32_t n, int32_t nout, int32_t k, float* out, int32_t ldo, void* stream);
/* the same Linear on the reference's own input format for the cluster / label methods -- an int64 one-hot ROW per sample, cond
 * [n_src, k] (unsupervised_cluster.py:33-46, supervised_label.py:31-40): y[r, :] = bias + sum over the non-zero entries of row
 * r % n_src of value * w[:, entry]; any integer row (multi-hot, counts), entries summed in ascending k.  k <= 7,936 (the list of
 * entries lives in LDS); SGD_ERR_ARG above that -- the caller keeps the dense sgd_linear_splitk for longer rows */
int sgd_linear_sparse_rows(const int64_t* cond, const uint8_t* mask, const float* w, const float* bias, const float* nullproj,
                           int32_t n_src, int32_t n, int32_t nout, int32_t k, float* out, int32_t ldo, void* stream);

int sgd_nhwc_to_nchw(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, float* out, void* stream);
/* broadcast null_kv rows into the shared K/V buffer (crossattetion_lr.py:95-97):
 * kv[b, row, 0:d] = null_kv[0], kv[b, row, d:2d] = null_kv[1] */
int sgd_fill_null_kv(const float* null_kv, int32_t batch, int32_t rows_per_b, int32_t row, int32_t d,
                     float* kv, void* stream);

/* --------------------------------------------------------------------------------------
 * Sampler step kernels (one launch per step).  eps_nhwc is the UNet output at 2B
 * ([cond ; uncond] halves, NHWC) when cfg_mode != 0, else at B.
 *   guided = cfg_mode 1 (imagen): (1-w)*eps_u + w*eps_c     (openaimodel.py:855)
 *            cfg_mode 2 (cfg)   : (1+w)*eps_c - w*eps_u     (openaimodel.py:857)
 * x, z, x_out, x0_out are NCHW [b,c,h,w].
 * DDPM ancestral step: ddpm_sampler.py:132-137,154-192.  coef = {sqrt_recip_ac, sqrt_recipm1_ac,
 * post_mean_coef1, post_mean_coef2, exp(0.5*post_log_var)*nonzero*temperature} for this t.
 * DDIM step: ddim_plms_sampler.py:346-391. coef = {sqrt_one_minus_at, 1/sqrt(a_t) is NOT used
 * (the reference divides by a_t.sqrt()), a_t, a_prev, sigma_t*temperature, 0}.
 * -------------------------------------------------------------------------------------- */
int sgd_ddpm_step(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                  const float* coef /* HOST [5] */, int32_t clip, int32_t b, int32_t c, int32_t hw,
                  float* x_out, float* x0_out, void* stream);
int sgd_ddim_step(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                  const float* coef /* HOST [4]: sqrt_one_minus_at, a_t, a_prev, sigma_t */, float temperature,
                  int32_t clip, int32_t b, int32_t c, int32_t hw, float* x_out, float* x0_out, void* stream);
/* The same two steps with the per-step coefficients read from DEVICE memory (coef_dev[5] / coef_dev[4], same meaning as
 * above) and x updated IN PLACE (x_out == x allowed: the update is elementwise).  With the UNet inputs (x, t) and these
 * coefficients in fixed device buffers, one sampling step = UNet program + this launch has no by-value argument that
 * changes between steps, so it can be captured into a hipGraph once per trajectory and replayed
 * (sgdm_amd/diffusion.py: _GraphedStep; reference loop: ddpm_sampler.py:194-238, ddim_plms_sampler.py:302-344). */
int sgd_ddpm_step_dev(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                      const float* coef_dev, int32_t clip, int32_t b, int32_t c, int32_t hw,
                      float* x_out, float* x0_out, void* stream);
int sgd_ddim_step_dev(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                      const float* coef_dev, float temperature, int32_t clip, int32_t b, int32_t c, int32_t hw,
                      float* x_out, float* x0_out, void* stream);
/* ((x+1)*127.5).clamp(0,255).to(uint8)  (diffusion_utils/util.py:99-100) */
/* Dynamic thresholding (sampling kwarg dtp < 1; clip_x0_minus_one_to_one, diffusion_utils/util.py:70-79):
 *   s[n] = max(1, quantile(|x0[n]|, dtp)),  x0 <- clamp(x0, -s, s) / s
 * sgd_x0_quantile forms x0 like the step kernels (kind 0: DDPM coef[5], kind 1: DDIM coef[4]) and selects the two order
 * statistics lo / hi of |x0| per sample (exact radix select), interpolating with frac like torch.quantile;
 * the *_dyn steps are sgd_ddpm_step / sgd_ddim_step with that per-sample scale instead of the static clip. */
int sgd_x0_quantile(int32_t kind, const float* x, const float* eps_nhwc, int32_t cfg_mode, float w,
                    const float* coef /* HOST */, int32_t b, int32_t c, int32_t hw, int32_t lo, int32_t hi, float frac,
                    float* s_out /* [b] */, void* stream);
int sgd_ddpm_step_dyn(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                      const float* coef /* HOST [5] */, const float* dyn_s, int32_t b, int32_t c, int32_t hw,
                      float* x_out, float* x0_out, void* stream);
int sgd_ddim_step_dyn(const float* x, const float* eps_nhwc, const float* z, int32_t cfg_mode, float w,
                      const float* coef /* HOST [4] */, float temperature, const float* dyn_s, int32_t b, int32_t c,
                      int32_t hw, float* x_out, float* x0_out, void* stream);

/* pooled guidance token of the token-guidance path (cond_token_num > 1, openaimodel_ca.py:999-1004):
 * cond [n, tokens, c] -> out [n, c] = cond[:, 0, :] (cls = 1, use_cls_token_as_pooled) or mean over tokens (cls = 0) */
int sgd_token_pool(const float* cond, int32_t n, int32_t tokens, int32_t c, int32_t cls, float* out, void* stream);

/* GEGLU gate of the SpatialTransformer feed-forward (reference dynamic/attention.py:38-45, GEGLU.forward):
 * in [rows, 2*inner] = Linear(x) -> out[r, c] = in[r, c] * gelu(in[r, inner + c]), exact erf GELU; inner % 4 == 0 */
int sgd_geglu(const float* in, int64_t rows, int32_t inner, float* out, void* stream);

int sgd_to_uint8(const float* x, int64_t count, uint8_t* out, void* stream);
/* guided eps only (forward_with_cond_scale return value): eps_nhwc [2b,h,w,c] -> NCHW [b,c,h,w] */
int sgd_cfg_combine(const float* eps_nhwc, int32_t cfg_mode, float w, int32_t b, int32_t c, int32_t hw,
                    float* out_nchw, void* stream);


/* weights of the adjoint convolution (input gradient): w_src is the FORWARD weight [cout, cin, k, k];
 * the packed operator maps cout input channels to cin output channels with flipped taps, so
 * sgd_igemm(dy, w_dgrad) == conv_transpose(dy, W) for stride 1 (autograd of nn.Conv2d, openaimodel.py:248,274). */
int sgd_pack_weight_dgrad(const float* w_src, void* w_dst, int32_t cout_fwd, int32_t cin_fwd, int32_t ksize,
                          int32_t prec, int32_t* cin_p, int32_t* cout_p /* HOST out */, void* stream);

/* --------------------------------------------------------------------------------------
 * Training step: backward kernels (autograd of the ops above; the reference gets these from
 * torch.autograd through ResBlock / AttentionBlock, lightning_module.py:215-245 + PL's loss.backward()).
 * -------------------------------------------------------------------------------------- */
/* weight gradient of a fused conv / linear:  dW[co, ci, tap] = sum_rows gy[row, co] * act(x)[row shifted by tap, ci]
 * `fwd` describes the FORWARD launch (its input side: x0/x1, prologue, resample, geometry are re-used to recompute the
 * activated input; its w/bias/res/y fields are ignored).  Arithmetic follows fwd->prec: exact-fp32 MFMA, or the
 * three-product 16-bit split (3x3 stride-1 and 1x1 / linear; other geometries fall back to exact fp32).  Partial sums
 * of `ksplit` row slices are written to `slabs` [ksplit][taps][cout][cin] and folded by sgd_wgrad_reduce into the
 * reference layout.  bias_slabs (optional, [sgd_wgrad_bias_rows(..)][cout]): partial column sums of gy = the bias
 * gradient, taken from the gy rows the kernel (or its pre-pass) stages anyway; fold with sgd_wgrad_reduce_bias /
 * sgd_colsum_fold over that many rows. */
int sgd_wgrad(const sgd_igemm_args* fwd /* HOST */, const float* gy, int32_t gy_ld, int32_t cout,
              float* slabs, int32_t ksplit, float* bias_slabs, void* stream);
/* sgd_wgrad with a scratch buffer (DEVICE, sgd_wgrad_scratch_bytes(fwd, cout) bytes; any launch may reuse it): 3x3 stride-1
 * convolutions in the split-precision modes first write the gradient rows and the activated input once as 16-bit hi / lo
 * planes (two element-wise passes) and the weight-gradient kernel's loader waves only copy them -- without it every block
 * repeats the GroupNorm-affine + SiLU + split of its operands (cin / 32 resp. cout / 128 times per element).  Same
 * results bit for bit; cases the planes form does not cover run exactly as sgd_wgrad. */
int64_t sgd_wgrad_scratch_bytes(const sgd_igemm_args* fwd /* HOST pointer */, int32_t cout);
/* rows of bias_slabs the launch writes (no launch, host arithmetic): ksplit, or -- planes form, whose pre-pass sums the
 * gradient rows in ~1024 / (cout / 128) row chunks -- the number of chunks.  scratch_bytes: what sgd_wgrad_scratch will be
 * given (0: sgd_wgrad). */
int sgd_wgrad_bias_rows(const sgd_igemm_args* fwd /* HOST pointer */, int32_t cout, int32_t gy_ld, int32_t ksplit,
                            int64_t scratch_bytes);
int sgd_wgrad_scratch(const sgd_igemm_args* fwd /* HOST pointer */, const float* gy, int32_t gy_ld, int32_t cout,
                      float* slabs, int32_t ksplit, float* bias_slabs, void* scratch, int64_t scratch_bytes, void* stream);
/* out[c] (+)= scale * sum_k partial[k, c]   (fixed order, double accumulation) */
int sgd_colsum_fold(const float* partial, int32_t chunks, int32_t c, float* out, int32_t accumulate, float scale,
                    void* stream);
/* dw[co, ci, tap] (OIHW / [cout, cin]) = (accumulate ? dw : 0) + scale * sum_k slabs[k][tap][co][ci]
 * (`scale` undoes the power-of-two gradient scaling that keeps the split-f16 dgrad operands in range) */
int sgd_wgrad_reduce(const float* slabs, int32_t ksplit, int32_t taps, int32_t cout, int32_t cin,
                     float* dw, int32_t accumulate, float scale, void* stream);
/* sgd_wgrad_reduce + the bias gradient of the same layer in one launch: bias_slabs [bias_rows, cout] are the partial column
 * sums the weight-gradient launch wrote (bias_rows = sgd_wgrad_bias_rows); dbias[cout] = scale * their sum (fixed order,
 * double accumulation). */
int sgd_wgrad_reduce_bias(const float* slabs, int32_t ksplit, int32_t taps, int32_t cout, int32_t cin, float* dw,
                          int32_t accumulate, float scale, const float* bias_slabs, int32_t bias_rows, float* dbias,
                          void* stream);
/* column sums of two [rows, c] matrices (row stride ld) in one launch, rows <= 256: GroupNorm's dgamma and dbeta from the
 * per-sample tables of sgd_gn_bwd_coef. */
int sgd_colsum_pair(const float* g1, const float* g2, int32_t rows, int32_t c, int32_t ld, float* out1, float* out2,
                    int32_t accumulate, float scale, void* stream);
/* out[c] = (accumulate ? out[c] : 0) + scale * sum_rows g[row, c]   (bias / norm-affine gradients).
 * Deterministic two-stage reduction through `work` [work_chunks, c] (caller-owned scratch, e.g. 256 chunks). */
int sgd_colsum(const float* g, int32_t rows, int32_t c, int32_t ld, float* out, int32_t accumulate, float scale,
               float* work, int32_t work_chunks, void* stream);

/* GroupNorm(+FiLM)+SiLU backward, split like the forward (util.py:199-216, openaimodel.py:246-247,312-316):
 *   pre = a*x + b ; u = silu ? SiLU(pre) : pre ; the consumer returned gu = dL/du.
 *   gu may live at another resolution (ResBlock up/down, openaimodel.py:301-306): gu_mode SGD_RS_NONE same rows,
 *   SGD_RS_AVGPOOL2: forward pooled u 2x2 => gu is at 1/2 resolution, each x pixel receives gu/4;
 *   SGD_RS_UP2: forward upsampled u x2 => gu is at 2x resolution, each x pixel receives the sum of its 2x2 block.
 * sgd_gn_bwd_reduce: S[n, c, 2] = (sum gpre, sum gpre * x) over the hw rows, gpre = gu * SiLU'(pre)      */
int sgd_gn_bwd_reduce(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t c_total, int32_t c_off,
                      const float* a, const float* b, int32_t silu,
                      const float* gu, int32_t gu_ld, int32_t gu_mode, float drop_p, uint32_t drop_seed,
                      float* S, void* stream);
/* per-(n,c) coefficients of dx = A*gpre + B*x + C from S and the forward statistics (`sums` of sgd_chan_stats),
 * plus per-sample parameter gradients dgamma_nc / dbeta_nc [n, c] (sum over n with sgd_colsum) and the FiLM
 * gradient dfilm[n, film_ld] (scale grad at +0, shift grad at +c; may be NULL). */
/* ABI 19: S is [n, c, 2] (s_chunks = 1: sgd_gn_bwd_reduce) or the per-chunk partial sums [n, s_chunks, c, 2] of sgd_gn_bwd_reduce_rows,
 * which the coefficient kernels fold in chunk order themselves. */
int sgd_gn_bwd_coef(const float* S, int32_t s_chunks, const float* sums, const float* gamma, const float* beta,
                    const float* film, int32_t film_ld, int32_t n, int32_t c, int32_t groups, int32_t hw, float eps,
                    float* A, float* B, float* Cc, float* dgamma_nc, float* dbeta_nc /* [n, c]: fold with sgd_colsum */,
                    float* dfilm, void* stream);
/* dx[row, c_off + c] (+)= A*gu*SiLU'(a x + b) + B*x + C  (+ extra residual-path gradient gres, same gu_mode rules)
 * written into dst (row stride dst_ld, channel offset dst_off); accumulate: add to what is there. */
/* sgd_gn_bwd_coef + sgd_colsum_pair(dgamma_nc, dbeta_nc) in one launch (n <= 256, 8 * n * c / groups + 32 * n + 4 <= 64 KiB of LDS, else
 * SGD_ERR_ARG: use the two calls): A, B, Cc, dfilm as sgd_gn_bwd_coef; dgamma[c] / dbeta[c] (+)= scale * column sums over the
 * images, bit-identical to the two-call route. */
int sgd_gn_bwd_coef_fold(const float* S, int32_t s_chunks, const float* sums, const float* gamma, const float* beta, const float* film,
                         int32_t film_ld, int32_t n, int32_t c, int32_t groups, int32_t hw, float eps, float* A, float* B,
                         float* Cc, float* dfilm, float* dgamma, float* dbeta, int32_t accumulate, float scale, void* stream);
/* ABI 19: the reduce pass as a row stream (same-resolution gradient only; csrc/backward.hip "Round 6"): every wave walks contiguous
 * 8 KiB pieces of the tensors' rows, a block = one chunk of an image.  sgd_gn_bwd_rows_chunks: chunks per image of a shape, 0 = not
 * served (fewer than 32 x 32 pixels, channel count not a power of two in 16 .. 1024, image not a whole number of 32 KiB windows:
 * use sgd_gn_bwd_reduce).  sgd_gn_bwd_reduce_rows: P[n, chunks, c_total, 2] partial sums (any `chunks` that divides the shape's own
 * count: the sources of a concatenated GroupNorm share one table), folded by sgd_gn_bwd_coef(_fold) with s_chunks = chunks. */
int sgd_gn_bwd_rows_chunks(int32_t n, int32_t h, int32_t w, int32_t c);
int sgd_gn_bwd_reduce_rows(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t c_total, int32_t c_off,
                           const float* a, const float* b, int32_t silu, const float* gu, int32_t gu_ld, float drop_p,
                           uint32_t drop_seed, int32_t chunks, float* P, void* stream);
int sgd_gn_bwd_apply(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t c_total, int32_t c_off,
                     const float* a, const float* b, int32_t silu,
                     const float* gu, int32_t gu_ld, int32_t gu_mode, float drop_p, uint32_t drop_seed,
                     const float* A, const float* B, const float* Cc,
                     const float* gres, int32_t gres_ld, int32_t gres_mode,
                     float* dst, int32_t dst_ld, int32_t dst_off, int32_t accumulate, void* stream);
/* plain SiLU backward for the embedding MLPs: gx = g * SiLU'(x) (rows x c, contiguous) */
int sgd_silu_bwd(const float* x, const float* g, int64_t count, float* gx, void* stream);

/* LayerNorm backward (autograd of crossattetion_lr.py:36-43 / nn.LayerNorm): g = dL/dy for y = xhat*gamma + beta.
 * dst (+)= dL/dx (+ gres, the gradient of a residual branch around the norm, may be NULL);
 * gxhat[row, c] = g*xhat (column-sum it for dgamma; dbeta = column sum of g); gxhat may be NULL. */
int sgd_ln_bwd(const float* x, const float* g, const float* gamma, int32_t rows, int32_t c, float eps,
               float* dst, int32_t accumulate, float* gxhat, const float* gres, void* stream);
/* adjoint of nearest-upsample x2 (mode SGD_RS_UP2: g at 2x res) / avg-pool 2x2 (SGD_RS_AVGPOOL2: g at 1/2 res):
 * dst[n,h,w,c] (+)= resample^T(g)   (openaimodel_ca.py:128 Upsample, openaimodel.py:200 Downsample) */
int sgd_resample_bwd(const float* g, int32_t n, int32_t h, int32_t w, int32_t c, int32_t mode, float* dst,
                     int32_t accumulate, void* stream);

/* attention_ldm.CrossAttention / LinearCrossAttention (dynamic/attention_ldm.py:198-298), SURVEY row A23.
 * sgd_attention_masked: sgd_attention's exact-fp32 core with a per-key validity mask kmask[batch, tk] (bytes, 1 = attend):
 * a masked key gets weight 0, the result of sim.masked_fill(~mask, -FLT_MAX) before the softmax (:246-249); key 0 (the
 * null key the module prepends, always valid) must be unmasked.
 * sgd_linear_attention: out = (softmax over channels of q) * scale . (softmax over KEYS of k)^T v per (batch, head)
 * (:283-296), masked keys taking k = -FLT_MAX, v = 0; out has q's head layout (head stride q_hs, row stride out_ld). */
int sgd_attention_masked(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v, int32_t kv_ld,
                         int32_t kv_hs, const uint8_t* kmask, int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d,
                         float scale, float* out, int32_t out_ld, float* lse, void* stream);
int sgd_linear_attention(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v, int32_t kv_ld,
                         int32_t kv_hs, const uint8_t* kmask, int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d,
                         float scale, float* out, int32_t out_ld, void* stream);
/* legacy QKV attention backward (autograd of openaimodel.py:403-420), same addressing as sgd_attention:
 * dq/dk/dv are written with the same row strides / head strides as q/k/v (i.e. into a gqkv tensor).
 * Multi-query (kv_hs == 0, crossattetion_lr.py:115-137): dk/dv are summed over the heads inside the kernel.
 * Head dims 16 / 32 / 64 / 128 (config/dynamic/unet_fast_s64.yaml: 1024 channels / 8 heads). */
int sgd_attention_bwd(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v, int32_t kv_ld,
                      int32_t kv_hs, const float* o /* forward output */, int32_t o_ld,
                      const float* dout, int32_t dout_ld, const float* lse /* from sgd_attention */,
                      float* dvec /* workspace [batch, heads, tq] */,
                      int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d, float scale,
                      float* dq, float* dk, float* dv, void* stream);
/* sgd_attention_bwd in split precision (the f16x3 engine): every operand as hi + lo f16 halves, three
 * v_mfma_f32_32x32x16_f16 products, fp32 accumulate; P carried x 2^14 and dS x a per-wave running power of two
 * (csrc/attention.hip).  Same arguments, same outputs within the engine's tolerance; head dims 16 / 32 / 64. */
int sgd_attention_bwd_split(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v, int32_t kv_ld,
                      int32_t kv_hs, const float* o /* forward output */, int32_t o_ld,
                      const float* dout, int32_t dout_ld, const float* lse /* from sgd_attention */,
                      float* dvec /* workspace [batch, heads, tq] */,
                      int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d, float scale,
                      float* dq, float* dk, float* dv, void* stream);

/* Backward of the two attention_ldm cores above (autograd of dynamic/attention_ldm.py:239-254 / :283-296; the reference trains these
 * classes through torch.autograd).  sgd_attention_masked_bwd: sgd_attention_bwd's exact-fp32 kernels with the forward's key mask --
 * a masked key had weight exactly 0, so dk / dv of its rows are 0 and nothing flows through it into dq; lse from
 * sgd_attention_masked; head dims 16 / 32 / 64 / 128; kv_hs == 0 only with heads == 1.
 * sgd_linear_attention_bwd: dq / dk / dv of sgd_linear_attention for dout [b, tq, *] (row stride dout_ld, head h at +h*q_hs like
 * out), written with q's / k's / v's own strides; masked keys receive 0; d <= 128; fixed summation order (no atomics). */
int sgd_attention_masked_bwd(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v, int32_t kv_ld,
                             int32_t kv_hs, const uint8_t* kmask, const float* o /* forward output */, int32_t o_ld,
                             const float* dout, int32_t dout_ld, const float* lse /* from sgd_attention_masked */,
                             float* dvec /* workspace [batch, heads, tq] */,
                             int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d, float scale,
                             float* dq, float* dk, float* dv, void* stream);
int sgd_linear_attention_bwd(const float* q, int32_t q_ld, int32_t q_hs, const float* k, const float* v, int32_t kv_ld,
                             int32_t kv_hs, const uint8_t* kmask, const float* dout, int32_t dout_ld,
                             int32_t batch, int32_t heads, int32_t tq, int32_t tk, int32_t d, float scale,
                             float* dq, float* dk, float* dv, void* stream);

/* q_sample + loss (diffusion/ddpm.py:54-86, ddpm_sampler.py:116-119):
 *   x_noisy = sa[t]*x0 + s1ma[t]*noise      (NCHW in, NCHW out; tables are the float32 schedule buffers) */
int sgd_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac, const float* sqrt_1mac,
                 int32_t b, int64_t chw, float* out, void* stream);
/* --------------------------------------------------------------------------------------
 * The UNet stem (openaimodel.py:560-566, openaimodel_ca.py:735-741: a 3x3 convolution of the 3- or 4-channel input,
 * stride 1, padding 1) as a plain fp32 kernel (ABI 14): x NHWC [n,h,w,cin], cin 3 or 4; w the PARAMETER itself,
 * [cout][cin][3][3]; y NHWC with leading dimension y_ld; cout % 4 == 0, <= 1024.  stats (or NULL): the GroupNorm partial
 * statistics of y in sgd_igemm_args.stats' layout, [n][parts][2][cout] with parts = sgd_conv3_narrow_in_parts(h, w).
 * adjoint != 0: the INPUT GRADIENT of a 3x3 conv with 3 / 4 output channels (the head, openaimodel.py:724-728): x is the
 * output gradient [n,h,w,cin], w that conv's parameter [cin][cout][3][3] (taps flipped inside), y its input gradient. */
int sgd_conv3_narrow_in_parts(int32_t h, int32_t w);
int sgd_conv3_narrow_in(const float* x, const float* w, const float* bias, float* y, float* stats, int32_t n, int32_t h,
                        int32_t wd, int32_t cin, int32_t cout, int32_t y_ld, int32_t adjoint, void* stream);
/* The output head (openaimodel.py:830-835): y[n, h, w, :cout] = conv3x3(act(x)) + bias with cout = 3 or 4 and
 * act(x) = SiLU?(x * pa[n, c] + pb[n, c]) (the GroupNorm coefficients of sgd_gn_coef*; pa = pb = NULL: no prologue).
 * Plain fp32 FMAs (every arithmetic mode), cin % 32 == 0.  w9 is the conv's weight as [tap 0..8][cout][cin] -- a
 * transposed copy of the OIHW parameter the caller keeps in step with it (the weights reach the FMAs as scalar operands). */
int sgd_conv3_narrow_out(const float* x, const float* pa, const float* pb, int32_t silu, const float* w9, const float* bias,
                         float* y, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t y_ld, void* stream);

/* per_sample[b] = mean_chw (noise - eps)^2 ; geps_nhwc = d(mean_b per_sample)/d eps laid out NHWC for the backward
 * program (eps_nhwc is the UNet output in NHWC, noise NCHW) */
int sgd_mse_loss(const float* eps_nhwc, const float* noise_nchw, int32_t b, int32_t c, int32_t hw,
                 float* per_sample, float* geps_nhwc, void* stream);

/* --------------------------------------------------------------------------------------
 * Optimizer step: AdamW (lightning_module_common.py:20-42: torch.optim.AdamW defaults, decoupled weight decay)
 * and the LitEma shadow update (dynamic/ema.py:25-44) of every parameter in ONE launch:
 *   p *= 1 - lr*wd;  m += (1-b1)(g-m)  [torch's lerp_];  v = b2*v + (1-b2) g*g;
 *   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps);      shadow -= ema_omd * (shadow - p)
 * table: DEVICE array of `count` entries; g == NULL: the parameter got no gradient this step (AdamW skips it as
 * torch does, its shadow still follows); m/v/ema may be NULL only together with g / when ema_omd < 0 (no EMA).
 * chunk_start: DEVICE int32[count + 1], prefix sum of ceil(n / 4096) (block b serves the tensor that owns chunk b). */
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    float* ema;
    int64_t n;
} sgd_opt_tensor;
int sgd_adamw_ema_step(const sgd_opt_tensor* table, const int32_t* chunk_start, int32_t count, int32_t total_chunks,
                       float lr, float one_minus_beta1, float beta2, float one_minus_beta2 /* host doubles, rounded
                       once: 1.0f - 0.999f is off by 1.3e-5 relative */, float eps, float weight_decay,
                       float bias_correction1, float bias_correction2, float ema_one_minus_decay,
                       const float* skip_if_nonzero /* DEVICE float or NULL (ABI 18): when *skip_if_nonzero != 0 the launch
                       writes NOTHING -- the health flag of the step's gradients (a balanced-tail time-out of the backward
                       program, summed over the ranks of a data-parallel job: sgdm_amd/train.py) gates the optimizer, so a
                       poisoned step never reaches parameters, moments or EMA shadows */, void* stream);

/* --------------------------------------------------------------------------------------
 * Gradient exchange (ABI 18; SURVEY 8(b), last row).  Replaces the per-bucket all-reduce torch DDP issues under
 * pl.trainer.strategy=ddp (config/pl/default.yaml:2, README.md:84-94) for hosts that do not go through torch.distributed:
 * bucket[0..count) = SUM over the ranks of `nccl_comm` (an ncclComm_t the caller created), in place, fp32, enqueued on
 * `stream` (the caller's side stream: record an event behind the launches that fill the bucket, make the side stream wait
 * for it, call this, record the completion event -- what sgdm_amd/ddp.py does through ProcessGroupNCCL).  The 1 / world of
 * the average is folded into the backward program's un-scaling (sgd_wgrad_reduce*'s `scale`), so SUM is the whole collective.
 * librccl is resolved at first use, not linked: sgd_exchange_bind(path) names the instance whose communicators will be
 * passed (a PyTorch process must name torch/lib/librccl.so: communicators are only valid inside the copy that made them);
 * NULL / never called: an instance already loaded in the process, else the system's librccl.so.1.
 * Returns 0, 1 (invalid argument) or 2 (no RCCL library / the collective failed; the reason is written to stderr). */
int sgd_exchange_bind(const char* librccl_path);
int sgd_allreduce_bucket(void* nccl_comm, float* bucket, int64_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGDM_HIP_H */
