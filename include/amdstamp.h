/*
 * amdstamp.h -- C ABI of libamdstamp.so: the MI355X (gfx950 / CDNA4) hot path behind
 * KatherLab/STAMP's Extractor / Encoder / MIL-model seams.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host or the comment says "host";
 *   - every function returns 0 on success and a negative amds_status on failure; the message for the
 *     calling thread's last failure is amds_last_error(). Nothing here calls abort()/exit(): the
 *     reference wraps each slide in try/except and skips it on error
 *     (reference src/stamp/preprocessing/__init__.py:290-336), so failures must come back as values;
 *   - no hidden allocation and no device synchronisation: the caller owns all memory, passes a
 *     workspace (size from the matching *_workspace_bytes) and a hipStream_t (as void*);
 *   - row-major contiguous tensors; leading dimensions passed where views are allowed;
 *   - "act dtype": AMDS_F16 (default; fp16 operands, fp32 accumulate) or AMDS_BF16.
 *
 * Each entry point cites the reference interface (file:line under /root/reference) it stands behind.
 */
#ifndef AMDSTAMP_H
#define AMDSTAMP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMDS_VERSION_MAJOR 0
#define AMDS_VERSION_MINOR 2

typedef enum {
    AMDS_OK = 0,
    AMDS_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    AMDS_ERR_WORKSPACE = -2, /* workspace too small */
    AMDS_ERR_HIP = -3,       /* a HIP runtime call failed */
    AMDS_ERR_NODEVICE = -4,  /* no gfx950 device visible */
    AMDS_ERR_RANGE = -5      /* amds_check_finite: a result holds non-finite values (an intermediate left the 16-bit range) */
} amds_status;

typedef enum { AMDS_F16 = 0, AMDS_BF16 = 1, AMDS_F32 = 2 } amds_dtype;

/* GEMM epilogues (amds_gemm) */
typedef enum {
    AMDS_EPI_BIAS = 0,        /* out_act[m][n] = acc + bias[n]                                  */
    AMDS_EPI_BIAS_GELU = 1,   /* out_act = gelu_erf(acc + bias)          (nn.GELU, exact erf)   */
    AMDS_EPI_BIAS_RELU = 2,   /* out_act = max(acc + bias, 0)                                   */
    AMDS_EPI_RESIDUAL = 3,    /* x_f32[m][n] += scale[n] * (acc + bias[n])   (LayerScale + add) */
    AMDS_EPI_BIAS_F32 = 4,    /* out_f32[m][n] = acc + bias[n]                                  */
    AMDS_EPI_SWIGLU = 5,      /* packed fc1: out_act[m][j] = silu(g_j) * v_j, weights block-interleaved
                                 by amds_pack_swiglu_rows (timm SwiGLUPacked: fc1 -> chunk(2) -> silu(x1)*x2) */
    AMDS_EPI_PATCH = 6,       /* x_f32[(m/np)*T + P + m%np][n] = acc + bias[n] + pos[m%np][n] (patch embed) */
    AMDS_EPI_BIAS_GELU_F32 = 7,/* out_f32 = gelu_erf(acc + bias)                                 */
    AMDS_EPI_BIAS_RELU_F32 = 8 /* out_f32 = max(acc + bias, 0)                                   */
} amds_epilogue;

/* ------------------------------------------------------------------------------------------------
 * Library / context
 * ---------------------------------------------------------------------------------------------- */

/* Version as major*100+minor. */
int amds_version(void);
/* Message of the calling thread's most recent failing call ("" if none). */
const char* amds_last_error(void);
/* Fills name (<= n bytes) with the device's gcnArchName; AMDS_ERR_NODEVICE if none. */
int amds_device_info(int device, char* name_host, int n, int* cu_count_host, size_t* hbm_bytes_host);

/* Per-device context: owns what would otherwise be hidden process state -- the live profiler and the side stream
 * + fork/join events of the overlapped tile-encoder schedule.  One context per device and process: amds_create(device) returns the
 * device's context (creating it on first call, sharing it afterwards); amds_destroy drops one reference and, with the last one,
 * synchronises the device (the only synchronisation in the library) and frees events / stream.  Thread-safe: state is guarded by a
 * mutex per context.  The compute entry points stay context-free -- they take every buffer and the stream from the caller and
 * keep no state.  amds_create returns NULL on error (amds_last_error()). */
typedef struct amds_ctx amds_ctx;
amds_ctx* amds_create(int device);
void amds_destroy(amds_ctx* ctx);
int amds_ctx_device(const amds_ctx* ctx);

/* Live per-kernel timing for bench.py's roofline line: while enabled on the context of the calling thread's current device, every
 * launch made through this library is bracketed by a pair of HIP events recorded on the launch stream.  kind: 0 = MFMA GEMM
 * (work = 2*M*N*K flops), 1 = ViT attention (flops), 2 = LayerNorm (bytes), 3 = im2col (bytes),
 * 4 = fp32-MFMA GEMM (flops), 5 = fp8-MFMA GEMM (flops).  amds_profile_read waits for the recorded events and returns the summed
 * duration, launch count and summed work of one kind since the last reset (at most 32768 launches). */
int amds_profile_enable(amds_ctx* ctx, int on);
int amds_profile_reset(amds_ctx* ctx);
int amds_profile_read(amds_ctx* ctx, int kind, double* total_ms_host, long* launches_host, double* total_work_host);

/* ------------------------------------------------------------------------------------------------
 * Building blocks (also used by the MIL heads)
 * ---------------------------------------------------------------------------------------------- */

/* fp32 -> act dtype cast with optional zero padding: dst[r][c] = c < cols ? src[r*ld_src + c]*scale_r : 0
 * for r < rows, c < ld_dst. Used once, at weight-pack time. */
int amds_cast_pad(const float* src, int ld_src, void* dst, int ld_dst, int rows, int cols,
                  int dtype, void* stream);

/* LayerNorm over the last dim (torch.nn.LayerNorm semantics: biased variance, eps inside sqrt).
 * Replaces nn.LayerNorm calls on the path, e.g. reference
 * src/stamp/modeling/models/vision_tranformer.py:163,189,278 and timm Block.norm1/norm2.
 * x: fp32 rows at stride x_row_stride (elements); y: act dtype (or fp32 if out_dtype==AMDS_F32) at
 * stride y_row_stride. cols % 4 == 0, cols <= 8192. */
int amds_layernorm(const float* x, long x_row_stride, const float* gamma, const float* beta,
                   void* y, long y_row_stride, int rows, int cols, float eps, int out_dtype,
                   void* stream);

/* C = A[M,K] * W[N,K]^T with a fused epilogue (nn.Linear + activation + residual).
 * A, W: act dtype, K-contiguous, lda/ldw in elements (multiples of 8); K % 64 == 0; N % 128 == 0 or
 * N % 96 == 0 (pad weights with amds_cast_pad). M arbitrary. `out` is act dtype or fp32 depending on `epi`;
 * ldo in elements. bias/scale/pos are fp32 (scale may be NULL = 1; bias may be NULL = 0).
 * np/T/P only for AMDS_EPI_PATCH. acc_scale multiplies the accumulator before bias (1.0f normally).
 * Replaces nn.Linear on the path: timm Attention.qkv/proj, Mlp.fc1/fc2; reference
 * src/stamp/modeling/models/vision_tranformer.py:164-168,312-316. */
int amds_gemm(const void* A, long lda, const void* W, long ldw, int M, int N, int K,
              int dtype, int epi, void* out, long ldo, const float* bias, const float* scale,
              const float* pos, int np, int T, int P, float acc_scale, void* stream);

/* CLIP's activation in place on 16-bit rows: u = u * sigmoid(1.702 u) (HF transformers `quick_gelu`, the MLP of the PLIP extractor's vision tower,
 * reference src/stamp/preprocessing/extractor/plip.py:16-22 -> CLIPModel.get_image_features).  cols, ld multiples of 8. */
int amds_quick_gelu_inplace(void* u, long ld, long rows, int cols, int dtype, void* stream);

/* Weights-stationary GEMM for narrow layers (K = 96, 192 or 384; N % 32 == 0): out = epi([LayerNorm](A) W^T + bias).
 * The W slice of a workgroup lives in LDS for the whole launch and row groups stream through it from global memory
 * straight into MFMA operand registers.  If ln_gamma/ln_beta are given, A is the fp32 residual stream [M][lda] and
 * LayerNorm over its K columns (nn.LayerNorm, biased variance) is applied on the fly -- the `norm1`/`norm2` +
 * `qkv`/`fc1` pairs of a Swin block (reference ctranspath.py:659, 693-695) in one pass; otherwise A is act dtype.
 * epi: AMDS_EPI_BIAS, AMDS_EPI_BIAS_GELU (act dtype out), AMDS_EPI_RESIDUAL (fp32 out += ...), AMDS_EPI_BIAS_F32.
 * Supported (K, epi, LN) combinations are those the Swin stages need; others return AMDS_ERR_INVALID. */
int amds_gemm_rowstream(const void* A, long lda, const float* ln_gamma, const float* ln_beta, float ln_eps,
                        const void* W, long ldw, int M, int N, int K, int dtype, int epi, void* out, long ldo,
                        const float* bias, void* stream);

/* The whole MLP branch of a 96-channel Swin block in one pass over the fp32 residual stream x [M][96]:
 *   x += fc2(gelu(fc1(LayerNorm(x))))        (reference ctranspath.py:693-695 with _Mlp :355-383; hidden width 384)
 * fc1_w [384][96], fc2_w [96][384] act dtype (row-major, unpadded), biases and LayerNorm parameters fp32.  Both weight
 * matrices stay in LDS; the hidden activation only ever exists in MFMA registers. */
int amds_swin_mlp96(float* x, int M, const void* fc1_w, const float* fc1_b, const void* fc2_w, const float* fc2_b,
                    const float* ln_gamma, const float* ln_beta, float ln_eps, int dtype, void* stream);

/* Same branch for 192-channel blocks (hidden 768), weights streamed through LDS in 12 chunks from an image pre-packed in MFMA
 * fragment order: amds_swin_mlp192_pack(fc1_w [768][192], fc2_w [192][768], packed [768*192*2 elements]) once per weight set. */
int amds_swin_mlp192_pack(const void* fc1_w, const void* fc2_w, void* packed, int dtype, void* stream);
int amds_swin_mlp192(float* x, int M, const void* packed_w, const float* fc1_b, const float* fc2_b, const float* ln_gamma,
                     const float* ln_beta, float ln_eps, int dtype, void* stream);

/* Tuning hook: same as amds_gemm with an explicit kernel (-1 = library default: 12 when N % 256 == 0 and the grid fills the chip
 * (PATCH epilogue: 8), else 0 / 1).  0 = 128x128 tile, 1 = 128x96 tile, 8 = 256x256x64 eight-wave staggered two-group
 * pipeline, 10 = 256x256x64 four-wave kernel (128x128 wave tiles), 12 = the same on v_mfma 16x16x32 (13: its A/B schedule),
 * 3 / 7 = BK = 32 predecessors, 9 = ping-pong experiment (two 256x128 workgroups per CU).  Ids other than 12 / 13 give
 * bit-identical results; 12 / 13 sum the bias first and 32 products per MFMA: they differ from the others in the last bits.
 * -2 = -1 plus: when M is not a multiple of 256 and dropping the last, partial row tile (<= 128 rows) saves a whole wave of workgroups on
 * kernel 12, those rows run through kernel 0 as a second launch (the MIL training step: M = bags x 1025).  Rows of that tile then differ
 * in the last bits from what kernel 12 would give, so paths that promise identical rows across batch compositions use -1. */
int amds_gemm_ex(int cfg, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                 int dtype, int epi, void* out, long ldo, const float* bias, const float* scale,
                 const float* pos, int np, int T, int P, float acc_scale, void* stream);

/* OPT-IN fp8 (OCP e4m3) GEMM on v_mfma_f32_16x16x128_f8f6f4 (gfx950: 5 PFLOP/s dense peak, twice the fp16 rate), fp32 accumulate -- the "fp8 MFMA
 * weights" of BASELINE.json configs[4].  The reference computes in fp32 (src/stamp/preprocessing/__init__.py:324-325): e4m3 has 3 mantissa bits,
 * so this is never the default and its accuracy delta is measured and stated (tests/test_gpu_fp8.py).
 *   out[m][n] = act( (sum_k A8[m][k] * W8[n][k]) * rowscale[m] * colscale[n] + bias[n] )
 * A8 [M][lda], W8 [N][ldw]: e4m3 bytes, K contiguous, K % 128 == 0, N % 256 == 0, pitches multiples of 16.  rowscale [M] (per-row activation
 * scale, amds_quantize_rows_e4m3), colscale [N] (per-output-channel weight scale; for RESIDUAL the caller multiplies LayerScale into it and
 * into bias), bias [N]: fp32, each may be NULL.  epi: AMDS_EPI_BIAS / AMDS_EPI_BIAS_GELU (out f16 [M][ldo]), AMDS_EPI_SWIGLU (packed fc1 with
 * the 32-row gate / value interleave of amds_pack_swiglu_rows: out f16 [M][N / 2] = silu(gate) * value) or AMDS_EPI_RESIDUAL (out fp32 [M][ldo],
 * out += ...).  Replaces the same nn.Linear calls as amds_gemm when the host opts in. */
int amds_gemm_fp8(const void* A8, long lda, const void* W8, long ldw, int M, int N, int K, int epi, void* out, long ldo, const float* bias,
                  const float* colscale, const float* rowscale, void* stream);
/* q[r][c] = e4m3(x[r][c] / scale[r]), scale[r] = max_c |x[r][c]| / 448 (1 for an all-zero row): per-row dynamic quantisation of an activation
 * (f16 or fp32 rows, cols % 4 == 0, <= 8192) or, applied to a weight matrix [N][K], its per-output-channel scales. */
int amds_quantize_rows_e4m3(const void* x, long ldx, void* q, long ldq, float* scale, int rows, int cols, int in_dtype, void* stream);
/* Producers of the fp8 path that spare a 16-bit round trip (all opt-in, like amds_gemm_fp8):
 *   amds_layernorm_quant_e4m3  nn.LayerNorm of fp32 rows fused with the row quantisation: q / scale as amds_quantize_rows_e4m3 of the normalised row,
 *                              rownorm[r] (optional) = its L2 norm
 *   amds_row_bound_scale       us[r] = (rownorm[r] * c0 + c1) / 448 -- with c0 = max_n ||w_n||, c1 = max_n |b_n| a scale that provably covers
 *                              |Linear(h_r)| (Cauchy-Schwarz) and |gelu(Linear(h_r))|: the OUTPUT row scale of ...
 *   amds_gemm_fp8_out8         ... amds_gemm_fp8 (BIAS / BIAS_GELU) writing e4m3 directly: out8[m][n] = e4m3(value / out_rowscale[m]); a bounding
 *                              scale instead of the row maximum costs fp8 nothing in relative precision (it is a floating-point format: only the
 *                              subnormal floor moves up), and no kernel has to re-read a 16-bit copy to find the maximum */
int amds_layernorm_quant_e4m3(const float* x, long ldx, const float* gamma, const float* beta, float eps, void* q, long ldq, float* scale,
                              float* rownorm, int rows, int cols, void* stream);
int amds_row_bound_scale(const float* rownorm, float c0, float c1, float* us, int rows, void* stream);
int amds_gemm_fp8_out8(const void* A8, long lda, const void* W8, long ldw, int M, int N, int K, int epi, void* out8, long ldo8,
                       const float* out_rowscale, const float* bias, const float* colscale, const float* rowscale, void* stream);

/* LayerNorm folded into the GEMMs around it (tile encoder, round 2).  timm's Block computes x += proj(attn(norm1(x))); x += fc2(act(fc1(
 * norm2(x)))) (reference extractors virchow2.py:29-30, uni2.py:32-43 -> timm VisionTransformer.forward).  With
 *   W'[n][k] = W[n][k] * gamma[k] (re-rounded to the act dtype), colsum[n] = sum_k W'[n][k], bias'[n] = bias[n] + sum_k W[n][k] * beta[k]:
 *   Linear(LayerNorm(x))[m][n] = rstd[m] * (sum_k x[m][k] W'[n][k] - mean[m] * colsum[n]) + bias'[n]
 * so the GEMM can read the UN-normalised rows and apply the row statistics in its epilogue, and the stand-alone LayerNorm (one
 * read of the fp32 residual stream + one write, 7 % of the ViT-L step) disappears:
 *   producer form (epi = AMDS_EPI_RESIDUAL; bias, if any, and scale as in amds_gemm; acc_scale 1): besides out += ..., writes
 *       xh      act dtype [M][ldo]: a 16-bit copy of the updated rows (the A operand of the next GEMM), and
 *       rowpart fp32 [M][N/128][2]: (sum, sum of squares) of the updated rows per 128-column slab (amds_ln_rowstat adds the slabs);
 *   consumer form (epi = BIAS / BIAS_GELU / SWIGLU; A = the 16-bit copy, W = W', bias = bias'):
 *       out = act(acc * rowstat[m][0] + (colsum[n] * rowstat[m][1] + bias[n])),   rowstat fp32 [M][2] = (rstd, -mean * rstd).
 * N % 256 == 0, K % 64 == 0 (the production kernel only).  Same arithmetic as LayerNorm-then-GEMM up to where the 16-bit rounding
 * happens (x instead of the normalised x; W * gamma instead of W): equal error against fp32 when |mean| << std over a row. */
int amds_gemm_lnfold(const void* A, long lda, const void* W, long ldw, int M, int N, int K, int dtype, int epi, void* out,
                     long ldo, const float* bias, const float* scale, void* xh, float* rowpart, const float* rowstat,
                     const float* colsum, void* stream);

/* The RESIDUAL producer of amds_gemm_lnfold with the residual stream held as TWO fp16 planes instead of fp32 rows + a 16-bit copy:
 *   x = hi + lo;   x += scale[n] * (A W^T + bias[n]);   hi = fp16(x);   lo = fp16(x - hi);   rowpart = partial (sum, sum of squares) of the fp32 x
 * in place on hi / lo [M][ld] (fp16; ld elements).  hi is exactly the 16-bit copy the consumer GEMMs read as their A operand, so a block's
 * proj / fc2 launches move 4 + 4 bytes per element instead of 4 + 4 + 2; x keeps ~22 mantissa bits (elements below 2^-13 |x|_row keep the
 * absolute resolution of fp16 subnormals, 6e-8).  fp16 operands only (a bf16 pair would hold 16 bits).  Tile encoder: the default on the
 * folded path (vit.hip); replaces the fp32 rows `x += ...` of `Block.forward` in timm's VisionTransformer behind
 * src/stamp/preprocessing/extractor/uni2.py:32-43, virchow2.py:29-30. */
int amds_gemm_lnfold_planes(const void* A, long lda, const void* W, long ldw, int M, int N, int K, void* hi, void* lo, long ld,
                            const float* bias, const float* scale, float* rowpart, void* stream);
/* x fp32 [M][ldx] -> hi = fp16(x), lo = fp16(x - hi) [M][ld] + (rstd, -mean * rstd) per row: amds_ln_stats_cast for the plane form. */
int amds_ln_stats_split(const float* x, long ldx, int M, int D, float eps, void* hi, void* lo, long ld, float* rowstat, void* stream);
/* out[i][:] = hi[i * row_stride][:] + lo[i * row_stride][:] in fp32, i < rows (row_stride 1: every row; T: the class rows of [B*T] tokens). */
int amds_planes_to_f32(const void* hi, const void* lo, long ld, long row_stride, float* out, long ldo, long rows, int D, void* stream);

/* rowpart [M][NP][2] (NP = N/128 of the producer) -> rowstat [M][2] = (rstd, -mean * rstd) with biased variance over D columns */
int amds_ln_rowstat(const float* rowpart, int M, int NP, int D, float eps, float* rowstat, void* stream);
/* Same, also counting into diag (device int[2], caller-zeroed, may be NULL): [0] rows whose sum of squares reaches the act dtype's
 * max^2 (fp16: 65504^2) -- since |x| <= sqrt(sum x^2), rows below that limit provably hold no element the 16-bit copy cannot represent, rows
 * at or above it may (real ViT-H / ViT-g checkpoints have massive-activation channels; a saturated copy shows up as non-finite features);
 * [1] rows with |mean| > 8 sigma, where rounding x instead of x - mean costs the folded form precision (amds_gemm_lnfold's stated
 * precondition is |mean| << std).  amds_vit_forward counts over every folded LayerNorm of the call into its workspace
 * (amds_vit_workspace_diag_offset). */
int amds_ln_rowstat_diag(const float* rowpart, int M, int NP, int D, float eps, float* rowstat, int* diag, int dtype, void* stream);
/* the first LayerNorm of a stack: x fp32 [M][D] (pitch ldx) -> xh act dtype [M][D] (pitch ldxh) + rowstat [M][2] */
int amds_ln_stats_cast(const float* x, long ldx, int M, int D, float eps, void* xh, long ldxh, float* rowstat, int dtype, void* stream);

/* Reorders the 2H rows of a SwiGLUPacked fc1 weight/bias so that gate/value columns of one hidden
 * unit land in the same MFMA lane: dst blocks of 32 rows alternate [gate 32j..][value 32j..].
 * H % 32 == 0. src/dst fp32 [2H][cols] (bias: cols = 1). */
int amds_pack_swiglu_rows(const float* src, float* dst, int H, int cols, void* stream);

/* Multi-head self-attention over packed qkv (timm Attention.forward / F.scaled_dot_product_attention;
 * reference nn.MultiheadAttention at src/stamp/modeling/models/vision_tranformer.py:191,217-227).
 * qkv: act dtype [B*T][3*H*64] with q|k|v thirds (each [H][64]); out: act dtype [B*T][H*64].
 * head_dim is fixed at 64; T <= 288 (whole K/V of a head staged in LDS). softmax(q k^T / 8) v. */
int amds_attention_vit(const void* qkv, void* out, int B, int T, int H, int dtype, void* stream);

/* Same with an explicit head_dim: 64 (above) or 80 (ViT-H/14, e.g. Virchow2: dim 1280, 16 heads); qkv thirds are
 * [H][head_dim]. */
int amds_attention_vit_hd(const void* qkv, void* out, int B, int T, int H, int head_dim, int dtype, void* stream);

/* Same contract for ANY T (K/V streamed through LDS in 64-key tiles, online softmax; the T x T matrix is never
 * materialised).  Used by the MIL heads: bags of 1024 tiles in training, whole slides (tens of thousands of
 * tiles) at deploy time (reference src/stamp/modeling/models/vision_tranformer.py:191, 217-227, mask=None path
 * of src/stamp/modeling/models/__init__.py:286-313).  The K / V tiles travel by buffer-form LDS-DMA: one bag's q | k | v
 * rows must stay below the 2 GB a buffer descriptor addresses (T * 3 * H * 128 bytes < 2^31: T < 699 050 at 8 heads) --
 * AMDS_ERR_INVALID otherwise; the same holds for every amds_attention_* entry of the streaming kernels (forward, masked,
 * ALiBi, training forward / backward). */
int amds_attention(const void* qkv, void* out, int B, int T, int H, int dtype, void* stream);
/* ONE query row per (bag, head) against all T keys / values of the bag as stored in the packed qkv tensor [B*T][3*H*64] (16-bit): out[b][h*64..] = softmax(q[b][h*64..] K_b,h^T / 8) V_b,h,
 * fp32 arithmetic, 16-bit q [B][ldq] and out [B][ldo].  The class token's attention in the LAST block of the MIL `vit` head, whose other rows nothing reads (reference
 * src/stamp/modeling/models/vision_tranformer.py: the head takes `x[:, 0]` behind the last block).  T <= 32768. */
int amds_attention_row(const void* q, long ldq, const void* qkv, void* out, long ldo, int B, int T, int H, int dtype, void* stream);
/* The same for query row `qrow` of every bag IN PLACE of the token-major tensors, in the training forward's form and its backward -- for a block of which only
 * that row is read afterwards (the last block of the MIL `vit` head: `self.mlp_head(x[:, 0])`, reference src/stamp/modeling/models/vision_tranformer.py):
 *   fwd: out[(b T + qrow)][h*64..] = (M o softmax(q k^T / 8)) v with dropout p on the probabilities (the bits amds_attention_fwd_train draws for that row),
 *        lse[(b H + h) T + qrow] as amds_attention_fwd_lse; the other rows of out and lse are not touched.
 *   bwd: the WHOLE dqkv [B*T][3*H*64] of a block whose other query rows carry no gradient: dK, dV of every token (rank-1 in the one query), dQ zero but for
 *        row qrow.  out / dout: the [B*T][H*64] tensors (only row qrow of each bag is read). */
int amds_attention_row_fwd_train(const void* qkv, void* out, float* lse, int B, int T, int H, int qrow, int dtype, float p, uint64_t seed,
                                 uint32_t stream_id, void* stream);
/* The same one-query pair for the ALiBi attention (reference vision_tranformer.py:42-74, mask = None so that the class row carries the distance term too):
 * out = sum_k (p_k - bias_scale_h |c_q - c_k| / running_mean_h) v_k at row (bag, qrow) of out / u / osm [B*T][H*64] (u = the distance-weighted value sum, osm = the
 * softmax part alone: what the backward needs); backward: dqkv [B*T][3*H*64] complete (zero dQ off the query row), dbs [B][H] = -dO . U (sum over bags = d bias_scale_h). */
int amds_attention_row_alibi_fwd_train(const void* qkv, const float* coords, const float* inv_running_mean, const float* bias_scale, void* out, void* u, void* osm,
                                       float* lse, int B, int T, int H, int qrow, int dtype, void* stream);
int amds_attention_row_alibi_bwd_train(const void* qkv, const void* osm, const void* u, const void* dout, const float* lse, const float* coords, const float* bias_scale,
                                       const float* inv_running_mean, void* dqkv, float* dbs, int B, int T, int H, int qrow, int dtype, void* stream);
int amds_attention_row_bwd_train(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int T, int H, int qrow,
                                 int dtype, float p, uint64_t seed, uint32_t stream_id, void* stream);
/* Process-wide switch (default 1; AMDS_MIL_CLS_TAIL=0 in the environment starts at 0): the MIL `vit` head's last block computed for the class rows alone where the
 * block's other rows are dead (amds_mil_vit_forward without ALiBi / mask; the attention of amds_mil_vit_train_forward / _backward without ALiBi). */
int amds_set_mil_cls_tail(amds_ctx* ctx, int on);
int amds_get_mil_cls_tail(amds_ctx* ctx);

/* ALiBi variant of the reference's MultiHeadALiBi (src/stamp/modeling/models/vision_tranformer.py:42-74, eval mode):
 *   out = softmax(q k^T / 8) v  -  head_scale[h] * cdist(coords_q, coords_k) v        (bias applied AFTER the softmax)
 * coords: fp32 [B][T][2] (class token at (0,0), :349-351); head_scale[h] = bias_scale_h / running_mean_h, fp32 [H].
 * `out` is ALWAYS bf16 (the un-normalised distance term can exceed the fp16 range on long bags). */
int amds_attention_alibi(const void* qkv, const float* coords, const float* head_scale, void* out, int B, int T, int H,
                         int dtype, void* stream);

/* The reference's `mask != None` forward (src/stamp/modeling/models/vision_tranformer.py:355-381; pinned by the reference's
 * tests/test_model.py:28-32).  pad: u8 [B][T], 1 = padded tile, class token included at t = 0 (never padded).
 * blocked(q, k) = (pad[q] & pad[k]) | (q > 0 & k == 0)  -- the outer-product mask of :363-367, restated literally.
 * amds_attention_masked (nn.MultiheadAttention branch): blocked scores are -inf before the softmax; the reference passes
 *   attn_mask.repeat(heads, 1, 1) (:224), so (bag b, head h) uses the pad row of bag (b*mask_heads + h) % B -- reproduced;
 *   mask_heads = the model's real head count (<= H: heads h >= mask_heads are zero padding whose output is zero under any mask).
 * amds_attention_alibi_masked (MultiHeadALiBi branch, :62-70): softmax over all keys, blocked products zeroed afterwards, no
 *   distance term on the class-token row and column (:370-372); out bf16. */
int amds_attention_masked(const void* qkv, const uint8_t* pad, void* out, int B, int T, int H, int mask_heads, int dtype, void* stream);
int amds_attention_alibi_masked(const void* qkv, const float* coords, const float* head_scale, const uint8_t* pad, void* out,
                                int B, int T, int H, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tile encoder (ViT) -- the model behind Extractor.model (reference
 * src/stamp/preprocessing/extractor/__init__.py:17-28; called at src/stamp/preprocessing/__init__.py:324-325)
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    int img;          /* 224 */
    int patch;        /* 14 or 16 */
    int dim;          /* 1024 / 1280 / 1536 */
    int depth;        /* 24 / 32 */
    int heads;        /* dim / 64 or dim / 80 */
    int hidden;       /* MLP hidden width (input width of fc2) */
    int n_prefix;     /* cls + register tokens */
    int mlp_kind;     /* 0 = Linear-GELU-Linear, 1 = SwiGLUPacked (fc1 out = 2*hidden), 2 = Linear-quick_gelu-Linear (HF CLIP: x sigmoid(1.702 x); plain packing only) */
    int layerscale;   /* 0/1 */
    int dtype;        /* AMDS_F16 / AMDS_BF16 */
    float ln_eps;     /* 1e-6 for timm ViTs */
} amds_vit_cfg;

typedef struct {
    const float* ln1_w; const float* ln1_b;
    const void*  qkv_w; const float* qkv_b;     /* [3*dim][dim] act dtype, [3*dim] */
    const void*  proj_w; const float* proj_b;   /* [dim][dim] */
    const float* ls1;                           /* [dim] or NULL */
    const float* ln2_w; const float* ln2_b;
    const void*  fc1_w; const float* fc1_b;     /* GELU: [hidden][dim]; SwiGLU: [2*hidden][dim] block-interleaved */
    const void*  fc2_w; const float* fc2_b;     /* [dim][hidden_pad] */
    const float* ls2;
    /* LayerNorm folded into the GEMMs (amds_gemm_lnfold), set in all blocks or in none.  When set, qkv_w / fc1_w hold W * gamma
     * (ln1 / ln2) rounded to the act dtype, qkv_b / fc1_b hold b + W beta, these hold sum_k of the rounded W * gamma per output row
     * (same row order as the weight, i.e. block-interleaved for SwiGLU), and ln1_* / ln2_* are not read. */
    const float* qkv_colsum;                    /* [3*dim] or NULL */
    const float* fc1_colsum;                    /* [hidden] / [2*hidden] or NULL */
} amds_vit_block;

/* Exact class-token rows (optional; amds_vit_weights.exact_host).  What the reference stores per tile is ONE row of the output,
 * model(tiles)[:, 0].half() (src/stamp/preprocessing/__init__.py:324-325): when this array is given, that row's own chain -- query row,
 * attention output, proj / fc1 / fc2 rows of every block -- is ALSO computed in exact fp32 (fp32 MFMA, the original un-folded weights,
 * keys / values of all tokens as the 16-bit path stored them) on a separate fp32 class stream, and written over the class rows of the
 * main path after every sub-layer.  fp32 device pointers; LayerNorm parameters are the block's ln1_* / ln2_*. */
typedef struct {
    const float* q_w;    const float* q_b;     /* [dim][dim], [dim]: the q third of attn.qkv                          */
    const float* proj_w; const float* proj_b;  /* [dim][dim], [dim]: attn.proj, rows multiplied by ls1 when LayerScale */
    const float* fc1_w;  const float* fc1_b;   /* [fc1_out][dim] in timm's order, unpadded (SwiGLU: gate rows, then value rows) */
    const float* fc2_w;  const float* fc2_b;   /* [dim][exact_hidden], [dim]: mlp.fc2, rows multiplied by ls2          */
} amds_vit_exact_block;

/* OPT-IN fp8 GEMMs of the blocks (amds_gemm_fp8; GELU-MLP models; never the default -- see its comment).  When amds_vit_weights.fp8_host
 * is set, the blocks' weights must be the PLAIN packing (no LayerNorm fold: ln1_* / ln2_* and the biases are read), and every Linear of a
 * block runs as  LayerNorm / attention / GELU output (f16) -> amds_quantize_rows_e4m3 (per-row scale) -> amds_gemm_fp8  with these e4m3 weights
 * and fp32 per-output-channel vectors (cs = weight scale, for proj / fc2 times LayerScale; *_b = bias times LayerScale). */
typedef struct {
    const void* qkv_w8;  const float* qkv_cs;                        /* [3*dim][dim] e4m3, [3*dim] */
    const void* proj_w8; const float* proj_cs; const float* proj_b;  /* [dim][dim], [dim], [dim] */
    const void* fc1_w8;  const float* fc1_cs;                        /* [hidden][dim], [hidden] */
    float fc1_wnorm_max; float fc1_babs_max;                         /* max_n ||fc1.weight[n]||_2 and max_n |fc1.bias[n]|: the bound behind amds_row_bound_scale */
    const void* fc2_w8;  const float* fc2_cs;  const float* fc2_b;   /* [dim][hidden], [dim], [dim] */
} amds_vit_fp8_block;

typedef struct {
    const void*  patch_w;    /* [dim][kp] act dtype; conv weight flattened (c,i,j), divided by std[c]; kp = roundup(3*p*p, 64) */
    const float* patch_b;    /* [dim] conv bias - sum_k W[k]*mean[c]/std[c] */
    const float* prefix;     /* [n_prefix][dim] fp32: cls/reg tokens (+ their pos-embed rows if any) */
    const float* pos_patch;  /* [n_patches][dim] fp32 */
    const amds_vit_block* blocks_host; /* HOST array of `depth` structs holding device pointers */
    const float* norm_w; const float* norm_b;
    /* Compensated patch embedding (0 = off).  The tile transform is folded into the patch weight, which then multiplies the RAW 0..255
     * values: rounding W/std to 16 bits costs (rms of u8 / rms of the normalised pixel) ~ 2x the relative error an ordinary operand
     * rounding does, on the tensor every block builds on (measured: the largest single contribution to the feature error, DESIGN.md
     * section 5).  With patch_lo_shift = s > 0, patch_w is [dim][2*kp] = [hi | (W/std - hi) * 2^s] (both act dtype; s = 11 for fp16, 8 for
     * bf16 keeps the low part in the normal range) and the patch matrix carries the pixel values twice, the second copy scaled by 2^-s
     * (exact): the fp32 accumulator sees W/std to ~22 bits.  0.2 % of the path's flops become 0.4 %. */
    int patch_lo_shift;
    const amds_vit_exact_block* exact_host;   /* HOST array of `depth` structs, or NULL (off) */
    int exact_hidden;                         /* the MLP's real (unpadded) hidden width, e.g. 3416 for Virchow2 (cfg.hidden is padded) */
    const amds_vit_fp8_block* fp8_host;       /* HOST array of `depth` structs, or NULL (off) */
    const float* pre_norm_w;                  /* [dim] LayerNorm applied to the embedded tokens before the first block (HF CLIP `pre_layrnorm`, timm */
    const float* pre_norm_b;                  /* `norm_pre`), or NULL: none (every timm preset).  Not filled by amds_vit_pack: the host sets them. */
    /* Class-row tail of the LAST block (NULL = off; AMDS_PACK_CLS_TAIL fills it).  What the reference keeps of the trunk's output is the class
     * row, `model(tiles)[:, 0]` (src/stamp/preprocessing/__init__.py:324-325, extractor/virchow2.py:29-30), and in the last block only the keys /
     * values of the other tokens reach that row.  When this struct is given and no token tensor is requested (amds_vit_forward, tokens_f32 ==
     * NULL of amds_vit_forward_tokens), the last block computes the k | v columns of qkv for all tokens and then ONLY the class row's chain
     * -- query, attention, proj, MLP -- in fp32 with these un-folded weights (the kernels of the exact class-token path), and the final
     * LayerNorm reads that row: 3.5 % of ViT-L/14's products (5.64 of 162.02 GFLOP per tile) are never computed because nothing reads them.
     * The stored features are those of the full last block up to the 16-bit path's rounding (this row is carried in fp32 instead).
     * Requires mlp_kind 0 / 1; ignored when a token tensor is requested. */
    const amds_vit_exact_block* cls_tail;     /* HOST pointer to ONE struct (device pointers inside): the last block's fp32 rows */
} amds_vit_weights;

/* ---- weight packing in the library (so that a non-Python host can use the tile encoder through this ABI alone) -------------------------
 * A timm VisionTransformer checkpoint as HOST fp32 tensors, row-major exactly as the state_dict stores them -- what the reference's
 * extractor factories hand to load_state_dict (src/stamp/preprocessing/extractor/uni2.py:32-43, virchow2.py:34-45, h_optimus_0.py:15-30,
 * reddino.py:40-57): keys blocks.{l}.norm1 / attn.qkv / attn.proj / ls1.gamma / norm2 / mlp.fc1 / mlp.fc2 / ls2.gamma. */
typedef struct {
    const float* norm1_w; const float* norm1_b;   /* [dim] */
    const float* qkv_w;   const float* qkv_b;     /* [3*dim][dim], [3*dim] (q | k | v thirds) */
    const float* proj_w;  const float* proj_b;    /* [dim][dim], [dim] */
    const float* ls1;                             /* [dim] or NULL (no LayerScale) */
    const float* norm2_w; const float* norm2_b;
    const float* fc1_w;   const float* fc1_b;     /* GELU: [hidden][dim]; SwiGLUPacked: [2*hidden][dim], gate rows then value rows */
    const float* fc2_w;   const float* fc2_b;     /* [dim][hidden], [dim] */
    const float* ls2;
} amds_vit_host_block;
typedef struct {
    const float* patch_w;    /* patch_embed.proj.weight [dim][3][patch][patch] */
    const float* patch_b;    /* [dim] */
    const float* cls_token;  /* [dim] */
    const float* reg_token;  /* [n_prefix-1][dim] or NULL */
    const float* pos_embed;  /* [n_patches + (no_embed_class ? 0 : n_prefix)][dim] */
    const amds_vit_host_block* blocks;   /* depth structs */
    const float* norm_w; const float* norm_b;
    int hidden;              /* the MLP's real hidden width (cfg.hidden is its 64-padded value) */
    int no_embed_class;      /* timm's flag: 1 = position rows cover the patches only */
    double mean[3]; double std[3];       /* the extractor's Normalize (folded into the patch embedding, in double) */
} amds_vit_host_weights;
#define AMDS_PACK_LNFOLD      1   /* LayerNorm folded into qkv / fc1 (amds_gemm_lnfold): W * gamma, b + W beta, row sums */
#define AMDS_PACK_PATCH_SPLIT 2   /* patch weight as a 16-bit [hi | lo] pair (patch_lo_shift) */
#define AMDS_PACK_EXACT       4   /* also the fp32 rows of the exact class-token path (amds_vit_exact_block) */
#define AMDS_PACK_CLS_TAIL    8   /* also the fp32 rows of the LAST block (amds_vit_weights.cls_tail = &out_exact[depth - 1]; out_exact [depth] required) */
/* Bytes of the packed image for these flags (0 on error: amds_last_error()). */
size_t amds_vit_pack_bytes(const amds_vit_cfg* cfg_host, const amds_vit_host_weights* src_host, int flags);
/* Packs into the DEVICE buffer dev_image (>= amds_vit_pack_bytes, 256-byte aligned) and fills the caller's host structs -- out_w, out_blocks
 * [depth], out_exact [depth] (may be NULL without AMDS_PACK_EXACT) -- with pointers into it; they stay valid while dev_image lives and are what
 * amds_vit_forward takes.  Arithmetic on the host in double (threads over blocks; AMDS_PACK_THREADS overrides the count), one upload; the
 * call waits for the upload (a one-time call: the exception to "no synchronisation"). */
int amds_vit_pack(const amds_vit_cfg* cfg_host, const amds_vit_host_weights* src_host, int flags, void* dev_image, size_t bytes,
                  amds_vit_weights* out_w_host, amds_vit_block* out_blocks_host, amds_vit_exact_block* out_exact_host, void* stream);
/* The same image into HOST memory (for hosts that manage their own uploads, and for tests without a GPU): pointers in the out structs are
 * target_base + offset (target_base NULL: image_host itself). */
int amds_vit_pack_host(const amds_vit_cfg* cfg_host, const amds_vit_host_weights* src_host, int flags, void* image_host, size_t bytes,
                       const void* target_base, amds_vit_weights* out_w_host, amds_vit_block* out_blocks_host,
                       amds_vit_exact_block* out_exact_host);

/* Workspace bytes for a forward over at most `batch` tiles per internal chunk. */
size_t amds_vit_workspace_bytes(const amds_vit_cfg* cfg_host, int batch);

/* Byte offset, inside a workspace sized by amds_vit_workspace_bytes(cfg, batch), of int32[2] range diagnostics of the LayerNorm-folded path
 * (see amds_ln_rowstat_diag): the forward ADDS to them; zero them when you want a fresh count, read them whenever the stream has drained.
 * (0 on error.) */
size_t amds_vit_workspace_diag_offset(const amds_vit_cfg* cfg_host, int batch);

/* tiles: u8 [B][img][img][3] (HWC, as decoded) -> feats: fp16 [B][dim] = CLS token of the final
 * LayerNorm, i.e. model(tiles)[:, 0].half() (reference src/stamp/preprocessing/__init__.py:324-325,
 * src/stamp/preprocessing/extractor/virchow2.py:29-30). The u8 -> (x/255-mean)/std transform
 * (reference h_optimus_0.py:22-30 etc.) is folded into patch_w/patch_b.
 * Processes the batch in chunks of `chunk` tiles (workspace sized for `chunk`).
 * When the host has created a context for the calling thread's current device (amds_create), a chunk whose token rows do not fill its
 * last 256-row GEMM tile (e.g. the reference's DataLoader batch: 64 x 257 rows) runs its last few tiles as an independent chain of
 * launches on that context's side stream -- tiles are independent through the whole network -- and `stream` waits for that chain before
 * the final norm: same kernels, same results bit for bit, no extra wave of workgroups per GEMM.  AMDS_VIT_TAIL=0 disables it. */
int amds_vit_forward(const amds_vit_cfg* cfg_host, const amds_vit_weights* w_host,
                     const uint8_t* tiles, void* feats_f16, int B, int chunk,
                     void* ws, size_t ws_bytes, void* stream);

/* The guard a host runs over features BEFORE it persists them (the reference writes `model(tiles)[:, 0].half()` straight into the .h5,
 * src/stamp/preprocessing/__init__.py:324-345; its per-slide try/except :328-336 is the only place an error can surface).  Counts the
 * non-finite values among x[0..n) (dtype 0 = f16, 1 = bf16, 2 = f32) into *count_dev (device int, zeroed by this call), copies the count to
 * *count_host and SYNCHRONISES `stream` (the one call of the tile-encoder API that does).  AMDS_OK if it is 0, AMDS_ERR_RANGE otherwise:
 * the 16-bit activation formats of the fast path hold |x| <= 65504 (f16); a checkpoint with larger intermediates (massive-activation
 * channels) turns the affected tiles' features into NaN -- every overflow propagates to the class row through the attention -- so a zero
 * count is also the proof that no intermediate of the call overflowed.  On AMDS_ERR_RANGE re-run the tiles with a packing that keeps
 * fp32 residual rows (amds_vit_pack without AMDS_PACK_LNFOLD) or bf16 activations; `stamp_amd.vit.HipViT` does so by itself. */
int amds_check_finite(const void* x, long n, int dtype, int* count_dev, int* count_host, void* stream);

/* Same, but also returns the final-LayerNorm'd tokens (fp32 [B][T][dim]) for parity checks and for
 * CLS (+) mean-patch extractors (reference src/stamp/preprocessing/extractor/virchow_full.py:30-35).
 * tokens_f32 may be NULL. */
int amds_vit_forward_tokens(const amds_vit_cfg* cfg_host, const amds_vit_weights* w_host,
                            const uint8_t* tiles, void* feats_f16, float* tokens_f32, int B, int chunk,
                            void* ws, size_t ws_bytes, void* stream);

/* Same result as amds_vit_forward, with consecutive chunks alternating between `stream` and one library-owned side
 * stream so that two chunks are in flight (ws must hold 2 x amds_vit_workspace_bytes(chunk)); `stream` waits for the
 * side stream before the call's work is considered complete. */
int amds_vit_forward_overlapped(amds_ctx* ctx, const amds_vit_cfg* cfg_host, const amds_vit_weights* w_host, const uint8_t* tiles,
                                void* feats_f16, int B, int chunk, void* ws, size_t ws_bytes, void* stream);

/* Building blocks of the exact class-token path (exposed for tests; amds_vit_forward calls them when exact_host is set).
 *   amds_attention_cls_f32   out[b][h*hd..] = softmax(q[b][h*hd..] . K_b,h^T / sqrt(hd)) V_b,h in fp32: ONE fp32 query row per (tile, head)
 *                            against the keys / values of all T <= 288 tokens of that tile as stored in the packed act-dtype qkv tensor
 *   amds_vit_cls_gather      xc[b][:] = x[b*T][:]
 *   amds_vit_cls_scatter     x[b*T][:] = xc[b][:]; if xh / rowstat: 16-bit copy of that row and its (rstd, -mean*rstd) (amds_ln_stats_cast's form)
 *   amds_mlp_act_f32         kind 0: u = gelu_erf(u) over `hidden` columns; kind 1 (SwiGLUPacked): u[:, j] = silu(u[:, j]) * u[:, hidden + j];
 *                            kind 2: u = silu(u) */
int amds_attention_cls_f32(const float* q, long ldq, const void* qkv, float* out, long ldo, int B, int T, int H, int head_dim,
                           int dtype, void* stream);
int amds_vit_cls_gather(const float* x, float* xc, int B, int T, int D, void* stream);
int amds_vit_cls_scatter(const float* xc, float* x, void* xh, float* rowstat, int B, int T, int D, float eps, int dtype, void* stream);
int amds_mlp_act_f32(float* u, long ld, int rows, int hidden, int kind, void* stream);

/* u8 HWC tiles -> im2col patch matrix [B*np][kp] (act dtype, raw 0..255 values, zero padded).
 * Exposed for tests; amds_vit_forward calls it internally. */
int amds_tile_im2col_u8(const uint8_t* tiles, void* out, int B, int img, int patch, int kp,
                        int dtype, void* stream);
/* lo_shift > 0: rows are 2*kp wide, columns kp.. repeat the values scaled by 2^-lo_shift (exact) -- the A operand for a patch weight
 * split into [hi | (W - hi) * 2^lo_shift] (amds_vit_weights.patch_lo_shift); lo_shift = 0 is amds_tile_im2col_u8. */
int amds_tile_im2col_u8_ex(const uint8_t* tiles, void* out, int B, int img, int patch, int kp,
                           int dtype, int lo_shift, void* stream);

/* Stand-alone tile transform: (u8/255 - mean[c]) / std[c], HWC -> CHW, fp32 out [B][3][H][W].
 * This is Extractor.transform on an already-224x224 tile (ToTensor + Normalize; reference
 * src/stamp/preprocessing/extractor/h_optimus_0.py:22-30, mstar.py:19-25). */
int amds_tile_normalize_u8(const uint8_t* hwc, float* chw, int B, int H, int W,
                           const float mean_host[3], const float std_host[3], void* stream);

/* Tile background filter (reference src/stamp/preprocessing/tiling.py:280-291 `_has_enough_texture`):
 * frac[b] = mean(cv2.Canny(tile.convert("L"), low, high)) / 255 for u8 HWC tiles [B][S][S][3], S <= 224; the reference keeps
 * a tile when frac >= canny_cutoff (0.02) with low = 40, high = 100.  edges (u8 [B][S][S], 0/255) and gray (u8 [B][S][S])
 * are optional outputs for tests.  Grey = Pillow's fixed-point ITU-R 601; Canny = OpenCV aperture 3, L1 magnitude. */
int amds_tile_edge_fraction_u8(const uint8_t* tiles, float* frac, uint8_t* edges, uint8_t* gray, int B, int S, int low, int high,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * CTransPath tile encoder = ConvStem + Swin-T, the reference's in-tree network
 * (src/stamp/preprocessing/extractor/ctranspath.py; factories ctranspath.py:34-70, chief_ctranspath.py:20-57)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int img;          /* 224 (multiple of 28 * 2^(n_stages-1)) */
    int embed;        /* 96 */
    int n_stages;     /* 4 */
    int depths[4];    /* 2,2,6,2 */
    int heads[4];     /* 3,6,12,24 (dim/32) */
    int dtype;        /* AMDS_F16 / AMDS_BF16: MFMA operand type; the residual stream is fp32 */
    float ln_eps;     /* 1e-5 (nn.LayerNorm / BatchNorm2d defaults) */
} amds_swin_cfg;

typedef struct {
    const float* ln1_w; const float* ln1_b;
    const void*  qkv_w; const float* qkv_b;     /* [3C][C] act dtype, [3C] */
    const float* bias_lane;                     /* [heads][2][2][64][16] fp32: rel-pos bias * log2e in MFMA lane order, -30000 on pad keys */
    const void*  proj_w; const float* proj_b;   /* [C][C] */
    const float* ln2_w; const float* ln2_b;
    const void*  fc1_w; const float* fc1_b;     /* [4C][C] */
    const void*  fc2_w; const float* fc2_b;     /* [C][4C] */
    const void*  mlp_pack;                      /* amds_swin_mlp192_pack image for C = 192 blocks, else NULL */
} amds_swin_block;

typedef struct {
    const float* ln_w; const float* ln_b;       /* [4C] */
    const void*  red_w;                         /* [2C][4C] act dtype, no bias */
} amds_swin_merge;

typedef struct {
    const float* stem;                          /* packed ConvStem parameters, see amds_swin_stem */
    const amds_swin_block* blocks_host;         /* HOST array of sum(depths) structs holding device pointers */
    int n_blocks;
    amds_swin_merge merges[3];
    const float* norm_w; const float* norm_b;
    const uint64_t* mask_bits;                  /* [4][64] shifted-window masks by window type: bit (kt*2+qt)*16 + r of lane's word = masked (-100) */
} amds_swin_weights;

size_t amds_swin_workspace_bytes(const amds_swin_cfg* cfg_host, int batch);

/* tiles: u8 [B][img][img][3] -> feats = model(tiles) with model.head = Identity (ctranspath.py:50-51, 975-988):
 * feats_f16 [B][8*embed] is `.half()` of it (src/stamp/preprocessing/__init__.py:324-325), feats_f32 the unrounded
 * value; either may be NULL.  The (u8/255 - mean)/std transform (ctranspath.py:56-64) is folded into the stem. */
int amds_swin_forward(const amds_swin_cfg* cfg_host, const amds_swin_weights* w_host, const uint8_t* tiles,
                      void* feats_f16, float* feats_f32, int B, int chunk, void* ws, size_t ws_bytes, void* stream);

/* ConvStem + patch LayerNorm (ctranspath.py:386-444, 905-911): u8 tiles -> fp32 tokens x [B][(img/4)^2][embed].
 * params (fp32, BatchNorm folded): a[3]=1/(255 std), b[3]=-mean/std, 2 pad, w1[27][C/8], b1[C/8], w2[9*C/8][C/4],
 * b2[C/4], w3[C/4][C], b3[C], ln_w[C], ln_b[C]; conv weights are stored [ci][ky][kx][co]. */
int amds_swin_stem(const uint8_t* tiles, float* x, const float* params, int B, int img, int embed, float eps, void* stream);

/* (Shifted-)window multi-head attention with relative-position bias (ctranspath.py:510-547, 654-690): tokens stay in
 * raster order, windows of 7x7 are addressed by index arithmetic (roll by -shift), head_dim 32.
 * qkv: act dtype [B*grid^2][ldq] = [q|k|v] x [heads][32]; out: act dtype [B*grid^2][ldo]. */
int amds_window_attention(const void* qkv, long ldq, void* out, long ldo, const float* bias_lane, const uint64_t* mask_bits,
                          int B, int grid, int dim, int heads, int shift, int dtype, void* stream);

/* The whole attention branch of a 96-channel Swin block in one pass over the fp32 residual stream x [B][grid^2][96]:
 *   x += proj(window_attention(qkv(LayerNorm(x))))        (reference ctranspath.py:654-692; 3 heads x 32, window 7)
 * qkv_w [288][96], proj_w [96][96] act dtype (row-major, unpadded); biases, LayerNorm parameters fp32; bias_lane / mask_bits as
 * for amds_window_attention.  Weights and bias tables stay in LDS, q/k/v/probabilities/head outputs only exist in registers. */
int amds_swin_attn96(float* x, const void* qkv_w, const float* qkv_b, const void* proj_w, const float* proj_b,
                     const float* ln_gamma, const float* ln_beta, const float* bias_lane, const uint64_t* mask_bits, int B,
                     int grid, int shift, float ln_eps, int dtype, void* stream);

/* PatchMerging up to the Linear (ctranspath.py:717-736): x fp32 [B][grid^2][dim] -> LayerNorm(concat of the 2x2 cell
 * members (0,0),(1,0),(0,1),(1,1)) as act dtype [B][(grid/2)^2][4*dim]. */
int amds_patch_merge_ln(const float* x, void* y, const float* gamma, const float* beta, int B, int grid, int dim, float eps,
                        int dtype, void* stream);

/* Final LayerNorm + AdaptiveAvgPool1d(1) over tokens (ctranspath.py:981-984): x fp32 [B][L][dim] -> [B][dim]. */
int amds_layernorm_meanpool(const float* x, void* out_f16, float* out_f32, const float* gamma, const float* beta, int B,
                            int L, int dim, float eps, void* stream);

/* Macenko stain normalisation of decoded RGB tiles (OPTIONAL stage, off by default: BASELINE.json's north_star names it, the reference
 * KatherLab/STAMP v2.5.0 does not contain it -- SURVEY.md F1 -- so parity is UNPINNED; checked against oracle/macenko.py's restatement of
 * Macenko et al., ISBI 2009).  tiles, out: u8 [B][H][W][3] (out != tiles); fit_out (may be NULL): fp32 [B][8] = haematoxylin vector (3),
 * eosin vector (3), 99th-percentile concentrations (2), zeros for tiles with fewer than 16 stained pixels (those are copied through).
 * Io = transmitted-light intensity (240), alpha = percentile of the stain angles (1), beta = optical-density threshold (0.15).
 * One workgroup per tile, the tile staged once through LDS (H * W * 3 + 9 KB <= 160 KB: tiles up to 224 x 224). */
int amds_macenko_normalize_u8(const uint8_t* tiles, uint8_t* out, float* fit_out, int B, int H, int W, float Io, float alpha, float beta, void* stream);

/* Supertile -> tiles (the resize + crop of the reference's WSI reader, src/stamp/preprocessing/tiling.py:326-343 and :225-246):
 * rgba u8 [n][S][S][4] as `openslide.read_region` returns it -> PIL `Image.resize((k*t, k*t))` (bicubic on the premultiplied image, 8-bit
 * two-pass resample with 22-bit fixed-point taps, un-premultiply) -> `.convert("RGB")` -> k x k tiles of t x t in row-major order:
 * tiles u8 [n*k*k][t][t][3].  Bit-exact with Pillow (pinned in tests).  bounds int32 [k*t][2] = (first tap, tap count), coef int32
 * [k*t][ksize]: Pillow's precompute_coeffs table for S -> k*t (stamp_amd.tiling.resize_coefficients); ws: n*S*k*t*4 bytes. */
size_t amds_supertiles_to_tiles_workspace_bytes(int n, int S, int k, int t);
int amds_supertiles_to_tiles_u8(const uint8_t* rgba, uint8_t* tiles, int n, int S, int k, int t, const int* bounds, const int* coef,
                                int ksize, void* ws, size_t ws_bytes, void* stream);

/* Resize(O, bicubic) + CenterCrop(t) on RGB u8 tiles -- the transform some of the reference's extractors put in front of their model
 * (src/stamp/preprocessing/extractor/gigapath.py:21-28: Resize(256, BICUBIC), CenterCrop(224)).  torchvision's Resize on a PIL
 * image is Pillow's own two-pass 8-bit resample (bit for bit as above, without the alpha handling); the crop offset is torchvision's
 * int(round((O - t) / 2.0)).  tiles [n][S][S][3] -> out [n][t][t][3]; bounds / coef: the tap tables of an S -> O resize (as for
 * amds_supertiles_to_tiles_u8).  ws: amds_tile_resize_crop_workspace_bytes(n, S, t). */
size_t amds_tile_resize_crop_workspace_bytes(int n, int S, int t);
int amds_tile_resize_crop_u8(const uint8_t* tiles, uint8_t* out, int n, int S, int O, int t, const int* bounds, const int* coef, int ksize, void* ws,
                             size_t ws_bytes, void* stream);

/* Keep-mask compaction on the device, between the texture filter and the tile encoder (reference tiling.py:171-193 `_tiles_with_tissue`
 * drops the rejected tiles one by one on the host): rows i < n of src (row_bytes each, a multiple of 16) whose score[i] >= cutoff (score NULL:
 * all) are appended in order to dst at row *count_dev, which is advanced; slot_out[i] = the row it went to, -1 if rejected (-2 if dst, of
 * capacity_rows, was full).  The fill level never leaves the device; the host reads slot_out when it needs the coordinates. n <= 4096. */
int amds_compact_rows_u8(const uint8_t* src, long row_bytes, const float* score, float cutoff, uint8_t* dst, int capacity_rows,
                         int* count_dev, int* slot_out, int n, void* stream);
/* After the first m rows of an accumulation buffer went to the encoder: rows [m, *count_dev) of src move to the front of dst (at most
 * max_rows of them: the launch geometry) and *count_dev -= m, all in stream order and without the host knowing the fill level. */
int amds_compact_shift_u8(const uint8_t* src, uint8_t* dst, long row_bytes, int m, int max_rows, int* count_dev, void* stream);
/* n_words 32-bit words from device memory to HOST-MAPPED (pinned) memory, written by a kernel's stores over the host link -- not by a copy command:
 * a device-to-host copy that waits for an encoder call sits in the copy queue and holds up the host-to-device copies of the next tiles submitted after it
 * (stamp_amd/preprocess.py).  zero_src: the source words are cleared behind the copy (event counters read and reset in stream order).  The pipelined
 * slide loop exports feature rows (fp16 pairs) and the folded LayerNorms' range counters this way; reference: `model(tiles).half().cpu()`,
 * src/stamp/preprocessing/__init__.py:324-327. */
int amds_export_words(void* src_dev, void* dst_host_mapped, long n_words, int zero_src, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TICON tile contextualiser in the form the reference's extractor uses it (src/stamp/preprocessing/extractor/ticon.py:691-718
 * `HOptimusTICON.forward`: every tile's embedding ALONE through `EncoderDecoder.forward` :543-562, one token, zero coordinates):
 * input projection (Linear, SiLU, Linear, LayerNorm :80-99), `depth` blocks (x += g1 * proj(v_proj(LN(x))): with one key the attention
 * :183-215 is the identity on its value; x += g2 * fc2(silu(x1) * x2), (x1 | x2) = fc1(LN(x)) :54-77), final LayerNorm :506.
 * Exact fp32 (fp32 MFMA): the stage is 0.05 % of the tile encoder's flops.  All pointers fp32 device; proj_w / proj_b and fc2_w / fc2_b carry the
 * block's LayerScale (gamma1 / gamma2 multiplied into rows and bias by the host at pack time).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float* ln1_w; const float* ln1_b;        /* residual1.norm                                  */
    const float* v_w; const float* v_b;            /* residual1.fn.v_proj            [dim][dim]       */
    const float* proj_w; const float* proj_b;      /* gamma1 * residual1.fn.proj     [dim][dim]       */
    const float* ln2_w; const float* ln2_b;        /* residual2.norm                                  */
    const float* fc1_w; const float* fc1_b;        /* residual2.fn.fc1               [hidden][dim]    */
    const float* fc2_w; const float* fc2_b;        /* gamma2 * residual2.fn.fc2      [dim][hidden/2]  */
} amds_ticon_block;
typedef struct {
    int in_dim, dim, hidden, depth;                /* hidden = int(dim * 16 / 3) (:58-66), even */
    const float* in_fc1_w; const float* in_fc1_b;  /* input_proj_<key>.fc1           [dim][in_dim]    */
    const float* in_fc2_w; const float* in_fc2_b;  /* input_proj_<key>.fc2           [dim][dim]       */
    const float* in_norm_w; const float* in_norm_b;
    const amds_ticon_block* blocks_host;           /* HOST array of `depth` entries (encoder.blocks)  */
    const float* norm_w; const float* norm_b;      /* enc_norm                                        */
} amds_ticon_weights;
size_t amds_ticon_tile_workspace_bytes(const amds_ticon_weights* w_host, int n_tiles);
/* emb: [n_tiles][in_dim] AMDS_F32 / AMDS_F16 (the tile encoder's feature rows); out: [n_tiles][dim] in out_dtype (AMDS_F32 or AMDS_F16). */
int amds_ticon_tile_forward(const amds_ticon_weights* w_host, const void* emb, int emb_dtype, void* out, int out_dtype, int n_tiles, void* ws,
                            size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gated-attention pooling (CHIEF slide encoder; reference
 * src/stamp/encoding/encoder/chief.py:74-89 CHIEFModel.forward, :255-275 Attn_Net_Gated)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float* fc_w;  const float* fc_b;   /* [L][F], [L]   attention_net.0 */
    const float* a_w;   const float* a_b;    /* [D][L], [D]   attention_net.3.attention_a.0 */
    const float* b_w;   const float* b_b;    /* [D][L], [D]   attention_net.3.attention_b.0 */
    const float* c_w;   const float* c_b;    /* [1][D], [1]   attention_net.3.attention_c */
    const float* packed;                     /* optional: amds_gated_attn_pack's image of fc_w / a_w / b_w (amds_gated_attn_packed_floats floats), or NULL --
                                              * the fused entries then pack into their workspace on every call (one more launch) */
} amds_gap_weights;

/* The fused pooling kernels read fc_w, a_w and b_w from a tile-ordered copy (each LDS tile image one contiguous run; csrc/gap_fused.hip).  Pack once per set
 * of weights and pass the buffer in amds_gap_weights.packed.  0 floats <=> the shape is not one the fused kernels take. */
size_t amds_gated_attn_packed_floats(int F, int L, int D);
int amds_gated_attn_pack(const amds_gap_weights* w_host, float* packed, int F, int L, int D, void* stream);

size_t amds_gated_attn_pool_workspace_bytes(int N, int F, int L, int D);
/* x: fp32 [N][F]; out: fp32 [F] = softmax_N(A) @ x  ("WSI_feature"); attn_raw: fp32 [N] (may be NULL).
 * fp32 arithmetic throughout (the reference runs this encoder in fp32, chief.py:117).  One launch (csrc/gap_fused.hip) when
 * amds_gated_attn_pool_batched_supported(F, L, D) -- CHIEF's two size_args are -- else the six-launch form below. */
int amds_gated_attn_pool(const float* x, const amds_gap_weights* w_host, float* out, float* attn_raw,
                         int N, int F, int L, int D, void* ws, size_t ws_bytes, void* stream);

/* MANY bags in one launch -- what the reference's per-slide loop (src/stamp/encoding/encoder/__init__.py:42-118 `encode_slides_`, chief.py:119-127
 * `_generate_slide_embedding`, :129-135 the patient concatenation) and EAGLE's scoring pass (eagle.py:96-118) hand over when the feature matrices of
 * several slides are resident: x = their rows concatenated [total_rows][F]; row_offsets (DEVICE, int64 [bags + 1], row_offsets[0] = 0,
 * row_offsets[bags] = total_rows) delimits the bags; out [bags][F]; attn_raw [total_rows] or NULL.  Every bag must be non-empty (the reference raises
 * on an empty feature matrix; the row of an empty bag is left untouched here).  Same arithmetic per bag as amds_gated_attn_pool; the result
 * is reproducible run to run (no atomics on data; partials merge in a fixed order).  Shapes: amds_gated_attn_pool_batched_supported (L = 256 or
 * 512, F and D multiples of 16).  row_offsets may be NULL when bags == 1.
 * mode: AMDS_GAP_AUTO picks the decomposition by the total row count -- up to 8192 rows a workgroup owns 16 rows and its four waves split the hidden
 * units (every SIMD of the chip busy on a single small bag), beyond that a workgroup owns 64 rows and a wave 16 of them end to end (weights shared
 * through LDS; 0.8 of the fp32 MFMA peak on large batches).  The two associate the gate pre-activation sums differently (last-bit differences);
 * AMDS_GAP_SLAB / AMDS_GAP_SPLIT force one (SPLIT: at most 12288 rows in total, F % 32 == 0, D % 64 == 0) -- with a forced mode a bag's result does
 * not depend on what else is in the batch. */
enum { AMDS_GAP_AUTO = 0, AMDS_GAP_SLAB = 1, AMDS_GAP_SPLIT = 2 };
int amds_gated_attn_pool_batched_supported(int F, int L, int D);
size_t amds_gated_attn_pool_batched_workspace_bytes(long total_rows, int bags, int F, int L, int D);
int amds_gated_attn_pool_batched(const float* x, const long long* row_offsets, int bags, long total_rows, const amds_gap_weights* w_host, float* out,
                                 float* attn_raw, int F, int L, int D, int mode, void* ws, size_t ws_bytes, void* stream);

/* The six-launch form (two exact-fp32 GEMMs through HBM, gate, softmax statistics, partial pooling, reduce; csrc/gap.hip): any F, L multiple of 4.
 * amds_gated_attn_pool falls back to it for shapes the fused kernel does not take; exported for the A/B in tools/gap_only.py. */
size_t amds_gated_attn_pool_unfused_workspace_bytes(int N, int F, int L, int D);
int amds_gated_attn_pool_unfused(const float* x, const amds_gap_weights* w_host, float* out, float* attn_raw,
                                 int N, int F, int L, int D, void* ws, size_t ws_bytes, void* stream);

/* EAGLE's tile selection (reference src/stamp/encoding/encoder/eagle.py:106-118: `torch.topk(attention_raw, min(25, N))`, then the mean of
 * the matching rows of the aggregation features): idx_out[r], r < k, = index of the r-th largest score (ties: the lower index first),
 * mean_out[c] = mean_r rows[idx_out[r]][c].  1 <= k <= min(32, n); rows f32 / f16 [n][ld]; score = attn_raw of amds_gated_attn_pool. */
int amds_topk_rows_mean(const float* score, int n, int k, const void* rows, long ld, int cols, int rows_dtype, int* idx_out, float* mean_out,
                        void* stream);

/* KEEP's image head (reference src/stamp/preprocessing/extractor/keep.py:38-47, `KEEPImageModel.encode_image`): out = normalize(W2 gelu(W1 feats + b1) + b2),
 * F.normalize's x / max(||x||_2, 1e-12), exact fp32.  feats: the ViT-L/16 trunk's class features [rows][in_dim] (AMDS_F32 / AMDS_F16); w1 [proj][in_dim],
 * w2 [proj][proj] fp32; out fp32 [rows][proj]. */
size_t amds_proj_head_l2norm_workspace_bytes(int rows, int in_dim, int proj_dim);
int amds_proj_head_l2norm(const void* feats, int feats_dtype, const float* w1, const float* b1, const float* w2, const float* b2, float* out, int rows,
                          int in_dim, int proj_dim, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small rows of the MIL path
 * ---------------------------------------------------------------------------------------------- */

/* Fixed-size bag building: dst[i] = i < n_idx ? (out dtype) src[idx[i]] : 0 for i < n_out -- the gather + ".float()"
 * + zero padding of reference src/stamp/modeling/data.py:811-862 (`_to_fixed_size_bag`); the indices (randperm /
 * linspace.round) are drawn by the caller.  idx: int64 device.  dtype pairs: f16->f32, f16->f16, f32->f32. */
int amds_gather_rows(const void* src, long src_ld, const long* idx, int n_idx, void* dst, long dst_ld, int n_out,
                     int cols, int in_dtype, int out_dtype, void* stream);

/* vary_precision (reference src/stamp/modeling/transforms.py:5-29): out = bits & (~0 << shifts[i]) on the 16- or
 * 32-bit pattern of every element; shifts (u8, drawn by the caller with torch.randint like the reference). */
int amds_vary_precision(const void* bits_in, const uint8_t* shifts, void* bits_out, long n, int elem_bytes, void* stream);

/* Mean over tiles: x [B][T][F] (f16/f32) -> out fp32 [B][F] (reference src/stamp/modeling/models/mlp.py:40-41). */
int amds_mean_pool(const void* x, float* out, int B, int T, int F, int in_dtype, void* stream);
/* its gradient: dx fp32 [B][T][F] = dy[B][F] / T (training the MLP / Linear heads on bags, mlp.py:40-41) */
int amds_mean_pool_bwd(const float* dy, float* dx, int B, int T, int F, void* stream);

/* out[M][N] = (relu?)(x[M][K] w[N][K]^T + bias) in exact fp32 (fp32-input MFMA); MLP / Linear heads, mlp.py:24-33. */
int amds_linear_f32(const float* x, const float* w, const float* bias, float* out, int M, int N, int K, int relu, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MIL `vit` head, deploy / validation forward as one call (reference
 * src/stamp/modeling/models/vision_tranformer.py:332-384 VisionTransformer.forward in eval mode; called from
 * src/stamp/modeling/models/__init__.py:288-313 validation_step / predict_step of the tile-level Lightning wrappers)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int n_feats;    /* dim_input  */
    int dim;        /* dim_model (head_dim = dim / heads <= 64, dim % 4 == 0) */
    int heads;
    int ff;         /* dim_feedforward */
    int classes;    /* dim_output */
    int layers;
    int alibi;      /* use_alibi */
    int dtype;      /* MFMA operand type of the 16-bit weights below: AMDS_F16 (inference packs) or AMDS_BF16 */
} amds_mil_vit_cfg;

/* Device pointers.  Padded sizes: Fp / Dp / FFp = n_feats / dim / ff rounded up to 256, Ha = heads rounded up to 4, Da = 64 Ha; padding
 * rows, columns and heads are zero (amds_cast_pad).  16-bit matrices are [N][K] row-major in cfg.dtype unless noted. */
typedef struct {
    const float* ln1_w; const float* ln1_b;     /* [dim]                       layers.l.0.norm                            :183, 215 */
    const void* in_w;   const float* in_b;      /* [3 Da][Dp], [3 Da]          in_proj (q | k | v, head h at rows 64h..64h+63; head_dim < 64:
                                                 *                             q rows scaled by sqrt(64 / head_dim))      :191 / :100-120 */
    const void* out_w;  const float* out_b;     /* [Dp][Da], [Dp]              out_proj / mhsa.fc; BF16 when cfg.alibi     :191 / :121, 154 */
    const float* head_scale;                    /* [Ha] bias_scale_h / running_mean_h (ALiBi only, else NULL)              :31, 60 */
    const float* ln2_w; const float* ln2_b;     /* [dim]                       layers.l.1.0                               :163 */
    const void* fc1_w;  const float* fc1_b;     /* [FFp][Dp], [FFp]            layers.l.1.1                               :164 */
    const void* fc2_w;  const float* fc2_b;     /* [Dp][FFp], [Dp]             layers.l.1.4                               :167 */
    /* training only (amds_mil_vit_train_*), NULL otherwise: the 16-bit matrices transposed, [K][N] (amds_transpose16), for the
     * input-gradient GEMMs, and the two factors of head_scale (ALiBi) */
    const void* in_wt;  const void* out_wt; const void* fc1_wt; const void* fc2_wt;
    const float* bias_scale; const float* inv_running_mean;      /* [Ha] each */
} amds_mil_vit_layer;

typedef struct {
    const float* class_token;                   /* [Dp]                                                                    :312, 347 */
    const void* proj_w; const float* proj_b;    /* [Dp][Fp], [Dp]              project_features.0                          :314, 342 */
    const amds_mil_vit_layer* layers_host;      /* HOST array of cfg.layers entries */
    const float* norm_w; const float* norm_b;   /* [dim]                       transformer.norm                           :278, 294 */
    const float* head_w; const float* head_b;   /* [classes][dim] fp32, [classes]   mlp_head.0                            :329, 384 */
    const void* proj_wt;                        /* [Fp][Dp] transposed proj_w: only for the gradient w.r.t. the bags (heat-maps), else NULL */
} amds_mil_vit_weights;

size_t amds_mil_vit_workspace_bytes(const amds_mil_vit_cfg* cfg_host, int n_bags, int n_tiles);
/* bags: [n_bags][n_tiles][n_feats] contiguous, bags_dtype AMDS_F32 / AMDS_F16 / AMDS_BF16 (staged to cfg.dtype rows of pitch Fp unless it
 * already has that form).  coords: fp32 [n_bags][n_tiles][2], required when cfg.alibi.  mask: u8 [n_bags][n_tiles], 1 = padded tile, or NULL
 * (the reference's `mask` argument, :359-381).  logits: fp32 [n_bags][classes].  Launches only, on `stream`; ws 256-byte aligned. */
int amds_mil_vit_forward(const amds_mil_vit_cfg* cfg_host, const amds_mil_vit_weights* w_host, const void* bags, int bags_dtype,
                         const float* coords, const uint8_t* mask, float* logits, int n_bags, int n_tiles, void* ws, size_t ws_bytes,
                         void* stream);

/* The TRAINING step of the same head, forward and backward as one call each (reference: the train-mode forward of
 * vision_tranformer.py:332-384 with its Dropout sites :157-169, :191, :314-318 live, and loss.backward() through it,
 * src/stamp/modeling/models/__init__.py:239-279).  cfg.dtype = AMDS_BF16 or AMDS_F16: the type of EVERY 16-bit tensor of the step (operand copies of the
 * weights, saved activations, 16-bit gradients); fp32 accumulation, residual stream and gradients.  bf16 = torch's float32_matmul_precision "medium"; fp16
 * (10 explicit mantissa bits, the TF32 class) = "high", which the reference sets before training (src/stamp/modeling/train.py:519) -- the caller then
 * multiplies dlogits by a power of two (1024 in stamp_amd/mil_train.py) and divides the gradients by it.  Dropout masks are counter-based functions of (seed, site, element): the backward regenerates them from the same
 * amds_mil_vit_dropout.  The ALiBi running means (:24-29) are updated by the caller BEFORE the forward (amds_cdist_rowsum). */
typedef struct {
    float p_proj;       /* project_features' Dropout = the constructor's `dropout`                    :314-318 */
    float p_att;        /* nn.MultiheadAttention's dropout = the constructor's `dropout` (ignored when cfg.alibi)  :191 */
    float p_ff;         /* feed_forward's two Dropouts (0.5: the reference never forwards `dropout` to them)       :157-169, :268-271 */
    uint64_t seed;
    int cls_tail;       /* the last block on its class rows alone: 1 / 0 = yes / no, the SAME value for a forward and the backwards that read its arena
                         * (the arena's layout depends on it); -1 = each call asks the context (amds_get_mil_cls_tail) -- only safe if nobody toggles it between */
} amds_mil_vit_dropout;  /* p = 0, seed = 0 = eval-mode arithmetic with saved activations (parity tests against autograd) */

/* fp32 device buffers in the PADDED layout of the weights they belong to (amds_mil_vit_layer / _weights); the caller slices the
 * reference shapes out of them (padding rows / columns / heads receive zeros or don't-care values). */
typedef struct {
    float* ln1_w; float* ln1_b;     /* [dim] */
    float* in_w;  float* in_b;      /* [3 Da][Dp], [3 Da] */
    float* out_w; float* out_b;     /* [Dp][Da], [Dp] */
    float* bias_scale;              /* [Ha]  (ALiBi only) */
    float* ln2_w; float* ln2_b;     /* [dim] */
    float* fc1_w; float* fc1_b;     /* [FFp][Dp], [FFp] */
    float* fc2_w; float* fc2_b;     /* [Dp][FFp], [Dp] */
} amds_mil_vit_layer_grads;
typedef struct {
    float* class_token;             /* [Dp] */
    float* proj_w; float* proj_b;   /* [Dp][Fp], [Dp] */
    const amds_mil_vit_layer_grads* layers_host;     /* HOST array of cfg.layers entries */
    float* norm_w; float* norm_b;   /* [dim] */
    float* head_w; float* head_b;   /* [classes][dim], [classes] */
} amds_mil_vit_grads;

/* `saved`: the activations the backward reads (one arena, written by the forward, read-only afterwards: several backwards may follow one
 * forward, e.g. one per class for a Jacobian).  `ws`: scratch of the backward.  0 on a bad configuration (amds_last_error). */
size_t amds_mil_vit_train_saved_bytes(const amds_mil_vit_cfg* cfg_host, int n_bags, int n_tiles);
size_t amds_mil_vit_train_workspace_bytes(const amds_mil_vit_cfg* cfg_host, int n_bags, int n_tiles, int split_k);
/* bags / coords / logits as amds_mil_vit_forward (no mask: the reference trains with mask=None, models/__init__.py:252). */
int amds_mil_vit_train_forward(const amds_mil_vit_cfg* cfg_host, const amds_mil_vit_weights* w_host, const void* bags, int bags_dtype,
                               const float* coords, const amds_mil_vit_dropout* drop_host, float* logits, int n_bags, int n_tiles,
                               void* saved, size_t saved_bytes, void* stream);
/* dlogits: fp32 [n_bags][classes].  grads_host NULL = no parameter gradients (input gradient only); dbags: fp32 [n_bags*n_tiles][Fp] or
 * NULL.  Weight gradients dW = dy^T x contract over the token dimension in `split_k` fp32 partials (32 is the tuned value) that are summed
 * in a fixed order: the step is deterministic. */
int amds_mil_vit_train_backward(const amds_mil_vit_cfg* cfg_host, const amds_mil_vit_weights* w_host, const float* dlogits,
                                const amds_mil_vit_dropout* drop_host, int n_bags, int n_tiles, const void* saved, size_t saved_bytes,
                                const amds_mil_vit_grads* grads_host, float* dbags, int split_k, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * barspoon head, deploy / validation forward as one call (reference src/stamp/modeling/models/barspoon.py:27-205
 * `EncDecTransformer` in eval mode: projector, sinusoidal position encoding, pre-norm transformer encoder over the tiles, one class
 * token per target decoded against the tiles by a pre-norm transformer decoder, one Linear head per target)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int n_feats;                /* d_features */
    int dim;                    /* d_model (head_dim = dim / heads <= 64 for both stacks, dim % 4 == 0) */
    int enc_heads, dec_heads;   /* num_encoder_heads, num_decoder_heads */
    int ff;                     /* dim_feedforward */
    int enc_layers, dec_layers;
    int n_targets;
    int positional_encoding;
    int dtype;                  /* MFMA operand type of the 16-bit weights: AMDS_F16 or AMDS_BF16 */
} amds_barspoon_cfg;
/* decoder layer: fp32 device pointers in the reference's shapes, except the cross-attention's K / V projection, which runs over the tiles:
 * ca_kv_w 16-bit [2 Db][Dp] (Db = 64 dec_heads: K heads then V heads, each padded to 64 channels; Dp = dim rounded up to 256), ca_kv_b fp32 [2 Db] */
typedef struct {
    const float* ln1_w; const float* ln1_b;             /* norm1                                                   */
    const float* sa_in_w; const float* sa_in_b;         /* self_attn.in_proj_{weight,bias}      [3 dim][dim], [3 dim] */
    const float* sa_out_w; const float* sa_out_b;       /* self_attn.out_proj                   [dim][dim], [dim]   */
    const float* ln2_w; const float* ln2_b;             /* norm2                                                   */
    const float* ca_q_w; const float* ca_q_b;           /* multihead_attn.in_proj rows 0..dim   [dim][dim], [dim]   */
    const void* ca_kv_w; const float* ca_kv_b;          /* multihead_attn.in_proj rows dim..3dim, padded (above)    */
    const float* ca_out_w; const float* ca_out_b;       /* multihead_attn.out_proj              [dim][dim], [dim]   */
    const float* ln3_w; const float* ln3_b;             /* norm3                                                   */
    const float* fc1_w; const float* fc1_b;             /* linear1                              [ff][dim], [ff]     */
    const float* fc2_w; const float* fc2_b;             /* linear2                              [dim][ff], [dim]    */
} amds_barspoon_dec_layer;
typedef struct {
    const void* proj_w; const float* proj_b;            /* projector.0: 16-bit [Dp][Fp], fp32 [Dp] (padded like amds_mil_vit_weights.proj_w) */
    const amds_mil_vit_layer* enc_layers_host;          /* HOST array: transformer_encoder.layers.l in the MIL `vit` head's padded layer form
                                                         * (norm1 -> ln1, self_attn -> in / out, norm2 -> ln2, linear1 -> fc1, linear2 -> fc2) */
    const float* class_tokens;                          /* [n_targets][dim], in target order                        :135-140 */
    const amds_barspoon_dec_layer* dec_layers_host;     /* HOST array */
    const float* const* head_w_host;                    /* HOST arrays of n_targets device pointers: heads.<t>.weight [n_out_t][dim], .bias [n_out_t] */
    const float* const* head_b_host;
    const int* n_out_host;                              /* HOST array: n_out_t */
    const float* pe_div;                                /* [dim / 4] fp32: 100000^(i / dim), as torch computes it   :176-178 */
} amds_barspoon_weights;
size_t amds_barspoon_workspace_bytes(const amds_barspoon_cfg* cfg_host, int n_bags, int n_tiles);
/* bags [n_bags][n_tiles][n_feats] (AMDS_F32 / F16 / BF16), positions fp32 [n_bags][n_tiles][2] (required when positional_encoding);
 * logits fp32 [n_bags][sum_t n_out_t], target t at column offset sum_{s<t} n_out_s.  Launches only, on `stream`; ws 256-byte aligned. */
int amds_barspoon_forward(const amds_barspoon_cfg* cfg_host, const amds_barspoon_weights* w_host, const void* bags, int bags_dtype,
                          const float* positions, float* logits, int n_bags, int n_tiles, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TransMIL building blocks (reference src/stamp/modeling/models/trans_mil.py), fp32 throughout
 * ---------------------------------------------------------------------------------------------- */

/* Precision of the fp32 batched products below -- the library's counterpart of `torch.set_float32_matmul_precision`, which the reference sets to "high"
 * before every training run (src/stamp/modeling/train.py:519) and to "medium" for deployment (src/stamp/modeling/deploy.py:398).  Process-wide, like torch's.
 *   AMDS_MATMUL_HIGHEST (default): fp32 operands on the fp32 MFMA (v_mfma_f32_32x32x2_f32), exact products.
 *   AMDS_MATMUL_HIGH: every operand value as the sum of two bf16 numbers, three bf16 MFMAs per product (hi hi + hi lo + lo hi), fp32 accumulate: ~16 mantissa
 *     bits per factor -- one of the two implementations torch documents for "high" (the other, TF32, keeps 10).
 * Applies to amds_bgemm_f32's tiled kernels as the MIL heads call them (every product of the TransMIL / Nystrom paths, the MLP heads' training GEMMs, the
 * heads' own small products); the 64 x 64 fallback kernel for small or unaligned products, everything that is not a matrix product, and the feature-extraction
 * paths that promise exact fp32 (the tile encoder's exact class-token stream, TICON, barspoon's class side) stay exact at every level. */
#define AMDS_MATMUL_HIGHEST 0
#define AMDS_MATMUL_HIGH 1
int amds_set_matmul_precision(amds_ctx* ctx, int level);
int amds_get_matmul_precision(amds_ctx* ctx);
/* Batched fp32 GEMM on the exact-fp32 MFMA: for z = (o, i), o < outer, i < inner:
 *   C[o,i] (+)= diag*I + alpha * A[o,i] * op(B[o,i]) + bias[n];  op(B) = B^T if transb (B stored [N][K]) else B ([K][N]).
 * Operand z starts at base + o*s?o + i*s?i (elements), so head slices of a packed qkv tensor are addressed in place.
 * transb bit 1 (value 2): A is stored [K][M] with pitch lda and the product is A^T op(B) -- the backward's "x^T dy" products without an
 * explicit transpose (vector loads when K, lda, ldb and the batch strides are multiples of 4 and the operands 16-byte aligned,
 * element loads otherwise). */
int amds_bgemm_f32(const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int transb,
                   float* C, int ldc, long sCo, long sCi, int outer, int inner, int M, int N, int K, float alpha,
                   float diag, const float* bias, int accumulate, void* stream);
/* The same product with TWO outputs: C = alpha A op(B) + diag I and C2 = alpha2 A op(B) + diag2 I (C2 laid out like C; no bias, no accumulate): the Moore-Penrose
 * iteration's `xz = x @ z` and `7 I - xz` (reference src/stamp/modeling/models/trans_mil.py:31-33) from one pass over the operands; bits of two amds_bgemm_f32 calls. */
int amds_bgemm_f32_dual(const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int transb, float* C, float* C2, int ldc,
                        long sCo, long sCi, int outer, int inner, int M, int N, int K, float alpha, float diag, float alpha2, float diag2, void* stream);
/* In-place softmax over the last dim of a contiguous [rows][cols] fp32 matrix (trans_mil.py:145). */
int amds_softmax_rows(float* x, long rows, int cols, void* stream);
/* Landmarks: out[z][j][c] = scale * sum_{t<l} x[z][j*l + t][c] (trans_mil.py:114-124, mask=None). */
int amds_landmark_mean(const float* x, long sxo, long sxi, int ld, float* out, int outer, int inner, int m, int l, int d,
                       float scale, void* stream);
/* z0 = x^T / (max row-abs-sum * max col-abs-sum), maxima over ALL nmat matrices (trans_mil.py:23-28). scratch8: 8 B. */
int amds_pinv_init(const float* x, float* z, int nmat, int n, void* scratch8, void* stream);
/* out[z][t][c] += sum_k w[i][k] v[z][t+k-taps/2][c]: per-head depth-wise conv along the sequence (trans_mil.py:71-78,150-151). */
int amds_dwconv_seq(const float* v, long svo, long svi, int ldv, const float* w, float* out, long soo, long soi, int ldo,
                    int outer, int inner, int n, int d, int taps, void* stream);
/* The same convolution for ONE position `row` of every sequence: out[z][c] += sum_k w[i][k] v[z][row+k-taps/2][c], out compact (strides soo / soi per outer / inner index). */
int amds_dwconv_seq_row(const float* v, long svo, long svi, int ldv, const float* w, float* out, long soo, long soi, int outer, int inner, int n, int d,
                        int taps, int row, void* stream);
/* PPEG: y = x + dw7x7(x) + dw5x5(x) + dw3x3(x) on the H x W token grid, class token passed through (trans_mil.py:274-283).
 * x, y: [B][1+H*W][C]; w7 [C][49], w5 [C][25], w3 [C][9], biases [C]. */
int amds_ppeg(const float* x, float* y, const float* w7, const float* b7, const float* w5, const float* b5, const float* w3,
              const float* b3, int B, int H, int W, int C, void* stream);

/* The whole deploy / validation forward of the TransMIL head as one call (reference src/stamp/modeling/models/trans_mil.py:299-326 in eval
 * mode; called from the same Lightning steps as the `vit` head, models/__init__.py:288-313): fp32 throughout.  heads = 8, landmarks =
 * dim / 2, 6 pseudo-inverse iterations, 33-tap residual convolution (:252-254, :52). */
typedef struct {
    int n_feats; int dim; int classes;             /* dim_input, dim_hidden (multiple of 8), dim_output */
    int train_cls_tail;                            /* amds_transmil_train_forward / _backward only: 1 = layer2's attention output, to_out, Dropout and their backward
                                                    * run on the class rows alone (only x[:, 0] is read after layer2, trans_mil.py:322-325; same values as 0 = every
                                                    * row, to fp32 rounding of the reordered sums); -1 = ask the context (amds_set_mil_cls_tail).  The two calls of one
                                                    * step must agree: the saved arena's layout follows it.  The inference forward ignores it (context setting). */
} amds_transmil_cfg;
typedef struct {
    const float* norm_w; const float* norm_b;      /* [dim]            layerN.norm                       :248 */
    const float* qkv_w;                            /* [3 dim][dim]     layerN.attn.to_qkv (no bias)      :64 */
    const float* out_w; const float* out_b;        /* [dim][dim],[dim] layerN.attn.to_out.0              :66 */
    const float* conv_w;                           /* [8][33]          layerN.attn.res_conv.weight       :71-78 */
} amds_transmil_layer;
typedef struct {
    const float* fc1_w; const float* fc1_b;        /* [dim][n_feats], [dim]     _fc1.0                   :290 */
    const float* cls_token;                        /* [dim]                                              :291 */
    amds_transmil_layer layer[2];                  /* layer1, layer2                                     :293-294 */
    const float* ppeg_w7; const float* ppeg_b7;    /* [dim][49], [dim]          pos_layer.proj           :269 */
    const float* ppeg_w5; const float* ppeg_b5;    /* [dim][25]                 pos_layer.proj1          :270 */
    const float* ppeg_w3; const float* ppeg_b3;    /* [dim][9]                  pos_layer.proj2          :271 */
    const float* norm_w; const float* norm_b;      /* [dim]                     norm                     :295 */
    const float* fc2_w; const float* fc2_b;        /* [classes][dim], [classes] _fc2                     :296 */
} amds_transmil_weights;                           /* device pointers, fp32 */
size_t amds_transmil_workspace_bytes(const amds_transmil_cfg* cfg_host, int n_bags, int n_tiles);
/* bags: [n_bags][n_tiles][n_feats] contiguous, AMDS_F32 / AMDS_F16 / AMDS_BF16 (cast to fp32 like the reference); logits fp32
 * [n_bags][classes].  Launches only, on `stream`; ws 256-byte aligned. */
int amds_transmil_forward(const amds_transmil_cfg* cfg_host, const amds_transmil_weights* w_host, const void* bags, int bags_dtype, float* logits,
                          int n_bags, int n_tiles, void* ws, size_t ws_bytes, void* stream);

/* One TransMIL layer's attention in TRAINING, forward and backward as one call each (reference trans_mil.py:81-163 with mask = None, the
 * residual of :263, `to_out`'s Dropout(0.1) of :66 live when p_drop > 0; loss.backward() through it, models/__init__.py:239-279):
 *   fwd:  x_res[b][n][dim] += Dropout(to_out(NystromAttention(y[b][n][dim])))     -- every intermediate the backward reads goes to `saved`
 *   bwd:  dy[b][n][dim] = gradient w.r.t. y given dx = the gradient of the residual stream after the block; parameter gradients into
 *         grads_host (reference shapes, or NULL).  `saved` is read-only in the backward (several backwards may follow one forward).
 * The dropout mask is the counter-based function of (seed, stream_id, element) of amds_dropout_add: pass the same triple to both. */
typedef struct {
    float* qkv_w;       /* [3 dim][dim]   attn.to_qkv.weight   */
    float* out_w;       /* [dim][dim]     attn.to_out.0.weight */
    float* out_b;       /* [dim]          attn.to_out.0.bias   */
    float* conv_w;      /* [8][33]        attn.res_conv.weight */
} amds_nystrom_grads;
size_t amds_nystrom_attn_saved_bytes(int dim, int n_bags, int n_tokens);
size_t amds_nystrom_attn_workspace_bytes(int dim, int n_bags, int n_tokens);
int amds_nystrom_attn_fwd(const amds_transmil_layer* w_host, int dim, const float* y, float* x_res, int n_bags, int n_tokens, float p_drop,
                          uint64_t seed, uint32_t stream_id, void* saved, size_t saved_bytes, void* stream);
int amds_nystrom_attn_bwd(const amds_transmil_layer* w_host, int dim, const float* dx, float* dy, const amds_nystrom_grads* grads_host, int n_bags,
                          int n_tokens, float p_drop, uint64_t seed, uint32_t stream_id, const void* saved, size_t saved_bytes, void* ws,
                          size_t ws_bytes, void* stream);

/* The whole TransMIL TRAINING step's forward and backward, one call each: the loop around amds_nystrom_attn_fwd / _bwd (reference
 * trans_mil.py:299-325 in train mode -- Dropout(0.1) on both `to_out`s, dropout sites 1 and 2 of `seed` -- and loss.backward() through it).
 * Gradients in the reference's shapes; PPEG's three kernels share one tap-correlation table ppeg_corr [50][dim] (amds_ppeg_wgrad: the 7x7 kernel's
 * gradient is taps 0..48, the 5x5 / 3x3 ones its central 25 / 9 taps, every bias tap 49 -- the host slices). */
typedef struct {
    float* fc1_w; float* fc1_b;          /* [dim][n_feats], [dim] */
    float* cls_token;                    /* [dim] */
    struct { float* norm_w; float* norm_b; float* qkv_w; float* out_w; float* out_b; float* conv_w; } layer[2];
    float* ppeg_corr;                    /* [50][dim] */
    float* norm_w; float* norm_b;        /* [dim] */
    float* fc2_w; float* fc2_b;          /* [classes][dim], [classes] */
} amds_transmil_grads;
size_t amds_transmil_train_saved_bytes(const amds_transmil_cfg* cfg_host, int n_bags, int n_tiles);
size_t amds_transmil_train_workspace_bytes(const amds_transmil_cfg* cfg_host, int n_bags, int n_tiles);
int amds_transmil_train_forward(const amds_transmil_cfg* cfg_host, const amds_transmil_weights* w_host, const void* bags, int bags_dtype, float p_drop,
                                uint64_t seed, float* logits, int n_bags, int n_tiles, void* saved, size_t saved_bytes, void* stream);
int amds_transmil_train_backward(const amds_transmil_cfg* cfg_host, const amds_transmil_weights* w_host, const float* dlogits, float p_drop, uint64_t seed,
                                 int n_bags, int n_tiles, const void* saved, size_t saved_bytes, const amds_transmil_grads* grads_host, float* dbags, void* ws,
                                 size_t ws_bytes, void* stream);

/* Backward pieces of the TransMIL head (training: the reference differentiates trans_mil.py with autograd inside
 * LitTileClassifier._step, src/stamp/modeling/models/__init__.py:239-279); fp32 like the forward.  The matrix products of the
 * backward are amds_bgemm_f32 calls.
 *   amds_softmax_rows_bwd     dp <- p o (dp - rowsum(p o dp)), in place                              (three softmaxes, :145)
 *   amds_landmark_mean_bwd    dx[z][j*l+t][c] (+)= scale * dout[z][j][c]                              (landmarks, :114-124)
 *   amds_dwconv_seq_wgrad     dw[head][k] = sum_{bag,t,c} dout[t][c] * v[t+k-taps/2][c]               (res_conv, :150-151; its data gradient
 *                             is amds_dwconv_seq with the taps reversed)
 *   amds_ppeg_wgrad           dcorr[tap][c], tap = r*7+q < 49: sum dy[b,i,j,c] * x[b,i+r-3,j+q-3,c]; tap 49: sum dy.  The 7x7 kernel's
 *                             gradient is taps 0..48, the 5x5 / 3x3 ones its central 25 / 9 taps, every bias tap 49 (:274-283; the data
 *                             gradient is amds_ppeg with flipped kernels and zero biases)
 *   amds_relu_bwd             dz = h > 0 ? dh : 0                                                       (_fc1, :290)
 *   amds_pinv_init_bwd        dx += d/dx of z0 = x^T / (max row-abs-sum * max col-abs-sum), INCLUDING the path through the two global
 *                             maxima (autograd differentiates them too, :26-28) */
int amds_softmax_rows_bwd(const float* p, float* dp, long rows, int cols, void* stream);
int amds_landmark_mean_bwd(const float* dout, float* dx, long sxo, long sxi, int ld, int outer, int inner, int m, int l, int d,
                           float scale, int accumulate, void* stream);
size_t amds_dwconv_seq_wgrad_workspace_bytes(int outer, int inner, int taps);
int amds_dwconv_seq_wgrad(const float* dout, long soo, long soi, int ldo, const float* v, long svo, long svi, int ldv, float* dw,
                          int outer, int inner, int n, int d, int taps, void* ws, size_t ws_bytes, void* stream);
size_t amds_ppeg_wgrad_workspace_bytes(int B, int C);
int amds_ppeg_wgrad(const float* x, const float* dy, float* dcorr, int B, int H, int W, int C, void* ws, size_t ws_bytes, void* stream);
int amds_relu_bwd(const float* h, const float* dh, float* dz, long n, void* stream);
size_t amds_pinv_init_bwd_workspace_bytes(int nmat);
int amds_pinv_init_bwd(const float* x, const float* dz0, float* dx, int nmat, int n, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MIL training step (reference src/stamp/modeling/models/__init__.py:133-141, 239-279): bf16 MFMA operands,
 * fp32 accumulate / residual / master weights / optimizer state
 * ---------------------------------------------------------------------------------------------- */

/* Batched / split-K amds_gemm on the production kernel: batch b uses A + b*bsA, W + b*bsW, out + b*bsOut (elements).
 * Weight gradients dW[N][K] = dy^T x contract over the token dimension: split it (bsA = bsW = chunk, K = chunk) into
 * nbatch fp32 partials and sum them with amds_colsum.  epi: AMDS_EPI_BIAS or AMDS_EPI_BIAS_F32. */
int amds_gemm_batched(const void* A, long lda, long bsA, const void* W, long ldw, long bsW, int M, int N, int K,
                      int nbatch, int dtype, int epi, void* out, long ldo, long bsOut, const float* bias,
                      float acc_scale, void* stream);

/* Weight-gradient partials straight from TOKEN-major operands (what autograd derives for `nn.Linear`: dW = dy^T x; reference
 * src/stamp/modeling/models/__init__.py:239-279 -> loss.backward()): part[s][n][k] = sum over the tokens t of split s of dy[t][n] * x[t][k], fp32,
 * s < split_k, each split = ceil(tokens / (64 split_k)) * 64 consecutive tokens (rows past `tokens` count as zeros).  dy [tokens][ld_dy], x [tokens][ld_x]
 * 16-bit (dtype); N, K multiples of 256.  No transposed or padded copies of the operands: the kernel (id 15, csrc/gemm_4w16.h) reads 64-token row tiles
 * into LDS and builds its MFMA fragments with gfx950's transpose read.  Sum the partials with amds_colsum(part, N*K, out, split_k, N*K, AMDS_F32, ...). */
int amds_wgrad_tn(const void* dy, long ld_dy, const void* x, long ld_x, long tokens, int N, int K, int split_k, int dtype, float* part, void* stream);
/* dst[c][r] = src[r][c] for 16-bit elements (dst leading dimension ld_dst >= R). */
int amds_transpose16(const void* src, long ld_src, void* dst, long ld_dst, int R, int C, void* stream);
/* The 16-bit operand refresh behind an optimiser step, ALL matrices in one launch (the MIL `vit` head kept ~20 amds_cast_pad / amds_transpose16 launches
 * per step for it): for every entry, dst[r][c] = (dtype) src[r][c] and -- when dst_t is not NULL -- dst_t[c][r] = the same value, rows x cols fp32 -> 16-bit,
 * both multiples of 64, leading dimensions in elements, 16-byte aligned.  At most 32 entries per call.  Reference step: the optimiser behind
 * `LitMilClassificationMixin` (src/stamp/modeling/models/__init__.py:239-279); the copies exist because the MFMA GEMMs read 16-bit operands. */
typedef struct amds_cast_entry {
    const float* src; long ld_src;
    void* dst; long ld_dst;
    void* dst_t; long ld_dst_t;
    int rows, cols, dtype;          /* AMDS_F16 or AMDS_BF16 */
} amds_cast_entry;
int amds_cast_transpose_multi(const amds_cast_entry* entries_host, int n, void* stream);
/* out[n] (+)= sum_m x[m][n]; deterministic two-stage reduction (bias gradients, split-K partial sums). */
size_t amds_colsum_workspace_bytes(int M, int N);
int amds_colsum(const void* x, long ld, float* out, int M, int N, int in_dtype, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* out_i[e] = sum over s < rows of part_i[s * count_i + e], e < count_i, for up to 32 (part, out, count) triples in ONE launch, with the association of
 * amds_colsum(part_i, count_i, out_i, rows, count_i, AMDS_F32, ...) (same bits): the split-K partials of every weight gradient of a backward pass
 * (amds_wgrad_tn) summed behind the last of them instead of one reduction launch per matrix.  counts: multiples of 4; pointers 16-byte aligned. */
int amds_sum_partials_multi(const float* const* parts_host, float* const* outs_host, const long* counts_host, int n, int rows, void* stream);
/* Column sums a backward pass can postpone, summed by ONE launch behind the last of them (the MIL `vit` step kept 21 small reduction launches for
 * LayerNorm parameter gradients, bias gradients' second stages, the head bias and the class token; reference step: `loss.backward()` of
 * src/stamp/modeling/models/__init__.py:239-279).  kind 0: out[n] = sum over r < rows of x[r * ld + n], fp32, rows <= 2048 -- the bits of
 * amds_colsum(x, ld, out, rows, cols, AMDS_F32, 0, ...).  kind 1: out[n] = the second stage of amds_colsum over `rows` chunk partials
 * x[c * cols + n] written by amds_colsum_partials (which reports the chunk count) -- amds_colsum_partials + a kind-1 entry = amds_colsum, same bits.
 * At most 32 entries per call. */
typedef struct amds_colsum_entry {
    const float* x;
    float* out;
    long ld;                 /* kind 0: row stride of x in elements (>= cols); kind 1: ignored */
    int rows, cols, kind;
} amds_colsum_entry;
int amds_colsum_partials(const void* x, long ld, float* part, int M, int N, int in_dtype, int* nchunk_out, void* stream);
int amds_colsum_multi(const amds_colsum_entry* entries_host, int n, void* stream);
/* LayerNorm forward that also stores mean / rstd per row (fp32), and its backward:
 *   dx = (add_skip ? dx : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;  dgamma (+)= sum dy*xhat; dbeta (+)= sum dy */
int amds_layernorm_train(const float* x, long x_row_stride, const float* gamma, const float* beta, void* y, long y_row_stride,
                         float* mean, float* rstd, int rows, int cols, float eps, int out_dtype, void* stream);
/* Same, and the rows x[.., 0 .. copy_cols) also go out unchanged into x_copy (row stride copy_row_stride): the training forward's
 * `x_mid = x_in; x_mid += out_proj(attention(...))` (reference vision_tranformer.py:291-293: `x = x + ...` keeps the block input alive for autograd)
 * without a device-to-device copy launch.  x_copy may be NULL (= amds_layernorm_train). */
int amds_layernorm_train_copy(const float* x, long x_row_stride, const float* gamma, const float* beta, void* y, long y_row_stride,
                              float* mean, float* rstd, int rows, int cols, float eps, int out_dtype, float* x_copy, long copy_row_stride,
                              int copy_cols, void* stream);
size_t amds_layernorm_bwd_workspace_bytes(int rows, int cols);
int amds_layernorm_bwd(const float* dy, long dy_stride, const float* x, long x_stride, const float* mean, const float* rstd,
                       const float* gamma, float* dx, long dx_stride, int add_skip, float* dgamma, float* dbeta, int accumulate_params,
                       int rows, int cols, void* ws, size_t ws_bytes, void* stream);
/* Same, and dx (after the skip add) also goes out as bf16 rows into dx_bf16 -- the operand of the GEMMs that take dx next -- multiplied by the mask and
 * scale of dropout site (seed, stream_id) at rate p when p > 0 (the bits of amds_dropout_cast_bwd over a [rows][cols] tensor): the backward of
 * `x = x + Dropout(...)` / of a plain residual add without a separate cast pass over dx.  dx_bf16 may be NULL (= amds_layernorm_bwd). */
int amds_layernorm_bwd_cast(const float* dy, long dy_stride, const float* x, long x_stride, const float* mean, const float* rstd,
                            const float* gamma, float* dx, long dx_stride, int add_skip, float* dgamma, float* dbeta, int accumulate_params,
                            int rows, int cols, void* ws, size_t ws_bytes, void* dx_bf16, long dx_bf16_stride, float p, uint64_t seed,
                            uint32_t stream_id, void* stream);
/* amds_layernorm_bwd_cast without its two parameter-gradient reductions: the per-64-row partials go to dgamma_part / dbeta_part, fp32
 * [ceil(rows / 64)][cols] each, for the caller to sum when it likes (kind-0 entries of amds_colsum_multi: ceil(rows / 64) rows of `cols`). */
int amds_layernorm_bwd_partials(const float* dy, long dy_stride, const float* x, long x_stride, const float* mean, const float* rstd,
                                const float* gamma, float* dx, long dx_stride, int add_skip, float* dgamma_part, float* dbeta_part,
                                int rows, int cols, void* dx_bf16, long dx_bf16_stride, float p, uint64_t seed, uint32_t stream_id, void* stream);
/* exact-erf GELU on a stored pre-activation and its derivative (dz = du * gelu'(z)). */
int amds_gelu_fwd(const void* z, void* u, long n, int in_dtype, int out_dtype, void* stream);
int amds_gelu_bwd(const void* z, const void* du, void* dz, long n, int z_dtype, int du_dtype, int dz_dtype, void* stream);
/* Attention forward that also stores the log2-domain log-sum-exp per query, lse fp32 [B][H][T], and the flash-style
 * backward: dqkv (same layout as qkv) from qkv, out, dout; dq_sum_ws: fp32 [B][H][T] scratch. */
int amds_attention_fwd_lse(const void* qkv, void* out, float* lse, int B, int T, int H, int dtype, void* stream);
int amds_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dq_sum_ws, void* dqkv,
                       int B, int T, int H, int dtype, void* stream);

/* ALiBi attention in TRAIN mode (reference MultiHeadALiBi / _ALiBi, src/stamp/modeling/models/vision_tranformer.py:34-74, with the
 * running-mean scaler already updated by the caller, :24-29):  out = softmax(q k^T/8) v - bias_scale_h * U,
 * U = sum_k cdist(c_q, c_k) * inv_running_mean_h * v.  out, U and Osm (the softmax part alone): bf16 [B*T][H*64];
 * lse fp32 [B][H][T] (log2 domain). */
int amds_attention_alibi_fwd_train(const void* qkv, const float* coords, const float* inv_running_mean, const float* bias_scale,
                                   void* out_bf16, void* u_bf16, void* osm_bf16, float* lse, int B, int T, int H, int dtype, void* stream);
/* Its backward (bf16 operands): dqkv as amds_attention_bwd, the value path carries the distance term (dV += (P - dist_scale_h D)^T dO,
 * dist_scale_h = bias_scale_h * inv_running_mean_h), and dbs_part [B][H][T] holds -sum_d dO U per query: summed over (b, q) it is
 * the gradient of bias_scale_h.  dq_sum_ws: fp32 [B][H][T] scratch. */
int amds_attention_alibi_bwd(const void* qkv, const void* osm, const void* u, const void* dout, const float* lse, const float* coords,
                             const float* bias_scale, const float* dist_scale, float* dq_sum_ws, float* dbs_part, void* dqkv,
                             int B, int T, int H, void* stream);
/* Training forward / backward of nn.MultiheadAttention WITH its dropout on the attention probabilities (the `dropout` constructor
 * argument of the reference's VisionTransformer reaches nn.MultiheadAttention, vision_tranformer.py:191; active in train mode):
 * out = drop(P) v, lse that of the undropped softmax.  The keep mask is a counter-based function of (seed, stream_id, b, h, q, k),
 * regenerated by the backward (same p / seed / stream_id), never stored.  P(drop) = round(p * 65536) / 65536.  p = 0 is
 * amds_attention_fwd_lse / amds_attention_bwd. */
int amds_attention_fwd_train(const void* qkv, void* out, float* lse, int B, int T, int H, int dtype, float p, uint64_t seed,
                             uint32_t stream_id, void* stream);
int amds_attention_bwd_train(const void* qkv, const void* out, const void* dout, const float* lse, float* dq_sum_ws, void* dqkv,
                             int B, int T, int H, int dtype, float p, uint64_t seed, uint32_t stream_id, void* stream);
/* The other dropout sites of the reference's MIL `vit` (vision_tranformer.py:157-169, 314-318), same counter-based masks:
 *   amds_gelu_dropout_fwd   u = drop(gelu_erf(z))              `Linear -> GELU -> Dropout` of project_features / feed_forward
 *   amds_gelu_dropout_bwd   dz = gelu'(z) * drop'(du)
 *   amds_dropout_add        x_out[r][c] = x_in[r][c] + drop(y[r][c])     feed_forward's trailing Dropout + the residual add
 *   amds_dropout_cast_bwd   dy[r][c] = (out dtype) drop'(dx[r][c])       gradient entering the last Linear's backward
 * Element index of [rows][cols] views = r * cols + c (row pitches ld* may be larger than cols).
 * amds_dropout_keep_scale(p) = 65536 / (65536 - round(p*65536)), the factor kept values are multiplied by.
 * amds_dropout_mask / amds_attention_dropout_mask write the keep masks themselves (u8 0/1; [n] resp. [B][H][T][T]) for tests. */
float amds_dropout_keep_scale(float p);
int amds_gelu_dropout_fwd(const void* z, void* u, long n, int in_dtype, int out_dtype, float p, uint64_t seed, uint32_t stream_id, void* stream);
int amds_gelu_dropout_bwd(const void* z, const void* du, void* dz, long n, int z_dtype, int du_dtype, int dz_dtype, float p, uint64_t seed,
                          uint32_t stream_id, void* stream);
int amds_dropout_add(const float* y, long ldy, const float* x_in, long ldx, float* x_out, long ldo, long rows, int cols, float p,
                     uint64_t seed, uint32_t stream_id, void* stream);
int amds_dropout_cast_bwd(const float* dx, long ldx, void* dy, long ldy, long rows, int cols, int out_dtype, float p, uint64_t seed,
                          uint32_t stream_id, void* stream);
/* The four element-wise dropout sites on `rows` pitched rows whose LOGICAL row index is r * row_mul (the class rows b * S of the MIL `vit` head's last block, of which
 * nothing else is read): the mask of element (r, c) is the one the functions above draw for element (r * row_mul, c) of the full tensor.  bf16 tensors; p = 0: no mask. */
int amds_gelu_dropout_fwd_rows(const void* z, long ldz, void* u, long ldu, long rows, int cols, long row_mul, float p, uint64_t seed, uint32_t stream_id, void* stream);
int amds_gelu_dropout_bwd_rows(const void* z, long ldz, const void* du, long ldu, void* dz, long lddz, long rows, int cols, long row_mul, float p, uint64_t seed,
                               uint32_t stream_id, void* stream);
int amds_dropout_add_rows(const float* y, long ldy, const float* x_in, long ldx, float* x_out, long ldo, long rows, int cols, long row_mul, float p, uint64_t seed,
                          uint32_t stream_id, void* stream);
int amds_dropout_cast_bwd_rows(const float* dx, long ldx, void* dy, long ldy, long rows, int cols, long row_mul, float p, uint64_t seed, uint32_t stream_id, void* stream);
int amds_dropout_mask(uint8_t* mask, long n, float p, uint64_t seed, uint32_t stream_id, void* stream);
int amds_attention_dropout_mask(uint8_t* mask, int B, int H, int T, float p, uint64_t seed, uint32_t stream_id, void* stream);
/* rowsum[b*T + q] = sum_k |coords[b,q] - coords[b,k]|: the batch statistic `_RunningMeanScaler` needs (mean of torch.cdist). */
int amds_cdist_rowsum(const float* coords, float* rowsum, int B, int T, void* stream);
/* fp16 -> bf16 (features are fp16 on disk; the training path feeds bf16 MFMA operands). */
int amds_convert_f16_bf16(const void* src, void* dst, long n, void* stream);
/* torch.optim.AdamW step (amsgrad=False) on flat fp32 buffers; `step` is the 1-based step count (bias correction). */
int amds_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMDSTAMP_H */
