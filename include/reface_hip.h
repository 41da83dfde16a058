/*
 * reface_hip.h -- C-ABI of the MI355X (gfx950) REFace inference kernels.
 *
 * The reference (Sanoojan/REFace) is 100 % Python: its "operator API" for the hot path is the
 * set of ATen calls issued by ldm/modules/diffusionmodules/openaimodel.py, ldm/modules/attention.py,
 * ldm/modules/diffusionmodules/model.py and ldm/models/diffusion/ddim.py.
 * Each entry point below replaces one family of those calls; the reference line it stands in for
 * is cited next to it.  The Python host layer (the modules of reface_amd) binds these with ctypes; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes; no torch types.  All pointers are DEVICE pointers, 16-byte aligned.
 *   - the caller owns every buffer; kernels never allocate.
 *   - `stream` is a hipStream_t passed as void* (the caller's current PyTorch HIP stream).
 *   - every function returns 0 on success, non-zero on error; rf_last_error() gives the message.
 *   - activations are channels-last: a tensor [B, H, W, C] is a row-major matrix [B*H*W, C].
 *   - dtype codes: RF_F32 = 0 (fp32 storage, exact-fp32 MFMA), RF_BF16 = 1 (bf16 storage, fp32 accumulate), RF_F16 = 4 (IEEE fp16 storage on
 *     v_mfma_f32_32x32x16_f16, fp32 accumulate: the bf16 mode's kernels at the same matrix-core rate with 11 significant bits instead of 8 --
 *     the "fp16" throughput mode; accepted wherever RF_BF16 is unless an entry point says otherwise, never mixed with bf16 inside one call).
 */
#ifndef REFACE_HIP_H
#define REFACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RF_F32 = 0, RF_BF16 = 1, RF_FP8_E4M3 = 2 /* OCP e4m3fn weights (rf_conv_gemm_desc.w_dtype only) */,
       RF_BF16X3 = 3 /* split-bf16 operand pairs, see rf_conv_gemm_desc.dtype and rf_split_bf16 */,
       RF_F16 = 4 /* IEEE binary16 storage (the "fp16" throughput mode) */ };

/* epilogue activations of rf_conv_gemm */
enum { RF_ACT_NONE = 0, RF_ACT_GEGLU = 1, RF_ACT_SILU = 2, RF_ACT_QUICK_GELU = 3, RF_ACT_GELU = 4,
       RF_ACT_RELU = 5, RF_ACT_SIGMOID = 6, RF_ACT_PRELU = 7 };

const char* rf_last_error(void);
int rf_version(void);

/*
 * rf_conv_gemm -- implicit-GEMM convolution / linear layer on the matrix cores:
 *
 *     out[m, n] = act( alpha * sum_k A[m, k] * W[n, k] + bias[n] + rowvec[sample(m), n] ) + residual[m, n]
 *
 * A is never materialised: row m is output pixel (b, oy, ox); k = (ky*KW + kx) * (C0 + C1) + c
 * addresses input pixel (oy*stride - pad_t + ky, ox*stride - pad_l + kx) of a channels-last source,
 * optionally nearest-upsampled x2 on the fly (`ups`) and optionally the channel concatenation of two
 * sources [src0 | src1] (UNet skip connections).  KH = KW = 1 with B*Hin*Win = M gives a plain GEMM
 * (nn.Linear / 1x1 conv); `batch` > 1 runs independent problems (strides in elements).
 *
 * Replaces: F.conv2d 3x3 / 1x1 in ResBlock, Downsample, Upsample, UNet in/out convs
 * (openaimodel.py:107,116,151-153,204,230,241,669,835), nn.Linear in attention / GEGLU / time-embed
 * (attention.py:40,60,159-170; openaimodel.py:218-224,632-636), torch.cat([h, hs.pop()])
 * (openaimodel.py:898), F.interpolate nearest (openaimodel.py:116), the VAE convs
 * (model.py:60-79,97-121,155-174) and bmm in the VAE AttnBlock (model.py:186-198).
 */
typedef struct rf_conv_gemm_desc {
    /* dtype RF_BF16X3: every fp32 operand value x is stored as the bf16 pair hi = bf16(x), lo = bf16(x - hi) -- a src0 pixel is */
    /* [C0 hi | C0 lo] (ld0 >= 2 C0), a W row holds per 64-element K tile [64 hi | 64 lo | 64 hi] (3 K bf16; ldw >= 3 K) -- and a product is */
    /* accumulated in fp32 as hi hi + hi lo + lo hi on the bf16 MFMA (relative error 2^-16 per product; out_dtype RF_F32, one source, */
    /* K and C0 multiples of 64, K / C0 / ld0 given in REAL elements): the fast form of the fp32 VAE convolutions (model.py:60-121). */
    int32_t dtype;        /* RF_F32 | RF_BF16 | RF_F16: element type of src0/src1/W; RF_BF16X3: split-bf16 pairs (above) */
    int32_t out_dtype;    /* element type of out and residual */
    int32_t M, N, K;      /* GEMM view; K = KH*KW*(C0+C1) (may be padded up to a multiple of 8) */
    const void* src0;
    const void* src1;     /* optional second source (channel concat), NULL if unused */
    int32_t C0, C1;       /* channels taken from src0 / src1 */
    int32_t ld0, ld1;     /* pixel pitch (elements) of src0 / src1 */
    int32_t Hin, Win;     /* spatial size of the stored source (before `ups`) */
    int32_t Hout, Wout;   /* spatial size of the output; M = B*Hout*Wout */
    int32_t KH, KW, stride, pad_t, pad_l, ups;
    const void* W;        /* [N][K] row-major with row pitch ldw */
    int32_t ldw;          /* row pitch of W in elements (0 = K) */
    const float* bias;    /* [N] fp32 or NULL */
    const float* rowvec;  /* [M / rows_per_sample][ldv] fp32 or NULL */
    int32_t rows_per_sample, ldv;
    const void* residual; /* [M][ldr] (out_dtype) or NULL */
    int32_t ldr;
    int32_t act;          /* RF_ACT_* ; GEGLU: W rows interleaved in blocks of 32 (value|gate), out has N/2 columns */
    void* out;
    int32_t ldo;
    float alpha;
    int32_t batch;        /* >= 1 */
    int64_t sA, sW, sO, sR;   /* batch strides (elements) of src0, W, out, residual */
    const float* act_vec; /* [N] fp32 per-column activation parameter (PReLU slopes) or NULL */
    int32_t korder;       /* 0: k = tap*(C0+C1) + c;  1: k = ((c / BK)*KH*KW + tap)*BK + c % BK, BK = 64 (bf16) / 32 (fp32) */
    void* workspace;      /* optional fp32 scratch for split-K (small-M / long-K problems that cannot fill 256 CUs); NULL = never split */
    int64_t workspace_bytes;
    int32_t gn_rows;      /* > 0: also emit GroupNorm(32) partial sums of `out` (rows per sample = H*W) for up to two consumers, see below */
    double* gn_part0;     /* consumer 0: [M / gn_rows][gn_nchunks0][32][2] fp64 (sum, sumsq) in the rf_groupnorm_stats layout, or NULL */
    int32_t gn_cpg0, gn_coff0, gn_slot0, gn_nchunks0;   /* its channels per group, position of out column 0 in its channel space, first chunk slot, chunk slots per sample */
    double* gn_part1;     /* consumer 1 (e.g. the decoder norm over [h | skip], which groups the same channels differently), or NULL */
    int32_t gn_cpg1, gn_coff1, gn_slot1, gn_nchunks1;
    /* fp8 weight path (BASELINE configs[4]): w_dtype = RF_FP8_E4M3 with dtype = RF_BF16 -- W holds OCP e4m3fn bytes, [N][ldw] with */
    /* ldw (BYTES) a multiple of 128 and every row zero-padded to it, dequantised on the way from LDS to the bf16 MFMA as */
    /* w = fp8 * wscale[n], wscale[n] a power of two (rf_quantize_fp8_rows produces both).  0 = W has the element type `dtype`. */
    /* Replaces the same nn.Conv2d / nn.Linear weights (openaimodel.py:204,230,241; attention.py:40,60,159-170), half the bytes. */
    int32_t w_dtype;
    const float* wscale;  /* [N] fp32, powers of two */
    /* fp8 x fp8 path (the fp8 matrix pipe, v_mfma_scale_f32_32x32x64_f8f6f4): dtype = w_dtype = RF_FP8_E4M3.  src0 holds e4m3fn activation bytes */
    /* (ld0 / C0 / K in bytes = elements) and ascale one E8M0 byte (scale 2^(e - 127)) per (pixel, 32-channel block) with pixel pitch as_ld (a */
    /* multiple of 4, pad bytes = a valid code), both written by rf_quantize_fp8_act or by rf_groupnorm_apply / rf_layernorm with out_dtype = */
    /* RF_FP8_E4M3; W as in the w_dtype path with every filter tap's channel run zero-padded to a multiple of 128 (K counts the padded run; a */
    /* k x k window needs C0 % 128 == 0 -- a 1 x 1 / linear layer may over-read up to 127 bytes of the next pixel against zero weights). */
    /* out_dtype RF_BF16.  Replaces the same nn.Conv2d / nn.Linear calls with both operands in fp8 (BASELINE configs[4]). */
    const void* ascale;
    int32_t as_ld;
    /* out_dtype = RF_FP8_E4M3 (fp8 x fp8 path with act = GEGLU only): out receives the N/2 gated values as e4m3fn bytes (ldo in bytes) and oscale */
    /* one E8M0 byte per (row, 32 output columns), row pitch os_ld -- the A operand of the following ff.net.2 GEMM (attention.py:60). */
    void* oscale;
    int32_t os_ld;
    /* LayerNorm folded around two bf16 GEMMs (attention.py:231-233, 239-243: norm1 in front of to_q / to_k / to_v, norm3 in front of ff.net.0). */
    /* PRODUCER of the tensor to be normalised (direct epilogue, no split-K): ln_stats_out [M][ln_out_parts][2] fp32 receives, per row and per */
    /* column stripe of its wave tile (rf_conv_gemm_plan2's wave_cols; ln_out_parts = N / wave_cols), the (mean, M2 = sum of squared deviations) */
    /* of the values as stored.  CONSUMER (A = that tensor, un-normalised): out = act(rstd[m] * (alpha * acc - mean[m] * ln_u[n]) + bias[n]) with */
    /* mean / rstd of row m combined from ln_stats_in [M][ln_in_parts][2] (ln_in_cols columns per part, eps = ln_eps); the caller folds gamma */
    /* into W's columns, passes ln_u[n] = sum_k W[n, k] (of the folded, rounded W) and W beta (+ the layer's bias) as `bias`. */
    /* Replaces the nn.LayerNorm pass between the two GEMMs (one read + one write of [M, C] and a launch). */
    void* ln_stats_out;
    int32_t ln_out_parts;
    const void* ln_stats_in;
    int32_t ln_in_parts, ln_in_cols;
    float ln_eps;
    const float* ln_u;
    /* Per-sample weights (plain GEMM, dtype RF_BF16 / RF_F32, no w_dtype): rows [s * rows_per_sample, (s + 1) * rows_per_sample) multiply */
    /* W + s * w_sample_stride (elements); 0 = one W for all rows.  rows_per_sample must be a multiple of the tile rows (rf_conv_gemm_plan2's BM). */
    /* What rf_groupnorm_fold_linear writes: a GroupNorm folded into the weights of the Linear / 1x1 conv behind it. */
    int64_t w_sample_stride;
} rf_conv_gemm_desc;

int rf_conv_gemm(const rf_conv_gemm_desc* d, void* stream);

/* The epilogue tiling rf_conv_gemm would use for `d` -- rows / columns of the block that finishes an output tile (the GEMM tile, or
 * the tile of the split-K reduce pass when splitk > 1) -- and the split-K factor; no launch.  Fused GroupNorm statistics
 * (gn_rows > 0) need gn_rows % bm == 0; a launch then writes (gn_rows / bm) * ceil(N / bn) chunk slots per sample,
 * slot = gn_slot + (row tile within the sample) * ceil(N / bn) + column tile.  Replaces the separate statistics pass of
 * nn.GroupNorm (util.py:214-216) over a tensor this GEMM has just produced. */
int rf_conv_gemm_plan(const rf_conv_gemm_desc* d, int32_t* bm, int32_t* bn, int32_t* splitk);
/* The same query with the whole tile plan: info8 = {statistics rows, statistics cols, splitk, BM, BN, wave_cols (columns of one wave's tile),
 * epilogue form (1 = direct register -> global, 0 = staged), flags: bit 0 = split-K through fragment slabs, bit 1 = the call runs as TWO GEMM
 * kernels (a 256-wide launch whose last round of tiles is 20-60 % full is split along N; the query then also validates the second part)}. */
int rf_conv_gemm_plan2(const rf_conv_gemm_desc* d, int32_t* info8);

/* Fused transformer feed-forward at C = 320 (the 64x64 level):  out = (GEGLU(x W1^T + b1)) W2^T + b2 + residual, bf16 in / out, fp32
 * accumulate, GELU on the 16-bit modes' sigmoid form (degree-5 argument, <= 2.6e-5 from the erf form: csrc/common.h).  The [M, 4C] hidden tensor stays in registers (tokens on lanes, see ffn.hip).
 *   w1p / b1p : ff.net.0.proj [8C, C] / [8C] with rows interleaved in blocks of 32 (value | gate), as rf_conv_gemm's GEGLU takes them
 *   w2q       : ff.net.2 [C, 4C] with the columns of every 16-group in the order 0-3, 8-11, 4-7, 12-15
 *   ln_eps > 0: every row of x is LayerNorm-ed first, in registers (two-pass fp32 statistics, normalised values rounded to bf16 as
 *               rf_layernorm stores them; NO affine: the host folds norm3's gamma into w1p's columns and W1 beta into b1p) -- the kernel then
 *               also replaces `self.norm3` (attention.py:231-233, 243) and x is the un-normalised residual stream.
 * Replaces FeedForward.forward (attention.py:40-76) + the residual add of BasicTransformerBlock (attention.py:243). */
/* (bf16 only: the fp16 mode goes through rf_ffn_block with wpo = NULL and dtype = RF_F16) */
int rf_ffn_geglu(const void* x, int ldx, const void* w1p, const float* b1p, const void* w2q, const float* b2, const void* residual, int ldr,
                 void* out, int ldo, int M, int C, float ln_eps, void* stream);

/* The same kernel with SpatialTransformer.proj_out behind the feed-forward (attention.py:268-272, 288-289) -- the token-resident tail of the block:
 *   out = ((GEGLU(LN?(x) W1^T + b1)) W2^T + b2 + residual) Wpo^T + bpo + res2[row % res2_rows]
 * The feed-forward's output row (rounded to bf16 where the unfused path stores it) never leaves the CU: it is re-laid-out in registers as the B operand
 * of one more MFMA contraction against wpo [C][C] (plain rows, bf16), streamed through the W2 buffers.  res2 = the transformer's input x_in
 * (`return x + x_in`); res2_rows > 0: the residual has that many rows and is shared by the batch halves (the CFG-shared first block).  gn_*: the
 * GroupNorm(32) partial sums of `out` for up to two consumers, as rf_conv_gemm_desc.gn_* (one chunk slot per 128-token block: slot = gn_slot + block
 * within the sample; gn_rows = H*W a multiple of 128).  wpo = NULL: exactly rf_ffn_geglu.  Replaces FeedForward.forward + the residual add +
 * proj_out + `x + x_in` -- and the statistics pass of the GroupNorm that reads the block's output. */
typedef struct rf_ffn_desc {
    const void* x; int32_t ldx;
    const void* w1p; const float* b1p;
    const void* w2q; const float* b2;
    const void* residual; int32_t ldr;
    void* out; int32_t ldo;
    int32_t M, C;
    float ln_eps;
    const void* wpo; const float* bpo;
    const void* res2; int32_t ldr2, res2_rows;
    int32_t gn_rows;
    double* gn_part0;
    int32_t gn_cpg0, gn_coff0, gn_slot0, gn_nchunks0;
    double* gn_part1;
    int32_t gn_cpg1, gn_coff1, gn_slot1, gn_nchunks1;
    int32_t dtype;        /* element type of x / w1p / w2q / residual / out / wpo / res2: RF_BF16 (also 0: the descriptor as it was before the field) or RF_F16 */
    /* attn1.to_out IN FRONT of the feed-forward (attention.py:239-243; needs wpo): wo != NULL makes `x` the attention's output [front_rows or M][ldx] and the block first forms */
    /*   x1 = x Wo^T + bo + ctx[row / rows_per_sample0] + res0[row % front_rows]        (out-projection + the sample's cross-attention vector + the residual tok)             */
    /* rounds it to `dtype`, stores it to x1 -- which must be the same buffer as `residual` -- and continues with it as the feed-forward's input.  front_rows > 0: x / res0  */
    /* have that many rows and are shared by the batch halves (the CFG-shared first block); 0: M rows.  Replaces that rf_conv_gemm launch and the re-read of its output.    */
    const void* wo; const float* bo;
    const float* ctx; int32_t ldc, rows_per_sample0;
    const void* res0; int32_t ldr0, front_rows;
    void* x1; int32_t ldx1;
} rf_ffn_desc;
int rf_ffn_block(const rf_ffn_desc* d, void* stream);

/* The token-resident FRONT of a SpatialTransformer block at C = 320 (attention.py:262-266 `norm`, 276-279 `proj_in`, 231-233 `norm1`, 239 + 159-170 to_q / to_k / to_v)
 * in one kernel (round 6):
 *   tok = x W'_s^T + r_s                 W'_s / r_s: proj_in with the GroupNorm folded in per sample (rf_groupnorm_fold_linear's w_out / rowvec_out)
 *   qkv = LayerNorm(tok) Wqkv'^T + b'    LayerNorm without affine in registers (two-pass fp32 statistics of the STORED tok row, normalised values rounded to `dtype`);
 *                                        the host folds norm1's gamma into wqkv's columns and Wqkv beta into bqkv
 * x [M][ldx] un-normalised, tok [M][ldt] (written: the attention's out-projection reads it back as its residual), qkv [M][ldq] columns [0, 3C), all in `dtype`
 * (RF_BF16, also 0, or RF_F16); wpi [S][C][C] with w_sample_stride elements between samples (0: one matrix), rowvec fp32 [S][ldv], rows_per_sample a multiple of 128
 * dividing M; wqkv [3C][C], bqkv fp32 [3C].  Replaces two rf_conv_gemm launches whose K = 320 main loops are a quarter of their time, the re-read of tok and the
 * LayerNorm statistics exchange between them. */
typedef struct rf_attn_in_desc {
    const void* x; int32_t ldx;
    const void* wpi; int64_t w_sample_stride;
    const float* rowvec; int32_t ldv, rows_per_sample;
    void* tok; int32_t ldt;
    const void* wqkv; const float* bqkv;
    void* qkv; int32_t ldq;
    int32_t M, C;
    float ln_eps;
    int32_t dtype;
} rf_attn_in_desc;
int rf_attn_in(const rf_attn_in_desc* d, void* stream);

/* Per-row fp8 quantisation of a weight matrix for the w_dtype = RF_FP8_E4M3 path: w [N][K] fp32 (row pitch K) ->
 * q [N][ldq] e4m3fn bytes (ldq >= K, a multiple of 128; the pad bytes are written as zero) and scale [N] = the smallest power of
 * two with max|w[n,:]| / scale <= 448 (e4m3fn's largest finite value); q = round-to-nearest-even(w / scale), saturating.
 * Done once at engine build (torch's `.to(float8_e4m3fn)` is the CPU reference in the tests). */
int rf_quantize_fp8_rows(const float* w, int N, int K, int ldq, void* q, float* scale, void* stream);

/*
 * GroupNorm(32 groups) over channels-last [B, HW, C]  (+ optional SiLU), two launches:
 *   rf_groupnorm_stats : per (b, chunk, group) partial (sum, sumsq) in fp64 -> `partial`
 *                        partial must hold B * nchunks * 32 * 2 doubles.
 *   rf_groupnorm_apply : y = (x - mean) * rstd * gamma + beta, optional SiLU.
 * Replaces nn.GroupNorm / GroupNorm32 + nn.SiLU (util.py:214-216, attention.py:76-77,
 * model.py:33-39; openaimodel.py:201-203,225-227,832-834).
 */
int rf_groupnorm_stats(int dtype, const void* x, int B, int HW, int C, int ldx, int nchunks, double* partial, void* stream);
/* [B][nchunks][32][2] -> [B][1][32][2] in a fixed summation order: compacts the many chunk slots that GEMM-fused statistics produce
 * on large images (one slot per output tile) so that rf_groupnorm_apply reads one slot per sample. */
int rf_groupnorm_finalize(const double* partial_in, int B, int nchunks, double* partial_out, void* stream);
int rf_groupnorm_apply(int dtype, const void* x, int B, int HW, int C, int ldx, int nchunks, const double* partial,
                       const float* gamma, const float* beta, float eps, int silu, int out_dtype, void* out, int ldo, void* stream);
/* GroupNorm(32) folded into the Linear / 1x1 conv that follows it (SpatialTransformer: `norm` then `proj_in`, attention.py:262-266 / 276-279):
 *   Linear(GN(x))[m, n] = sum_k W'_s[n, k] x[m, k] + r_s[n]        for the rows m of sample s, with
 *   W'_s[n, k] = W[n, k] rstd[s, g(k)] gamma[k]    (rounded to out_dtype)        r_s[n] = bias[n] + sum_k W[n, k] beta[k] - sum_k W'_s[n, k] mean[s, g(k)]
 * (mean / rstd from the same `partial` records rf_groupnorm_apply reads; the mean's term uses the ROUNDED W' so that it cancels exactly what
 * the matrix pipe accumulates).  W fp32 [N][C]; w_out [B][N][C] (rf_conv_gemm_desc.w_sample_stride = N * C), rowvec_out fp32 [B][N] (the GEMM's
 * per-sample vector, no bias).  Replaces the rf_groupnorm_apply pass over [B, HW, C] in front of that Linear: a read + a write of the tensor. */
int rf_groupnorm_fold_linear(const float* W, int N, int C, int B, int HW, int nchunks, const double* partial, const float* gamma, const float* beta,
                             const float* bias, float eps, int out_dtype, void* w_out, float* rowvec_out, void* stream);

/* GroupNorm(32) + SiLU + 3x3 convolution (stride 1, pad 1) to No <= 4 output channels in one pass over the RAW tensor -- the UNet's `out` head
 * (openaimodel.py:737-741: normalization(ch), nn.SiLU(), conv_nd(dims, model_channels, out_channels, 3, padding=1)): x [B][H*W][ldx] un-normalised in
 * `dtype` (RF_BF16 | RF_F16), `partial` its GroupNorm partial sums (the records rf_groupnorm_apply reads), W [No][9 C] in `dtype` (k = tap * C + c),
 * out [B*H*W][ldo] fp32 or `dtype`.
 * Kernel 1 normalises + activates every pixel's C values in registers (rounded to bf16 as the apply pass stores them) and multiplies them with
 * all nine taps' weights on the matrix pipe (per-tap partial products, fp32, into `workspace`: B*H*W * 160 bytes); kernel 2 sums, per output pixel,
 * the taps whose source pixel lies inside the image.  Replaces rf_groupnorm_apply + rf_conv_gemm for this layer: one read of the tensor instead of a
 * write + nine tap-shifted reads.  C in {320, 128, 64}. */
int rf_gn_silu_conv3x3_small(int dtype, const void* x, int B, int H, int W, int C, int ldx, int nchunks, const double* partial, const float* gamma,
                             const float* beta, float eps, int silu, const void* w, const float* bias, int No, int out_dtype, void* out, int ldo,
                             float* workspace, long long workspace_bytes, void* stream);

/* The UNet's stem -- 3x3 convolution (stride 1, pad 1) from the 9 input channels (stored in 16) to C = model_channels (openaimodel.py:666-671:
 * TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))) -- as one pixels-on-lanes kernel: x bf16 [B][H*W][ldx]
 * (channels 0..15 of a pixel; pad channels zero), W bf16 [C][144] with k = tap * 16 + c, bias fp32 [C], out bf16 [B*H*W][ldo].
 * dup_off != 0: every output row is also stored at out + dup_off elements (classifier-free guidance feeds both batch halves the same latent: one
 * half is computed).  GroupNorm(32) partial sums of the values as stored for up to THREE consumers, as rf_conv_gemm_desc.gn_* (one chunk slot per
 * 128-pixel block: slot = gn_slot + block within the sample; sample b of the kernel writes sample b of gn_part -- point gn_part at the consumer's first
 * sample of this producer, e.g. the duplicate half).  H*W a multiple of 128; C in {320, 128, 64}.  Replaces rf_conv_gemm for this layer and the
 * rf_groupnorm_stats pass over its output. */
typedef struct rf_stem_desc {
    const void* x; int32_t ldx;
    int32_t B, H, W, C;
    const void* w; const float* bias;
    void* out; int32_t ldo;
    long long dup_off;
    double* gn_part0;
    int32_t gn_cpg0, gn_coff0, gn_slot0, gn_nchunks0;
    double* gn_part1;
    int32_t gn_cpg1, gn_coff1, gn_slot1, gn_nchunks1;
    double* gn_part2;
    int32_t gn_cpg2, gn_coff2, gn_slot2, gn_nchunks2;
    int32_t dtype;        /* element type of x / w / out: RF_BF16 (also 0) or RF_F16 */
} rf_stem_desc;
int rf_conv3x3_stem(const rf_stem_desc* d, void* stream);


/* LayerNorm over the last dim of [M, C] (eps, affine).  Replaces nn.LayerNorm (attention.py:231-233,
 * xf.py:22-28, HF CLIP layer norms). */
int rf_layernorm(int dtype, const void* x, int M, int C, int ldx, const float* gamma, const float* beta, float eps,
                 int out_dtype, void* out, int ldo, void* stream);

/*
 * Fused multi-head attention  out = softmax(scale * q k^T) v  without materialising the scores.
 * q/k/v/out are [B, N, heads*d] views with pixel pitches ldq/ldk/ldv/ldo (so q, k, v may live in
 * one fused [B, N, 3*heads*d] buffer).  Nq query tokens, Nk key tokens per batch element.
 * Replaces attention.py:206-220 (einsum + softmax + einsum) and HF CLIPAttention.
 */
int rf_attention(int dtype, const void* q, const void* k, const void* v, void* out,
                 int B, int heads, int d, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo,
                 int64_t sq, int64_t sk, int64_t sv, int64_t so, float scale, void* stream);

/* fp8 activation producers of the fp8 x fp8 GEMM path (rf_conv_gemm_desc.ascale): bf16 in -> e4m3fn bytes q [rows][ldq] + one E8M0 scale byte per
 * (row, 32-channel block), scale [rows][lds]; the scale is the smallest power of two that brings the block's |max| under 448, values are
 * rounded to nearest even.  Bytes of q beyond C and scale bytes beyond C / 32 are NOT written (the caller keeps them 0 / 127).
 *   rf_quantize_fp8_act     : plain quantisation of x [M, C]
 *   rf_groupnorm_apply_fp8  : rf_groupnorm_apply (nn.GroupNorm + SiLU, util.py:214-216) with this output form
 *   rf_layernorm_fp8        : rf_layernorm (nn.LayerNorm, attention.py:231-233) with this output form */
int rf_quantize_fp8_act(const void* x, int64_t M, int C, int ldx, void* q, int ldq, void* scale, int lds, void* stream);
int rf_groupnorm_apply_fp8(const void* x, int B, int HW, int C, int ldx, int nchunks, const double* partial, const float* gamma,
                           const float* beta, float eps, int silu, void* q, int ldq, void* scale, int lds, void* stream);
int rf_layernorm_fp8(const void* x, int M, int C, int ldx, const float* gamma, const float* beta, float eps, void* q, int ldq, void* scale,
                     int lds, void* stream);

/* Row softmax over the last dim of [rows, cols] fp32, in place allowed (VAE AttnBlock, model.py:188). */
int rf_softmax_rows(float* x, int rows, int cols, int ld, void* stream);

/*
 * DDIM step glue (ddim.py:330-374):
 *   rf_ddim_pack_input : x_in[2B or B, h, w, Cpad] = [img | z_inpaint | mask | 0-pad], duplicated for CFG.
 *   rf_ddim_update     : e = e_u + s (e_c - e_u); pred_x0 = (x - sqrt(1-a_t) e) / sqrt(a_t);
 *                        x_prev = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev - sigma^2) e + sigma * noise.
 *   All latent tensors are fp32 NCHW [B, 4, h, w] (the sampler's external layout); eps is the UNet
 *   output in channels-last fp32 [2B or B, h*w, ld_eps].  `coefs` is a DEVICE array of 5 fp32 step
 *   coefficients {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2), sigma}, so one captured
 *   hipGraph of a step can be replayed for every timestep.
 */
int rf_ddim_pack_input(const float* img, const float* z_inpaint, const float* mask, int B, int hw, int dup,
                       int out_dtype, void* x_in, int Cpad, void* stream);
int rf_ddim_update(const float* eps, int ld_eps, int cfg, float scale, float* img, float* pred_x0, const float* noise,
                   int B, int hw, const float* coefs, void* stream);

/* Layout / dtype helpers. */
int rf_nchw_to_nhwc(const float* x, int B, int C, int HW, int out_dtype, void* out, int Cpad, void* stream);
int rf_nhwc_to_nchw(int dtype, const void* x, int B, int C, int HW, int ldx, float* out, void* stream);
int rf_cast(int in_dtype, const void* x, int out_dtype, void* out, int64_t n, void* stream);
/* fp32 [M, C] (row pitch ldx) -> split-bf16 pairs [M][C hi | C lo] (row pitch ldo >= 2C, bf16): hi = bf16(x), lo = bf16(x - hi), the
 * RF_BF16X3 operand form of rf_conv_gemm.  rf_groupnorm_apply writes the same form directly with out_dtype = RF_BF16X3; this pass is for
 * tensors that reach a convolution without a normalisation in between (VAE Upsample conv / nin_shortcut, model.py:53-57,117-121). */
int rf_split_bf16(const float* x, int64_t M, int C, int ldx, void* out, int ldo, void* stream);
/* sinusoidal timestep embedding [n, dim] = [cos(t f) | sin(t f)] (util.py:151-166); freqs [dim/2] fp32 */
int rf_timestep_embedding(const float* t, int n, int dim, const float* freqs, float* out, void* stream);
/* KL-VAE posterior sample (distributions.py:24-37, ddpm.py:857): moments NCHW [B, 2C, HW] = (mean | logvar),
 * eps NCHW [B, C, HW] or NULL (mode):  out = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps) */
int rf_gaussian_sample(const float* moments, const float* eps, float scale, float* out, int B, int C, int HW, void* stream);
/*
 * Conditioning-encoder side kernels (ArcFace IR-SE50: src/Face_models/encoders/helpers.py:56-119, model_irse.py:20-69,
 * ldm/models/diffusion/ddpm.py:112-124; CLIP ViT: HF CLIPVisionEmbeddings; ddpm.py:907-912).
 *   rf_channel_affine  : y[m,c] = prelu(x[m,c]*a[c] + b[c])   (eval BatchNorm as affine, optional PReLU slopes)
 *   rf_spatial_mean    : [B,HW,C] -> fp32 [B,C]              (SE squeeze, AdaptiveAvgPool2d(1))
 *   rf_se_scale_add    : out = r * s[b,c] + shortcut[b, oy*stride, ox*stride, c]   (SE excite + MaxPool2d(1,stride) shortcut)
 *   rf_adaptive_avgpool: AdaptiveAvgPool2d over a crop window of NCHW fp32 with input affine; NCHW fp32 or NHWC out
 *   rf_bilinear_resize : F.interpolate(bilinear, align_corners=False, antialias=False) of NCHW fp32 with input affine
 *   rf_clip_tokens     : [cls + pos[0] ; patch + pos[1:]] token assembly
 *   rf_l2norm_rows     : x / ||x||_2 per row (fp32)
 */
int rf_channel_affine(int dtype, const void* x, int ldx, const float* a, const float* b, const float* slope, int out_dtype,
                      void* y, int ldy, int64_t M, int C, void* stream);
int rf_spatial_mean(int dtype, const void* x, int B, int HW, int C, int ldx, float* out, void* stream);
int rf_se_scale_add(int dtype, const void* r, const float* s, const void* sc, int ldsc, int Hs, int Ws, int stride, void* out,
                    int B, int Ho, int Wo, int C, void* stream);
int rf_adaptive_avgpool(const float* x, int B, int C, int Hf, int Wf, int y0, int x0, int hc, int wc, const float* a, const float* b,
                        int Ho, int Wo, int out_nhwc, int out_dtype, int Cpad, void* out, void* stream);
int rf_bilinear_resize(const float* x, int B, int C, int Hi, int Wi, const float* a, const float* b, int Ho, int Wo, float* out, void* stream);
int rf_clip_tokens(int dtype, const void* patch, const float* cls, const float* pos, void* out, int B, int NP, int C, void* stream);
int rf_l2norm_rows(const float* x, float* y, int rows, int cols, void* stream);
/* conditioning combine (ddpm.py:1038-1039): out = (a*wa + b*wb + c*wc) / den (den = 0: no division); b, c may be NULL */
int rf_combine3(const float* a, const float* b, const float* c, float wa, float wb, float wc, float den, float* out, int64_t n, void* stream);
/* y = clamp((x + 1) / 2, 0, 1)  (scripts/inference_test_bench.py:494) */
int rf_to_image(const float* x, float* y, int64_t n, void* stream);

/*
 * Input preparation on the device -- what the test-bench dataset does per image on the host (ldm/data/test_bench_dataset.py:283-355:
 * torchvision ToTensor + Normalize, np.isin label masks, tensor products).  Same float operation order => bit-identical tensors (the resized source mask: 1 ulp).
 *   rf_u8_to_norm : uint8 [B, HW, 3] (HWC) -> fp32 [B, 3, HW]: (x / 255 - mean[c]) / std[c]
 *   rf_label_mask : uint8 label map [n] -> fp32 {0, 1}: lut256[label] != 0, optionally inverted (target keep-mask = 1 - isin)
 *   rf_mul_mask   : out[b, c, p] = x[b, c, p] * mask[b, p]
 */
int rf_u8_to_norm(const void* x_u8, int B, int HW, const float* mean3, const float* std3, float* out, void* stream);
int rf_label_mask(const void* labels_u8, int64_t n, const void* lut256_u8, int invert, float* out, void* stream);
int rf_mul_mask(const float* x, const float* mask, int B, int C, int HW, float* out, void* stream);
/*
 * rf_resize_u8_linear : cv2.resize(img, (Wo, Ho), interpolation=cv2.INTER_LINEAR) of uint8 HWC images [B, H, W, C] (image b at
 * x + b * image_stride bytes) -> uint8 [B, Ho, Wo, C]: what albumentations' A.Resize(224, 224) runs on the source face
 * (ldm/data/test_bench_dataset.py:141-148, 324; scripts/inference_swap_selected.py:525-553).  OpenCV's integer algorithm and operation order (no cv2-generated golden exists here: last bit unpinned)
 * (half-pixel centres, two taps per axis, 11-bit weights, NO antialiasing; exact 2:1 -> the fast-area average); cv2 itself is a
 * third-party dependency absent from /root/reference and from this image: see reface_amd/data.py::resize_u8_linear for the restatement.
 */
/*
 * rf_compose_outputs_u8 : the CLI's output files as uint8 HWC arrays, one packed record per image:
 *     [result | mask (grey replicated) | GT | inpaint | ref]  5 x [H][W][3],  then (with_grid) the 4-panel make_grid image
 *     [H + 4][4 W + 10][3] of (GT, inpaint, ref, result) with 2-pixel padding (pad bytes untouched: zero them once).
 * Every conversion is the reference's (255. * x).astype(np.uint8) of fp32 values (scripts/inference_test_bench.py:500-552:
 * un_norm, un_norm_clip, make_grid, rearrange + astype): truncation toward zero, low byte.  Inputs NCHW fp32 on the device:
 * result01 = clamp((x + 1) / 2, 0, 1), target / inpaint in [-1, 1], mask [B, 1, H, W] in {0, 1}, ref = the CLIP-normalised source
 * already resized to H x W.  Replaces the per-image host float passes; the host then only PNG-encodes slices of one D2H copy.
 */
int rf_compose_outputs_u8(const float* result01, const float* target, const float* inpaint, const float* mask, const float* ref, int B, int H, int W,
                          int with_grid, void* out_u8, int64_t record_bytes, void* stream);
int rf_resize_u8_linear(const void* x_u8, int B, int H, int W, int C, int64_t image_stride, int Ho, int Wo, void* out_u8, void* stream);
/* elementwise y = silu(x) on fp32 (emb path, openaimodel.py:219) */
int rf_silu_f32(const float* x, float* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REFACE_HIP_H */
