/*
 * diffsal.h -- C ABI of libdiffsal_hip.so: the MI355X (gfx950) operator library for
 * DiffSal's per-step denoiser (SalUNet) hot path.
 *
 * Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless said otherwise;
 *   - activations are channels-last: images [N,H,W,C], tokens [M,C]; fp32 by default.  Forward operators that take a
 *     `dtype` argument (DIFFSAL_F32 / DIFFSAL_BF16 / DIFFSAL_F16, declared below) read and write their `void*`
 *     activation tensors in that storage type; `float*` arguments (norm parameters, biases, depthwise weights,
 *     the timestep path, the sampler state) are always fp32, and all arithmetic / statistics are fp32;
 *   - every entry point enqueues work on `stream` (a hipStream_t) and returns immediately:
 *       0 on success, a negative DIFFSAL_E_* code on failure (diffsal_last_error() has the text);
 *   - no allocation, no synchronisation, no host-visible side effect inside an entry point,
 *     so every call is legal under HIP-graph capture;
 *   - re-entrant; the only global state is a thread-local error string (arithmetic mode and storage type are
 *     per-call arguments: diffsal_conv_desc.precision / .dtype and the `dtype` argument of the other operators).
 *
 * Each entry point cites the reference interface it replaces (R/ = junwenxiong/diff_sal).
 */
#ifndef DIFFSAL_H
#define DIFFSAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* diffsal_stream_t; /* hipStream_t */

enum {
  DIFFSAL_OK = 0,
  DIFFSAL_E_SHAPE = -1,   /* unsupported / inconsistent dimensions */
  DIFFSAL_E_ALIGN = -2,   /* pointer not 16-byte aligned */
  DIFFSAL_E_LAUNCH = -3,  /* hipGetLastError() after launch */
  DIFFSAL_E_ARG = -4      /* null pointer / bad enum */
};

enum { DIFFSAL_ACT_NONE = 0, DIFFSAL_ACT_RELU = 1, DIFFSAL_ACT_GELU_ERF = 2, DIFFSAL_ACT_SIGMOID = 3,
       /* training only, fp32 diffsal_conv_igemm only: out = (product + bias ...) * gelu'(residual[m, co]) -- `residual` carries the
        * pre-activation of the erf-GELU in front of the layer whose data gradient this product is, and is NOT added */
       DIFFSAL_ACT_GELU_GRAD = 5 };

int diffsal_version(void);
const char* diffsal_last_error(void);
/* Name and tile plan of the kernel the last diffsal_conv_igemm / diffsal_linear_pair / diffsal_conv_wino call of THIS thread
 * launched (e.g. "gemm_dma_kernel<96x96,3 stages>", "igemm_linear_kernel<64x64>", "igemm_kernel<128x96> split-K 2"): lets a
 * profiler attribute HIP-event times to kernels without a tracing tool (bench.py's roofline.dominant_kernel).  Thread-local,
 * never NULL; not part of the reference's surface. */
const char* diffsal_last_gemm_kernel(void);

/* Test / tuning switches (kernel-variant selection for bit-equality tests and tile sweeps; none changes results beyond
 * summation order).  The DIFFSAL_* environment variables of the same names are read ONCE when the library is loaded; afterwards
 * only this call changes a switch (value < 0 = unset).  Names: DIFFSAL_NO_PERSIST, DIFFSAL_NO_XCD_ORDER, DIFFSAL_NO_HALO,
 * DIFFSAL_FORCE_HALO, DIFFSAL_IGEMM_CFG, DIFFSAL_IGEMM16_CFG, DIFFSAL_PLAN_DEBUG, DIFFSAL_WGRAD_CFG, DIFFSAL_WGRAD_SPLITS,
 * DIFFSAL_WGRAD_VERBOSE, DIFFSAL_NO_FUSED_BLOCK, ... (the full list: kTuneNames in csrc/misc.hip); round 6:
 * DIFFSAL_NO_STREAM16 = 1 routes 16-bit storage back to the 8-byte forms of the HBM-bound kernels and to the kernels the planner took before
 * conv16_dma / gemm16_dma2; DIFFSAL_FORCE_HALO = 2 and DIFFSAL_GEMM_DMA16 = 3 / 4 take those two (gemm16_dma2 with its 256 x 96 / 192 x 192
 * tile) on every shape they can run; DIFFSAL_CONV16_TILE = 0 / 1 keeps conv16_dma off / on its 8 x 24 x 192-channel tile, DIFFSAL_CONV16_HALF = 0 / 1 off / on its 128-pixel tiles;
 * DIFFSAL_BLOCK16_WAVES = 4 / 8 fixes the wavefronts per workgroup of the fused C = 96 block; DIFFSAL_TAPSUM_ROWS_FORM = 1 .. 4 takes the
 * row-streamed head gather (and its lane mapping) instead of the LDS-staged one; DIFFSAL_NO_ATTN16_MFMA = 1 keeps the 16-bit attention core
 * of head dims 192 / 384 off the matrix cores.
 * Not part of the reference's surface (it has no such knobs). */
int diffsal_set_tuning(const char* name, int value);
int diffsal_get_tuning(const char* name);   /* current value, -1 if unset or unknown */

/* ---- K1: timestep embedding + MLP ------------------------------------------------------
 * R/models/saliency_decoder/sal_unet.py:15-33 (get_timestep_embedding) and :304-307.
 * t: [B] int64 (t_is_f32 == 0) or float (t_is_f32 == 1); freq: [ch/2] table exp(-j*ln(1e4)/(ch/2-1)).
 * temb_out[B,4ch] = W1 * swish(W0 * [sin(t f), cos(t f)] + b0) + b1 */
int diffsal_temb_mlp(const void* t, int t_is_f32, int B, int ch, const float* freq,
                     const float* w0, const float* b0, const float* w1, const float* b1,
                     float* hidden_ws /* scratch [B,4ch] */, float* temb_out, diffsal_stream_t stream);

/* out[B,N] = W[N,K] * f(in[B,K]) + bias, f = swish if swish_in.  ResnetBlock.temb_proj of all
 * blocks in one call (weights concatenated along N).  R/.../sal_unet.py:107,129. */
int diffsal_dense_small(const float* in, int B, int K, int swish_in, const float* w, const float* bias,
                        int N, float* out, diffsal_stream_t stream);

/* ---- K12 / K14 restructured: a 3x3 convolution (dilation dil) of a bilinearly up-sampled map, or of a SUM of up-sampled
 * maps, without forming the up-sampled map (csrc/tapsum.hip):
 *   conv3x3( sum_i up_i(z_i) )[p] = sum_i sum_tap up_i( W_tap z_i )[p + dil * delta_tap]       (1x1 mixing commutes with bilerp)
 * The nine channel mixings of every source run as ONE diffsal_conv_igemm at the SOURCE resolution,
 * Y_i [N, h_i, w_i, 9*C] = z_i x Wcat^T with Wcat[tap*C + co][ci] = W[co][ci][ky][kx]; this entry gathers
 *   out[n,Y,X,c] = act(scale[c] * (bias[c] + sum_i sum_tap [p' inside] bilerp_i(Y_i[..., tap*C + c]; p')) + shift[c])
 * with p' = (Y,X) + dil*(ky-1,kx-1) (taps outside the image are the convolution's zero padding).  Sources: 1..4, each a
 * power-of-two factor (>= 1) smaller than H x W; act NONE or RELU.  UpEmbed's first convolution
 * (bilinear x2 + 3x3 dilation 2, common_block.py:196-216): 4x fewer FLOPs; mt_proj on the 4-scale sum (sal_unet.py:480-489,
 * :407): 3x fewer.  Exact up to summation order. */
int diffsal_tapsum(const void* const* srcs, const int* hs, const int* ws, int n_src, void* out, int N, int H, int W, int C,
                   int dil, const float* bias, const float* scale, const float* shift, int act, int dtype,
                   diffsal_stream_t stream);
/* The same gather with MLPHead folded into its epilogue (common_block.py:111-122: 1x1 convolution C -> 1 + sigmoid; C <= 128):
 *   head_out[n, Y, X] = sigmoid( head_b[0] + sum_c head_w[c] * out[n, Y, X, c] ),   fp32 [N, H, W]
 * the C-channel map `out` is never stored (mt_proj -> BN -> ReLU -> logits of sal_unet.py:489 + :320 in one launch). */
int diffsal_tapsum_head(const void* const* srcs, const int* hs, const int* ws, int n_src, int N, int H, int W, int C, int dil,
                        const float* bias, const float* scale, const float* shift, int act, const float* head_w,
                        const float* head_b, float* head_out, int dtype, diffsal_stream_t stream);
/* Training: the adjoint of diffsal_tapsum (act NONE, no affine) with respect to ONE source's tap products,
 *   dy [N,h,w,9*C] (fp32) from du [N,H,W,C]; ws: diffsal_tapsum_bwd_ws_bytes(N, W, C, h) bytes of scratch (the three row passes).
 * Gather form, deterministic; replaces the full-resolution dgrad + wgrad of the convolution behind nn.Upsample in training
 * (common_block.py:196-216, sal_unet.py:480-489). */
long diffsal_tapsum_bwd_ws_bytes(int N, int W, int C, int h);
int diffsal_tapsum_bwd(const float* du, float* dy, float* ws, int N, int H, int W, int C, int h, int w, int dil,
                       diffsal_stream_t stream);

/* ---- K2 fused: conv_in followed directly by Downsample4x4's 3x3 stride-4 convolution (sal_unet.py:240,292 + :67-84) as ONE
 * 5x5 stride-4 convolution of the single-channel input; w25 [25][C] (tap-major) and bias [C] are the composed weights
 * W_eff[co][dy][dx] = sum_ci sum_{ky2+ky1=dy, kx2+kx1=dx} W2[co,ci,ky2,kx2] W1[ci,ky1,kx1], b_eff = b2 + sum W2 b1 (host side).
 * Exact for H, W multiples of 4.  x NCHW [B,1,H,W] fp32 -> out NHWC [B,H/4,W/4,C] in `dtype`. */
int diffsal_conv_in_s4(const float* x, const float* w25, const float* bias, void* out, int B, int H, int W, int C, int dtype,
                       diffsal_stream_t stream);

/* ---- K2: conv_in (1 -> C, 3x3, pad 1), NCHW[B,1,H,W] -> NHWC[B,H,W,C] -----------------
 * R/.../sal_unet.py:240,292.  Only pixels with (y % skip_mod != skip_mod-1 && x % skip_mod != skip_mod-1)
 * are written when skip_mod > 0 (the stride-4 consumer never reads the others, sal_unet.py:67-84). */
int diffsal_conv_in(const float* x, const float* w /*[C,9]*/, const float* bias, void* out,
                    int B, int H, int W, int C, int skip_mod, int act /*DIFFSAL_ACT_NONE | RELU*/, int dtype,
                    diffsal_stream_t stream);

/* ---- K3: GroupNorm(groups, eps) + swish on NHWC ----------------------------------------
 * R/.../sal_unet.py:36-44.  ws: >= diffsal_groupnorm_ws_bytes(B, groups) bytes of scratch. */
size_t diffsal_groupnorm_ws_bytes(int B, int groups);
int diffsal_groupnorm_swish(const void* x, const float* gamma, const float* beta, void* out,
                            int B, int HW, int C, int groups, float eps, void* ws, size_t ws_bytes, int dtype,
                            diffsal_stream_t stream);

/* ---- K4/K5/K10/K12/K13/K14: implicit-GEMM convolution / linear on fp32 MFMA -------------
 * out[m, co] = act( ((sum_k A[m,k] * w[co,k]) + bias[co]) * scale[co] + shift[co] + rowvec[img(m), co] )
 *              + residual[m, co]
 * A is the im2col view of in[N,H,W,Cin] (never materialised): m = (n, oy, ox), k = (ci / 32, ky, kx, ci % 32),
 * iy = oy*stride_h - pad_t + ky*dil_h, ix = ox*stride_w - pad_l + kx*dil_w, zero outside.
 * w: packed [Cout][Cin/32][KH*KW][32] (= [Cout][K] in that k order).  bias/scale/shift: [Cout] or NULL.  rowvec: [N, rowvec_ld] or NULL.
 * residual: [M, Cout] or NULL.  Cin % 32 == 0.  A plain linear layer is KH=KW=1, H=1, W=rows.
 * Replaces torch.nn.Conv2d / Conv3d(k,1,1) / Linear call sites:
 *   R/.../sal_unet.py:104-142 (ResnetBlock), :47-84 (Downsample*), common_block.py:33-36,150-223,
 *   attention.py:78-83, common_block.py:125-147, transformer.py:122,131. */
typedef struct diffsal_conv_desc {
  int N, H, W, Cin;
  int Ho, Wo, Cout;
  int KH, KW, stride_h, stride_w, pad_t, pad_l, dil_h, dil_w;
  int act;        /* DIFFSAL_ACT_* */
  int rowvec_ld;  /* leading dimension of rowvec (>= Cout) */
  int w_format;   /* 0: fp32 [Cout][K] (k order above).  1: the same rows pre-split for the bf16x3 mode by
                   * diffsal_split_weight: per 32-k slice 16 dwords of bf16 hi halves then 16 dwords of lo halves (same
                   * size); only valid with precision == DIFFSAL_PREC_BF16X3 */
  int precision;  /* DIFFSAL_PREC_*: arithmetic of the matrix-core loop for fp32 storage (below) */
  int dtype;      /* DIFFSAL_F32 / DIFFSAL_BF16 / DIFFSAL_F16: storage type of in, w, residual and out (bias, scale,
                   * shift, rowvec are always fp32; accumulation is always fp32).  16-bit storage uses the native
                   * v_mfma_f32_32x32x16_{bf16,f16}; `precision` must then be DIFFSAL_PREC_FP32 (= "native") */
} diffsal_conv_desc;

/* Arithmetic of diffsal_conv_igemm's matrix-core loop for fp32 tensors, chosen PER CALL (diffsal_conv_desc.precision):
 *   DIFFSAL_PREC_FP32 (default): exact fp32, v_mfma_f32_32x32x2_f32;
 *   DIFFSAL_PREC_BF16X3: each fp32 operand is split into bf16 hi + lo on its way into LDS and the product is formed as
 *     hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (error ~2^-16 relative per product,
 *     ~4e-6 of the output maximum on the network's convolutions; well inside the 1e-3 parity bar, but not bit-equal
 *     to fp32).  Opt-in; benchmark numbers for it are reported separately from the headline. */
enum { DIFFSAL_PREC_FP32 = 0, DIFFSAL_PREC_BF16X3 = 1 };
/* Storage type of activations / weights (BASELINE configs[1] "bf16", configs[4] "fp16"): statistics of every
 * normalisation, softmax and all accumulations stay fp32; parameters of norms, biases and the timestep path stay fp32. */
enum { DIFFSAL_F32 = 0, DIFFSAL_BF16 = 1, DIFFSAL_F16 = 2 };

/* ---- scratch sizes, one entry point --------------------------------------------------------------------------
 * Bytes of caller-provided scratch the operator `op` needs for a shape: `d` for the operators described by a diffsal_conv_desc,
 * `dims` (n_dims integers, in the order of the typed function's arguments) for the others; 0 = the operator needs none for this
 * shape, or op / argument count unknown.  The typed functions below (diffsal_*_ws_bytes, ..._floats) are the same numbers with
 * named arguments; the library never allocates, a workspace is whatever the caller's allocator hands over. */
enum {
  DIFFSAL_WS_GROUPNORM = 0,      /* dims = {B, groups}                 diffsal_groupnorm_ws_bytes (also diffsal_gn_affine) */
  DIFFSAL_WS_CONV_IGEMM = 1,     /* d                                  diffsal_conv_igemm_ws_bytes */
  DIFFSAL_WS_CONV_WINO = 2,      /* d                                  diffsal_conv_wino_ws_bytes */
  DIFFSAL_WS_CONV_WINO4 = 3,     /* d                                  diffsal_conv_wino4_ws_bytes */
  DIFFSAL_WS_CONV_WINO4_STATS = 4, /* d, dims = {groups}               diffsal_conv_wino4_stats_bytes */
  DIFFSAL_WS_CONV_WGRAD = 5,     /* d                                  diffsal_conv_wgrad_ws_bytes */
  DIFFSAL_WS_WGRAD_SEGMENTED = 6, /* dims = {segments, seg_rows, K, Cout}  diffsal_wgrad_segmented_ws_bytes */
  DIFFSAL_WS_TAPSUM_BWD = 7,     /* dims = {N, W, C, h}                diffsal_tapsum_bwd_ws_bytes */
  DIFFSAL_WS_SALIENCY_METRICS = 8, /* dims = {B}                       diffsal_saliency_metrics_ws_bytes */
  DIFFSAL_WS_ATTENTION_TAIL = 9, /* dims = {B, H, Lq, Lk, DV}          4 x diffsal_attention_general_tail_floats */
  DIFFSAL_WS_ATTENTION_BWD_QTAIL = 10, /* dims = {B, H, Lq, Lk, D, E}  4 x diffsal_attention_general_bwd_qtail_floats */
  DIFFSAL_WS_ATTENTION_BWD_DS = 11     /* dims = {B, H, Lq, Lk}        4 x diffsal_attention_general_bwd_ds_floats */
};
size_t diffsal_workspace_bytes(int op, const diffsal_conv_desc* d /*host, may be NULL*/, const long* dims /*host*/, int n_dims);

/* Bytes of scratch the call below needs for this shape (0 unless the planner picks split-K, which it does
 * when the M x Cout grid alone cannot fill the 256 CUs). */
size_t diffsal_conv_igemm_ws_bytes(const diffsal_conv_desc* d /*host*/);
int diffsal_conv_igemm(const diffsal_conv_desc* d /*host*/, const void* in, const void* w,
                       const float* bias, const float* scale, const float* shift, const float* rowvec,
                       const void* residual, void* out, void* ws, size_t ws_bytes, diffsal_stream_t stream);

/* 16-bit storage in, fp32 out (plain products: KH = KW = 1, d->dtype bf16 / f16): out [M, Cout] fp32 = act(in x w^T + bias), the sums
 * leave without a rounding to the storage type -- for consumers that add many products (the nine interpolated tap products of
 * mt_proj, R/models/saliency_decoder/sal_unet.py:480-489).  K % 192 == 0, Cout % 4 == 0. */
int diffsal_linear_f32out(const diffsal_conv_desc* d /*host*/, const void* in, const void* w, const float* bias, float* out,
                          diffsal_stream_t stream);

/* Winograd F(2x2, 3x3) form of the same operator for fp32 3x3 stride-1 convolutions with padding = dilation in {1, 2},
 * Cin % 32 == 0, Cout % 4 == 0 (ResnetBlock.conv1 / conv2, R/models/saliency_decoder/sal_unet.py:104-142; UpEmbed's second
 * convolution, common_block.py:196-216): 2.25x fewer multiplications; the result differs from the direct convolution by
 * transform rounding (~1e-6 relative).  `U` is the transformed weight G g G^T in blocked layout
 * [Cin/8][ceil(Cout/64)][16][64][8] (rows past Cout zero): the host mirror builds it with ops.pack_wino_weight.
 * diffsal_conv_wino_supported: 1 when the descriptor qualifies AND the planner expects a gain over diffsal_conv_igemm
 * (Cin, Cout >= 192; DIFFSAL_NO_WINOGRAD=1: never, DIFFSAL_FORCE_WINOGRAD=1: whenever the shape qualifies).
 * Epilogue and argument meaning as diffsal_conv_igemm; ws: diffsal_conv_wino_ws_bytes(d) bytes (transformed input + split slabs). */
int diffsal_conv_wino_supported(const diffsal_conv_desc* d /*host*/);
size_t diffsal_conv_wino_ws_bytes(const diffsal_conv_desc* d /*host*/);
int diffsal_conv_wino(const diffsal_conv_desc* d /*host*/, const float* x, const float* U, const float* bias,
                      const float* scale, const float* shift, const float* rowvec, const float* residual, float* out, void* ws,
                      size_t ws_bytes, diffsal_stream_t stream);
/* Winograd F(4x4, 3x3) form of the same operator (same layers and shape rules as diffsal_conv_wino, Cin % 96 == 0): 4x fewer
 * multiplications than the direct convolution; transform rounding ~1e-5 of the output maximum.  `U` is G g G^T as
 * [36][Cout][Cin] fp32 (ops.pack_wino4_weight).  Three launches: input transform, the 36 position products as one batched
 * plain product, output transform + epilogue.  ws: diffsal_conv_wino4_ws_bytes(d) = 36 * tiles * (Cin + Cout) floats.
 * diffsal_conv_wino4_supported: the shape qualifies and the planner expects a gain (DIFFSAL_NO_WINOGRAD4=1 /
 * DIFFSAL_NO_WINOGRAD=1: never, DIFFSAL_FORCE_WINOGRAD=1: whenever the shape qualifies). */
int diffsal_conv_wino4_supported(const diffsal_conv_desc* d /*host*/);
/* U = G g G^T on the device, fp32 (training: the weights change every step).  w [Cout][Cin][3][3]; dgrad = 0 -> U [36][Cout][Cin] (the forward
 * convolution); dgrad = 1 -> U [36][Cin][Cout] of the flipped kernel: the data gradient dX = conv(dY, .) of a stride-1, padding = dilation
 * convolution runs on the same F(4x4) path with it. */
int diffsal_wino4_weight(const float* w, float* U, int Cout, int Cin, int dgrad, diffsal_stream_t stream);
size_t diffsal_conv_wino4_ws_bytes(const diffsal_conv_desc* d /*host*/);
int diffsal_conv_wino4(const diffsal_conv_desc* d /*host*/, const float* x, const float* U, const float* bias,
                       const float* scale, const float* shift, const float* rowvec, const float* residual, float* out, void* ws,
                       size_t ws_bytes, diffsal_stream_t stream);
/* The same with the three launches selectable (stages: bit 0 input transform, bit 1 position products, bit 2 output transform +
 * epilogue; 7 = diffsal_conv_wino4): same arguments and workspace for every call of one convolution.  For profilers that bracket
 * operator launches (bench.py attributes the transforms to the HBM-bound classes and the products to the GEMM kernel). */
int diffsal_conv_wino4_stages(const diffsal_conv_desc* d /*host*/, const float* x, const float* U, const float* bias,
                              const float* scale, const float* shift, const float* rowvec, const float* residual, float* out,
                              void* ws, size_t ws_bytes, int stages, diffsal_stream_t stream);
/* Extensions of the F(4x4) path for a ResnetBlock (R/models/saliency_decoder/sal_unet.py:123-142: h = conv1(swish(norm1(x))) +
 * temb; out = nin_shortcut(x) + conv2(swish(norm2(h)))), all optional (zero / NULL = off):
 *   in_ab, in_swish   the input is read through y = a[n][c] x + b[n][c] (ab = [N][2][Cin]: scale row then shift row per image:
 *                     GroupNorm in affine form, diffsal_gn_affine / diffsal_gn_affine_wino4) and, if in_swish, y sigmoid(y), as the
 *                     input transform loads it: no normalised tensor in memory.  The zero padding applies to the normalised map.
 *   side_*            side_out [side_rows, Cout] = side_a [side_rows, Cin] x side_w [Cout, Cin]^T (no epilogue) computed by the SAME
 *                     launch as the 36 position products, as side_rows / tiles further problems of their shape (the block's 1x1
 *                     shortcut; diffsal_conv_wino4_side_supported: side_rows must be a multiple of the tile count).
 *   out_stats, out_groups   the output transform leaves per-(image, group) sums / sums of squares of the RESULT in out_stats
 *                     (diffsal_conv_wino4_stats_bytes bytes; 0 = not available for the shape) for diffsal_gn_affine_wino4:
 *                     the GroupNorm that follows needs no statistics pass. */
typedef struct diffsal_wino4_ext {
  const float* in_ab;
  const float* side_a;
  const float* side_w;
  float* side_out;
  double* out_stats;
  const float* up2_c;       /* see below */
  const float* up2_scale;
  const float* up2_shift;
  long long side_rows;
  int in_swish;
  int out_groups;
  int up2_act;
  int reserved;
} diffsal_wino4_ext;
/*   up2_c, up2_scale, up2_shift, up2_act   UpEmbed's second convolution (dilation 2) fed from the SOURCE-resolution result of the
 *                     first one (R/.../common_block.py:196-216): up2_c = conv3x3(z) on the extended grid [N][H/2 + 2][W/2 + 2][Cin]
 *                     (what diffsal_up2_conv_commute takes as c_ext); the input transform forms act(BN(interpolation of up2_c)) itself
 *                     -- the arithmetic of diffsal_up2_conv_commute's interior -- and reads from `x` only the 3-pixel border ring, which
 *                     diffsal_up2_conv_commute_ring must have written there.  dil = 2, H and W even, fp32; excludes in_ab. */
size_t diffsal_conv_wino4_stats_bytes(const diffsal_conv_desc* d /*host*/, int groups);
int diffsal_conv_wino4_side_supported(const diffsal_conv_desc* d /*host*/, long side_rows);
int diffsal_conv_wino4_ex(const diffsal_conv_desc* d /*host*/, const float* x, const float* U, const float* bias,
                          const float* scale, const float* shift, const float* rowvec, const float* residual, float* out,
                          void* ws, size_t ws_bytes, const diffsal_wino4_ext* ext /*host, may be NULL*/, int stages,
                          diffsal_stream_t stream);
/* GroupNorm(groups, eps; gamma, beta) in affine form: ab [B][2][C] with y = ab[b][0][c] x + ab[b][1][c] (statistics in fp64).
 * diffsal_gn_affine: from the tensor x [B, HW, C] (one statistics launch + a finishing launch; ws as diffsal_groupnorm_swish);
 * diffsal_gn_affine_wino4: from the sums a diffsal_conv_wino4_ex call left in ext.out_stats (same d, same groups).
 * R/models/saliency_decoder/sal_unet.py:41-44. */
int diffsal_gn_affine(const void* x, const float* gamma, const float* beta, float* ab, int B, int HW, int C, int groups,
                      float eps, void* ws, size_t ws_bytes, int dtype, diffsal_stream_t stream);
int diffsal_gn_affine_wino4(const diffsal_conv_desc* d /*host*/, const double* stats, const float* gamma, const float* beta,
                            int groups, float eps, float* ab, diffsal_stream_t stream);
/* zb [N][2 w + 2 h - 4][C] = the border pixels of z [N][h][w][C] in the order diffsal_up2_conv_commute reads their tap products
 * (top row, bottom row, left column, right column; the columns without their corner pixels).  C % 8 == 0. */
int diffsal_border_gather(const void* z, void* zb, int N, int h, int w, int C, int dtype, diffsal_stream_t stream);
/* conv3x3(dilation 2, padding 2)(bilinear_up2(z)) -> BN affine -> activation (UpEmbed's first convolution,
 * R/models/saliency_decoder/common_block.py:196-206) from c_ext = conv3x3(z) (dilation 1, zero padding) evaluated at the SOURCE
 * resolution on the grid extended by one pixel on every side ([N][h + 2][w + 2][C]: diffsal_conv_wino4 / diffsal_conv_igemm with
 * padding 2 and output (h + 2) x (w + 2)) and tap_border = the nine tap products W_tap z of the border pixels of z
 * ([N][2 w + 2 h - 4][9][C]: top row, bottom row, left column and right column without their corners; tap = 3 ky + kx).  A shift by two output pixels is a shift by
 * one source pixel: the result is the unclamped interpolation of c_ext plus corrections on a 3-pixel border ring (csrc/upconv.hip);
 * exact up to summation order.  c_ext, tap_border and out ([N][2 h][2 w][C]) are in the storage type `dtype` (fp32 arithmetic in
 * between); act: DIFFSAL_ACT_NONE or DIFFSAL_ACT_RELU. */
int diffsal_up2_conv_commute(const void* c_ext, const void* tap_border, const float* scale, const float* shift, void* out,
                             int N, int h, int w, int C, int act, int dtype, diffsal_stream_t stream);
/* the same, writing ONLY the 3-pixel border ring of `out` (for diffsal_conv_wino4_ex with ext.up2_c) */
int diffsal_up2_conv_commute_ring(const void* c_ext, const void* tap_border, const float* scale, const float* shift, void* out,
                             int N, int h, int w, int C, int act, int dtype, diffsal_stream_t stream);
/* Up to four independent convolutions / plain products (own descriptor, operands and output; bias + activation epilogue
 * only) in ONE launch: the four ReduceTemp products of a step (R/models/saliency_decoder/common_block.py:150-173,
 * sal_unet.py:480-487) have 336 .. 21504 rows and 3840 .. 480 columns of K, each alone fills a fraction of the chip.  fp32
 * problems that fit the LDS-DMA kernel (K a multiple of 96) share one grid, longest units first; anything else runs as n
 * diffsal_conv_igemm calls (same results).  `ws` / `ws_bytes`: the largest diffsal_conv_igemm_ws_bytes of the problems. */
int diffsal_conv_igemm_group(int n, const diffsal_conv_desc* const* descs, const void* const* in, const void* const* w,
                             const float* const* bias, void* const* out, void* ws, size_t ws_bytes, diffsal_stream_t stream);

/* Two plain products of ONE shape in one launch (grid z = 2): out_i [M,N] = in_i [M,K] x w_i [N,K]^T + bias_i.  The key and value
 * projections of a transformer block (attention.py:78-83: proj_k / proj_v on the 648 pooled tokens of a stage) are 8-26 us
 * launches at their latency floor; paired they are four launches per step instead of eight.  d: a 1x1 descriptor (KH = KW = 1,
 * act NONE, exact arithmetic, any storage type); ws: 2 x diffsal_conv_igemm_ws_bytes(d) bytes. */
int diffsal_linear_pair(const diffsal_conv_desc* d, const void* in0, const void* in1, const void* w0, const void* w1,
                        const float* bias0, const float* bias1, void* out0, void* out1, void* ws, size_t ws_bytes,
                        diffsal_stream_t stream);


/* ---- weight layout transforms (w: the reference's parameter layout [Cout][Cin][taps], taps = KH*KW or KT):
 *   mode 0: dst[co][(ci/32, tap, ci%32)] = w[co][ci][tap]            -- the `w` argument of diffsal_conv_igemm
 *   mode 1: dst[ci][(co/32, tap, co%32)] = w[co][ci][taps-1-tap]     -- `w` of the data-gradient conv (Cout % 32 == 0)
 *   mode 2: dst[co][ci][tap] = src[co][(ci/32, tap, ci%32)]          -- diffsal_conv_wgrad output back to parameter layout
 *   mode 3: dst[(tap, ci)][co] = w[co][ci][tap]                      -- GEMM weight of dXcols = dY W (Cout % 32 == 0)
 * Cin % 32 == 0, taps <= 25.
 * col2im_disjoint: data gradient of a convolution whose taps never overlap (stride >= kernel, e.g. the 3x3 stride-4
 * Downsample, sal_unet.py:47-84, and ReduceTemp's k=s=5 Conv3d, common_block.py:125-142): dx[n,iy,ix,:] = the one
 * cols[(n,oy,ox)][(ky,kx)][:] that maps there, else 0; cols [N*Ho*Wo, KH*KW*C] comes from one diffsal_conv_igemm GEMM. */
int diffsal_col2im_disjoint(const float* cols, float* dx, int N, int H, int W, int C, int Ho, int Wo, int KH, int KW,
                            int stride_h, int stride_w, int pad_t, int pad_l, diffsal_stream_t stream);
/* col2im_gather: the same for OVERLAPPING taps (1 < stride < kernel: the 3x3 stride-2 Downsample of the ResBlock encoder,
 * sal_unet.py:47-84): every input pixel sums, in (ky, kx) order, the column entries that map onto it.  With the mode-3
 * GEMM in front this is the strided data gradient at exactly the forward's MAC count. */
int diffsal_col2im_gather(const float* cols, float* dx, int N, int H, int W, int C, int Ho, int Wo, int KH, int KW,
                          int stride_h, int stride_w, int pad_t, int pad_l, diffsal_stream_t stream);
int diffsal_pack_weight(const float* src, float* dst, int Cout, int Cin, int taps, int mode, diffsal_stream_t stream);
/* Many repacks in one launch (training: every layer's layouts are rebuilt after each optimizer step).  jobs_dev: DEVICE array of
 * n_jobs entries sorted by tile0, tile0 = running sum of (Cin / 32) * ceil(Cout / 32); total_tiles = that sum over all jobs;
 * max_taps = largest taps of the batch.  Each job obeys diffsal_pack_weight's shape rules. */
typedef struct diffsal_pack_job {
  const void* src;
  void* dst;
  int Cout, Cin, taps, mode;
  int tile0, reserved;
} diffsal_pack_job;
int diffsal_pack_weight_many(const diffsal_pack_job* jobs_dev /*device*/, int n_jobs, int total_tiles, int max_taps,
                             diffsal_stream_t stream);
/* bf16x3 mode: split a packed fp32 weight [rows][K] (K % 32 == 0) into the w_format = 1 layout of diffsal_conv_desc
 * (hi = bf16(x), lo = bf16(x - hi)); n = rows * K.  Saves the kernel the conversion of its B operand. */
int diffsal_split_weight(const float* src, float* dst, long n, diffsal_stream_t stream);

/* ---- K16 (training): weight gradient of the layer above, in the SAME packed layout as `w`:
 * dw[co, k] = sum_m dy[m, co] * A[m, k].  Replaces the wgrad of autograd's conv / linear backward.
 * ws: >= diffsal_conv_wgrad_ws_bytes(d) bytes (partial slabs, summed in a fixed order).  Cout % 4 == 0.
 * dbias_part (optional, may be NULL): [diffsal_conv_wgrad_splits(d)][Cout] doubles receiving the column sums of dy per
 * M split -- the bias gradient rides along on the dY tiles the kernel stages anyway.  dbias_out (optional, needs dbias_part):
 * [Cout] floats, the finished bias gradient (split sums added in split order by the launch that sums the weight slabs, or
 * written directly when there is one split); without it finish with diffsal_reduce_partials(dbias_part, db, 1, splits, Cout, 0). */
size_t diffsal_conv_wgrad_ws_bytes(const diffsal_conv_desc* d /*host*/);
int diffsal_conv_wgrad_splits(const diffsal_conv_desc* d /*host*/);
int diffsal_conv_wgrad(const diffsal_conv_desc* d /*host*/, const float* in, const float* dy, float* dw_packed,
                       double* dbias_part, float* dbias_out, void* ws, size_t ws_bytes, diffsal_stream_t stream);
/* Batched form for token GEMMs: out[s][co][k] = sum over the seg_rows rows m of segment s of dy[m, co] * x[m, k]
 * (x: [segments*seg_rows, K], dy: [segments*seg_rows, Cout]).  Used by the attention backward (one segment per
 * image: dK = dS^T Q, dV = P^T dO; R/.../attention.py:97-108).  K % 32 == 0, Cout % 4 == 0. */
size_t diffsal_wgrad_segmented_ws_bytes(int segments, int seg_rows, int K, int Cout);
int diffsal_wgrad_segmented(const float* x, const float* dy, float* out, int segments, int seg_rows, int K, int Cout,
                            void* ws, size_t ws_bytes, diffsal_stream_t stream);
/* out[g, c] = sum of dy[m, c] over the rows of segment g (M / seg_rows segments): bias gradients (one segment)
 * and per-image vector gradients (one segment per image).  ws: >= (M/seg_rows) * 512 * C * 8 bytes (fp64 partials:
 * reductions across threads run in double so that heavily cancelling sums do not depend on the atomics order). */
int diffsal_colsum(const float* dy, float* out, int M, int C, int seg_rows, void* ws, size_t ws_bytes,
                   diffsal_stream_t stream);

/* dx = dy * act'(.): mode 1 ReLU (ref = output y), 2 GELU-erf (ref = pre-activation), 3 sigmoid (ref = output y).
 * Backward of the activations fused into the forward epilogues / nn.GELU (common_block.py:137). n % 4 == 0. */
int diffsal_act_bwd(const float* dy, const float* ref, float* dx, size_t n, int mode, diffsal_stream_t stream);

/* ---- K16 (training): normalisation layers ------------------------------------------------------
 * Per-channel dual sums over the rows of each segment -> part[segs][chunks][2][C] (fp64), chunks = diffsal_rowstats_chunks():
 *   mode 0 (x, x^2): BatchNorm (train) / GroupNorm statistics;  mode 1/2/3: (dz, dz*xhat) for BN+ReLU, GN+swish, plain.
 * diffsal_reduce_partials + diffsal_norm_finalize_* fold the partials into per-(segment, channel) vectors for the
 * element-wise kernels below.
 * Replace the forward / backward of nn.BatchNorm2d (train), nn.GroupNorm + swish (sal_unet.py:36-44,
 * common_block.py:33-36,196-216). */
int diffsal_rowstats_chunks(int M, int seg_rows);
int diffsal_rowstats(const float* x, const float* dy, const float* y, const float* mu, const float* rs,
                     const float* gamma, const float* beta, double* part, int M, int C, int seg_rows, int mode,
                     int stat_per_seg, diffsal_stream_t stream);
/* Sum fp64 per-workgroup partials part[segs][chunks][width] over the chunks, fixed order -> out[segs][width]
 * (double if out_is_f64, else float).  Every reduction kernel of the training step leaves such partials. */
int diffsal_reduce_partials(const double* part, void* out, int segs, int chunks, int width, int out_is_f64,
                            diffsal_stream_t stream);
/* GroupNorm / train-mode BatchNorm statistics -> the vectors the apply kernels consume.  sums[segs][2][C] = per-channel
 * (sum x, sum x^2); `groups` channel groups share statistics over n values each (BatchNorm: groups = C, segs = 1,
 * n = rows).  Writes mu, rs, scale = rs*gamma, shift = beta - mu*scale, each [segs][C].  BatchNorm extras, all optional:
 * bn_mean / bn_var [C] (batch mean, biased variance) and the nn.BatchNorm2d running-statistics update
 * run = (1-momentum) run + momentum * (mean | var * unbias)  (R/.../common_block.py:196-223 in train mode).
 * norm_finalize_bwd: t[segs][2][C] = (sum dz, sum dz*xhat) -> dbeta, dgamma [C] and k1, k2, k3 [segs][C] for
 * diffsal_norm_bwd_apply. */
int diffsal_norm_finalize_fwd(const double* sums, const float* gamma, const float* beta, float* mu, float* rs,
                              float* scale, float* shift, int segs, int C, int groups, double n, double eps,
                              float* bn_mean, float* bn_var, float* running_mean, float* running_var, float momentum,
                              float unbias, diffsal_stream_t stream);
int diffsal_norm_finalize_bwd(const double* t, const float* gamma, const float* rs, float* dgamma, float* dbeta, float* k1,
                              float* k2, float* k3, int segs, int C, int groups, double n, diffsal_stream_t stream);
/* out = act(x * scale[seg, c] + shift[seg, c]); act: DIFFSAL_ACT_NONE / RELU, or 4 = swish */
int diffsal_affine_act(const float* x, const float* scale, const float* shift, float* out, int M, int C, int seg_rows,
                       int act, diffsal_stream_t stream);
/* dx = k1*dz - k2 - k3*xhat with [seg, C] coefficient tables (dz, xhat as in rowstats modes 1..3) */
int diffsal_norm_bwd_apply(const float* x, const float* dy, const float* y, const float* mu, const float* rs,
                           const float* gamma, const float* beta, const float* k1, const float* k2, const float* k3,
                           float* dx, int M, int C, int seg_rows, int mode, diffsal_stream_t stream);
/* LayerNorm backward: dx (+ add[M, C] when given: the gradient arriving at x over the residual connection around the normalised
 * branch, so the tape needs no separate accumulation pass) and per-block partial (dgamma, dbeta) -> part[blocks][2][C],
 * blocks = diffsal_layernorm_bwd_blocks() */
int diffsal_layernorm_bwd_blocks(int M, int C);
/* n <= 3 tensors of one width in ONE launch (MViT's norm_q / norm_k / norm_v on the pooled tensors, R/models/mvit.py:562-570): host arrays of
 * n device pointers / row counts / eps; backward: part [n][blocks][2][C] with blocks = diffsal_layernorm_bwd_blocks(max M, C), finished by
 * diffsal_reduce_partials(part, out, n, blocks, 2*C, 0) -> [n][dgamma | dbeta]. */
int diffsal_layernorm_multi(const float* const* x, const float* const* gamma, const float* const* beta, float* const* out, const int* M,
                            int n, int C, const float* eps, diffsal_stream_t stream);
int diffsal_layernorm_bwd_multi(const float* const* x, const float* const* dy, const float* const* gamma, float* const* dx, double* part,
                                const int* M, int n, int C, const float* eps, diffsal_stream_t stream);
int diffsal_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* add, float* dx, double* part, int M, int C,
                          float eps, diffsal_stream_t stream);
/* out = x * keep / (1-p), keep from a counter-based hash of (seed, index); same call = its own backward.
 * Replaces nn.Dropout(0.1) of ResnetBlock in train mode (sal_unet.py:109,133). */
int diffsal_dropout(const float* x, float* out, size_t n, float p, uint64_t seed, diffsal_stream_t stream);

/* ---- K16 (training): depthwise projections and attention ---------------------------------------------
 * Plain depthwise k x k convolution on NHWC (w: [k*k][C]) and its two gradients; the training path runs
 * conv_proj_q/k/v (attention.py:36-76) as dwconv + LayerNorm so each half has an exact backward.
 * bwd_weight writes part[k*k][chunks][C], chunks = diffsal_dwconv_bwd_weight_chunks(); the caller sums the chunks. */
int diffsal_dwconv(const float* x, const float* w, float* out, int N, int H, int W, int C, int k, int stride, int pad,
                   diffsal_stream_t stream);
int diffsal_dwconv_bwd_data(const float* du, const float* w, float* dx, int N, int H, int W, int C, int k, int stride,
                            int pad, diffsal_stream_t stream);
int diffsal_dwconv_bwd_weight_chunks(int N, int H, int W, int k, int stride, int pad);
int diffsal_dwconv_bwd_weight(const float* x, const float* du, double* part, int N, int H, int W, int C, int k,
                              int stride, int pad, diffsal_stream_t stream);
/* Backward of diffsal_attention, pass 1: dq [N,Lq,C], plus the softmax P and dS = P (dP - <P,dP>) scale as
 * [N*Lq][ld] rows (column = head*Lk + t, ld >= heads*Lk, columns beyond heads*Lk untouched).  Pass 2 (dk, dv) is
 * diffsal_wgrad_segmented on (dS, q) and (P, dout): no atomics, deterministic. */
int diffsal_attention_bwd_blocks(int Lq, int C, int heads);
int diffsal_attention_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* p_out,
                          float* ds_out, int N, int Lq, int Lk, int C, int heads, int ld, float scale,
                          diffsal_stream_t stream);

/* ---- K16 (training): remaining backward kernels ----------------------------------------------------
 * adjoint of diffsal_resize_bilinear (dy [N,H,W,C] -> dx [N,h,w,C]); backward of pack_frames for the visual
 * features (frames [B,Tin,hw,C] -> NCTHW [B,C,Tv,hw]); head backward (dy = dpre w, part[blocks][C+1] holds
 * sum dpre*y and sum dpre); conv_in parameter gradients (part[10][chunks][C]: 9 taps + bias); dense_small backward. */
/* resize_bilinear_bwd: with ws >= N*h*W*C*4 bytes the adjoint runs as two one-axis passes (rows into ws, then
 * columns); ws may be NULL (single joint pass, much slower for large up-factors). */
int diffsal_resize_bilinear_bwd(const float* dy, float* dx, int N, int h, int w, int H, int W, int C, void* ws,
                                size_t ws_bytes, diffsal_stream_t stream);
int diffsal_unpack_frames(const float* frames, float* vis_grad, int B, int C, int Tv, int Tin, int hw,
                          diffsal_stream_t stream);
int diffsal_head_bwd(const float* y, const float* w, const float* s_out, const float* ds, float* dy, double* part,
                     int blocks, int M, int C, diffsal_stream_t stream);
int diffsal_conv_in_bwd(const float* x, const float* dy, double* part, int B, int H, int W, int C, int chunks,
                        diffsal_stream_t stream);
int diffsal_dense_small_bwd(const float* in, const float* w, const float* dout, float* dw, float* db, float* din,
                            int B, int K, int N, int swish_in, diffsal_stream_t stream);

/* backward of diffsal_audio_fuse: dout [B,C,T,H,W] -> dx [B,T,H,W,C] (frames layout) and the audio-map gradient as
 * per-output-row partials part[((b*T+t)*h + ys)*w + xs][dyr][c] (up = H/h rows per audio row, up <= 8): da_small is
 * diffsal_colsum(part viewed [B*T*h*w*up, C], seg_rows = up); with up == 1 part IS da_small [B*T, h*w, C]. */
int diffsal_audio_fuse_bwd(const float* a_small, const float* x, const float* dout, float* dx, float* part, int B,
                           int T, int H, int W, int C, int h, int w, diffsal_stream_t stream);

/* ---- K6: frame packing: visual features NCTHW[B,C,Tv,h,w] + noise NHWC[B,h,w,C] ->
 * NHWC frames [B,Tv+1,h,w,C] with the noise map as the LAST frame (quirk Q2).
 * Replaces torch.cat(dim=2) + rearrange().contiguous(), R/.../sal_unet.py:311-317,
 * transformer.py:273-279. noise may be NULL (then only Tv frames of a [B,Tout,...] buffer are written). */
int diffsal_pack_frames(const float* vis /*fp32: the module's input contract*/, const void* noise, void* out, int B,
                        int C, int Tv, int Tout, int hw, int dtype, diffsal_stream_t stream);
/* The same for up to four (vis, noise, out) triples of one batch size and storage type in ONE launch: the frame tensors of all
 * decoder stages (sal_unet.py:414-441 builds them stage by stage). */
int diffsal_pack_frames_multi(const float* const* vis, const void* const* noise, void* const* out, int n, int B, const int* C,
                              const int* Tv, const int* Tout, const int* hw, int dtype, diffsal_stream_t stream);


/* ---- bilinear resize, align_corners=False, NHWC ----------------------------------------
 * R/.../common_block.py:197 (nn.Upsample x2), sal_unet.py:325-327. */
int diffsal_resize_bilinear(const void* in, void* out, int N, int h, int w, int H, int W, int C, int dtype,
                            diffsal_stream_t stream);

/* out[n,Y,X,c] = sum_i bilinear(in_i[n, h_i, w_i, c] -> (H,W)), summed in order i = 0..n_in-1.
 * Replaces the per-stage F.interpolate + "+=" of R/.../sal_unet.py:482-487. n_in <= 4. */
int diffsal_resize_sum(const void* const* ins /*host array of device ptrs*/, const int* hs, const int* ws,
                       int n_in, void* out, int N, int H, int W, int C, int dtype, diffsal_stream_t stream);

/* ---- K7: audio fusion (after the align 1x1 conv) ----------------------------------------
 * R/.../transformer.py:133-146.  a_small: [B*T, h*w, C] tokens; x: NHWC frames [B,T,H,W,C];
 * out: contiguous [B,C,T,H,W] (the reference layout, which the caller then *reinterprets* as
 * [B*T, H*W, C] tokens -- quirk Q5).  up = H / h (nearest), 1 when no upsample. */
int diffsal_audio_fuse(const void* a_small, int a_ld /* elements between a_small's rows (>= C): the four stages' align
                       convolutions run as ONE product whose output rows hold all their channels side by side */,
                       const void* x, void* out, int B, int T, int H, int W, int C,
                       int h, int w, int dtype, diffsal_stream_t stream);

/* ---- K8: LayerNorm over C on tokens [M,C] ------------------------------------------------ */
int diffsal_layernorm(const void* x, const float* gamma, const float* beta, void* out, int M, int C,
                      float eps, int dtype, diffsal_stream_t stream);

/* ---- K9: depthwise projections + LayerNorm ----------------------------------------------
 * q: depthwise 3x3 (pad 1) on NHWC [N,H,W,C] then LN  -> [N, H*W, C].   R/.../attention.py:36-47,94
 * w9: [9][C] (centre temporal slice of the Conv3d weight, quirk Q8). */
int diffsal_dwconv3_ln(const void* x, const float* w9, const float* gamma, const float* beta, void* out,
                       int N, int H, int W, int C, float eps, int dtype, diffsal_stream_t stream);
/* k and v: depthwise kxk stride k (no pad) then LN -> [N, gh*gw, C] each; xk may differ from xv
 * (audio-fused K, attention.py:88-92).  wk, wv: [k*k][C]. */
int diffsal_dwpool_ln_kv(const void* xk, const void* xv, const float* wk, const float* wv,
                         const float* gk, const float* bk, const float* gv, const float* bv,
                         void* out_k, void* out_v, int N, int H, int W, int C, int k, float eps, int dtype,
                         diffsal_stream_t stream);
/* diffsal_dwconv3_ln (query branch) and diffsal_dwpool_ln_kv (key / value branch) of one transformer block in ONE launch: both
 * read the block's normalised frames and neither depends on the other (attention.py:86-95).  Same results as the two entries.
 * pre_gamma / pre_beta (or NULL): the block's own LayerNorm (transformer.py:110, x = self.norm(x)) is applied to every token as
 * it is loaded -- xq and xv are then the UN-normalised frames, xk too when pre_ln_k != 0 (visual-only: the key input is the same
 * normalised tensor; with audio it is the fused audio map, pre_ln_k = 0) -- so the normalised tensor is never written.  Bit-equal
 * with diffsal_layernorm followed by the plain form (the normalised value is rounded through the storage type). */
int diffsal_qkv_prep(const void* xq, const float* w9, const float* gq, const float* bq, void* out_q, const void* xk,
                     const void* xv, const float* wk, const float* wv, const float* gk, const float* bk, const float* gv,
                     const float* bv, void* out_k, void* out_v, int N, int H, int W, int C, int k, float eps,
                     const float* pre_gamma, const float* pre_beta, float pre_eps, int pre_ln_k, int dtype,
                     diffsal_stream_t stream);


/* ---- K8 + K10 fused for the finest stage (C = 96, hidden = 192: both MLP weights fit in LDS):
 *   x2 = x1 + fc2(gelu_erf(fc1(LayerNorm(x1; g2, be2, eps2)) + b1)) + b2 ... i.e. transformer.py:153-157's
 *        `x = x + mlp(norm2(x))` with Mlp of common_block.py:125-147;
 *   z  = LayerNorm(x2; gz, bez, epsz) (sal_unet.py:447,473 norm_mts), written only for tokens of frames < t_keep
 *        (token m belongs to frame (m / hw) % T; ReduceTemp reads frames 0..4 only) -- z may be NULL.
 * x1, x2, z: [M, C] fp32 tokens; w1 [hidden][C], w2 [C][hidden] as stored by nn.Linear.  One launch instead of four; the
 * hidden activations never leave registers.  x2 must not alias x1. */
int diffsal_mlp_block(const float* x1, const float* g2, const float* be2, float eps2, const float* w1, const float* b1,
                      const float* w2, const float* b2, float* x2, float* z, const float* gz, const float* bez, float epsz,
                      long M, int C, int hidden, int hw, int T, int t_keep, diffsal_stream_t stream);

/* The same stage on bf16 / fp16 storage: all three weight matrices fit in LDS, so proj + residual is fused in as well:
 *   x1 = x + o wp^T + bp;  x2 = x1 + fc2(gelu(fc1(LayerNorm(x1))));  z = LayerNorm(x2) on frames < t_keep
 * (attention.py:110, transformer.py:151-157, sal_unet.py:447).  o, x, wp, w1, w2, x2, z: 16-bit (dtype); vectors fp32. */
int diffsal_block16(const void* o, const void* x, const void* wp, const float* bp, const float* g2, const float* be2, float eps2,
                    const void* w1, const float* b1, const void* w2, const float* b2, void* x2, void* z, const float* gz,
                    const float* bez, float epsz, long M, int C, int hidden, int hw, int T, int t_keep, int dtype,
                    diffsal_stream_t stream);

/* ---- K11: attention core ------------------------------------------------------------------
 * o[n,l,:] = concat_h softmax_t( q[n,l,h,:] . k[n,t,h,:] * scale ) v[n,t,h,:],  Lk <= 32.
 * R/.../attention.py:97-108 (scale = C^-0.5, quirk Q6). */
int diffsal_attention(const void* q, const void* k, const void* v, void* o, int N, int Lq, int Lk,
                      int C, int heads, float scale, int dtype, diffsal_stream_t stream);

/* ---- K14 tail: 1x1 conv C->1 + sigmoid on NHWC -> [N,H,W] --------------------------------
 * R/.../common_block.py:111-122. */
int diffsal_head_sigmoid(const void* x, const float* w /*[C]*/, const float* bias /*[1]*/, float* out /*fp32*/,
                         int NHW, int C, int dtype, diffsal_stream_t stream);

/* The pooled key / value branch of a TransformerBlock in ONE launch, projections included (C = 96 / 192; attention.py:49-76,
 * 79-80, 88-99): per pooled token  k = Linear_k(LN_k(pool_k(LN_1?(xk)))),  v = Linear_v(LN_v(pool_v(LN_1(xv)))),  pool = depthwise
 * k x k stride-k convolution (weights wk / wv [k*k, C] fp32), LN_1 = the block's first LayerNorm applied as the tokens are loaded
 * (to xv always, to xk when pre_ln_k), proj_w* [C, C] in the storage type.  out_k, out_v [N, gh*gw, C]. */
int diffsal_kv_prep_proj(const void* xk, const void* xv, const float* wk, const float* wv, const float* gk, const float* bk,
                         const float* gv, const float* bv, const void* proj_wk, const float* proj_bk, const void* proj_wv,
                         const float* proj_bv, void* out_k, void* out_v, int N, int H, int W, int C, int k, float eps,
                         const float* pre_gamma, const float* pre_beta, float pre_eps, int pre_ln_k, int dtype,
                         diffsal_stream_t stream);

/* ---- fused first half of a TransformerBlock (C = 96, 2 heads; csrc/tblock.hip) --------------------------------------------
 * R/models/saliency_decoder/transformer.py:150-152, attention.py:36-47,87-110:
 *   xn = LayerNorm(x; g1, b1, eps1);  q = Linear_q( LayerNorm(dwconv3x3(xn; w9); gq, bq, epsq) );
 *   o  = softmax(q k^T * scale) v per head, with the block's PROJECTED keys / values k, v [N, Lk <= 32, C];
 *   fp32 storage:   out = x + Linear_p(o)      (x1; the rest of the block is diffsal_mlp_block)
 *   16-bit storage: out = o                    (the rest of the block is diffsal_block16)
 * x, out [N, H, W, C] channels-last in the storage type `dtype`; wq, wp [C, C] in the storage type, row = output feature; w9
 * [9, C] fp32, tap = 3 ky + kx (the centre temporal slice of the Conv3d weight, zero padding 1).  One persistent launch per
 * call instead of LayerNorm, depthwise-q + LayerNorm, the proj_q GEMM, the attention core and the proj GEMM. */
int diffsal_block_front(const void* x, const void* k, const void* v, const float* g1, const float* b1, float eps1,
                        const float* w9, const float* gq, const float* bq, float epsq, const void* wq, const float* bias_q,
                        const void* wp, const float* bias_p, void* out, int N, int H, int W, int C, int Lk, int heads,
                        float scale, int dtype, diffsal_stream_t stream);

/* ---- storage-type conversion: dst[i] = (dst type) src[i], round to nearest even.  Used once per parameter version to
 * put packed convolution / linear weights into the 16-bit storage type of a reduced-precision module (the reference
 * has no counterpart: it is fp32-only, R/diffusion_trainer.py:212-218). */
int diffsal_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, diffsal_stream_t stream);

/* ==== once-per-clip encoders around the denoiser =======================================================
 * ---- general softmax attention on the fp32 matrix cores (flash-style, one pass over the keys) -------------------------
 * out[b, l, h*DV + :] = softmax_t( scale * q[b,h,l,:] . k[b,h,t,:]  +  q_extra[b,h,l,:] . k_extra[t,:] ) v[b,h,t,:]
 *                       (+ residual[b,h,l,:], not on row 0 if skip_first)
 * q / k / v / residual are addressed through (batch, head, row) element strides, so slices of a fused qkv GEMM output
 * are read in place; q_extra [B,H,Lq,E] and k_extra [Lk,E] are contiguous and carry an additive attention bias as E
 * extra contraction columns (MViT: E = 48 or 32, see diffsal_relpos_project; otherwise E = 0 and both are NULL).
 * k_slots (optional, E > 0): the slot form of a ONE-HOT k_extra -- k_slots[t][0..2] = the columns of row t that are 1 (E = none, e.g. the
 * class token; [3] unused), 16-byte aligned.  With it the bias is three gathered q_extra values per (query, key) added on the vector unit
 * and the E extra contraction columns are not multiplied (a fifth fewer matrix instructions at E = 48); k_extra must still describe the
 * same matrix (the backward contracts with it).  The result differs from the contraction form by summation order only.
 * Built (D, E, DV): (96,48,96), (96,32,96), (96,0,96), (64,0,64), (32,0,32).  Replaces
 *   R/models/mvit.py:587-605 (MultiScaleAttention: attn = (q*scale) k^T, add_decomposed_rel_pos, softmax, attn v, + q),
 *   R/models/audio_attention.py:50-58 (dots, softmax, out). */
int diffsal_attention_general(const float* q, const float* q_extra, const float* k, const float* k_extra,
                              const int* k_slots /*[Lk][4] or NULL*/, const float* v, const float* residual, float* out, float* lse /*[B,H,Lq] row log-sum-exp, or NULL*/, int B,
                              int H, int Lq, int Lk, int D, int E, int DV, const long* q_strides /*host [3]*/,
                              const long* k_strides, const long* v_strides, const long* r_strides, float scale,
                              int skip_first, float* tail_ws, size_t tail_ws_floats, diffsal_stream_t stream);
/* tail_ws (optional): tail_ws_floats >= diffsal_attention_general_tail_floats(...) floats (a smaller buffer is ignored).  With it the blocks of the last, partly filled
 * round of workgroups are cut into pieces over the keys and a finishing launch merges the pieces' outputs by their
 * log-sum-exps; without it (NULL, or the function returned 0) every block runs whole. */
size_t diffsal_attention_general_tail_floats(int B, int H, int Lq, int Lk, int DV);
/* Backward (training of the encoders): P is recomputed from `lse`; delta_ws [B,H,Lq] is scratch.  Outputs are contiguous
 * head-major tensors: dq [B,H,Lq,D] (includes the residual path's dO when `residual` was given), dq_extra [B,H,Lq,E]
 * (NULL iff E == 0), dk [B,H,Lk,D], dv [B,H,Lk,DV]; k_extra is a constant table and gets no gradient.  No atomics. */
int diffsal_attention_general_bwd_splits(int B, int H, int Lq, int Lk); /* S; kv_part_ws needs S*B*H*Lk*(D+DV) floats if S > 1 */
/* q_tail_ws (optional): q_tail_ws_floats >= diffsal_attention_general_bwd_qtail_floats(...) floats (a smaller buffer is ignored).  With it the dq kernel cuts the blocks of its
 * last, partly filled round of workgroups into pieces over the keys and a finishing launch adds their partial rows (fixed
 * order); without it (NULL, or the function returned 0) every block runs whole. */
size_t diffsal_attention_general_bwd_qtail_floats(int B, int H, int Lq, int Lk, int D, int E);
/* ds_ws (optional): ds_ws_floats >= diffsal_attention_general_bwd_ds_floats(...) floats (NULL or a smaller buffer: ignored).  With it the dk / dv kernel leaves
 * dS [B*H][Lq][Lk rounded up to 32] there and dq is the one product dS K' (five tile products per query / key tile pair instead of seven: the recomputation of
 * S and dP in the dq kernel is traded for one write and one read of dS); same sums in the same order, dq bit-identical to the recomputing form. */
size_t diffsal_attention_general_bwd_ds_floats(int B, int H, int Lq, int Lk);
int diffsal_attention_general_bwd(const float* q, const float* q_extra, const float* k, const float* k_extra, const float* v,
                                  const float* residual, const float* out, const float* lse, const float* dout,
                                  float* delta_ws, float* kv_part_ws, float* q_tail_ws, size_t q_tail_ws_floats, float* ds_ws,
                                  size_t ds_ws_floats, float* dq, float* dq_extra, float* dk, float* dv, int B, int H, int Lq,
                                  int Lk, int D, int E, int DV, const long* q_strides, const long* k_strides,
                                  const long* v_strides, const long* r_strides, float scale, int skip_first,
                                  diffsal_stream_t stream);

/* ---- MViTv2 pieces that are not GEMMs / LayerNorms (R/models/mvit.py) -------------------------------------------------
 * im2col3d: columns of PatchEmbed3D's Conv3d (mvit.py:157-163, 983-989): x [B,C,T,H,W] -> cols [B*To*Ho*Wo][Kp],
 *   k = (c, kt, ky, kx) as in weight.reshape(Cout, -1), zero-padded to Kp (multiple of 32); the projection is one GEMM.
 * pool3d_ln: attention_pool (mvit.py:446-494): depthwise Conv3d 3x3x3 pad 1 stride (st,sh,sw) over the video tokens of
 *   every head + LayerNorm(D); class token (row 0) skips the conv.  in element (b, n, head, d) at
 *   in + b*in_stride_b + n*in_stride_n + head*D + d; w27 [27][D]; out [B, heads, 1 + To*Ho*Wo, D].
 * maxpool_tokens: the skip path's MaxPool3d (mvit.py:765-777) on tokens [B, 1+T*H*W, C], padding k/2, class token kept.
 * relpos_project: per query the E bias columns of add_decomposed_rel_pos (mvit.py:363-410), E = 48 or 32:
 *   E = 48: [0,kt) q.Rt[t], [8,8+kh) q.Rh[y], [24,24+kw) q.Rw[x]   (kt <= 8, kh <= 16, kw <= 24)
 *   E = 32: [0,kt) q.Rt[t], [8,8+kh) q.Rh[y], [16,16+kw) q.Rw[x]   (kt <= 8, kh <= 8, kw <= 16: every MViTv2-S stage at 224x384)
 *   rest 0, class-token row 0; Rt [qt][kt][D] etc. are the gathered tables (resize_decomposed_rel_pos, :330-361);
 *   q [BH, 1+qt*qh*qw, D] unscaled.  The key side of diffsal_attention_general carries one-hot rows in the same layout.
 * tokens_to_channels_first: [B, off+L, C] rows off.. -> [B, C, L] (the NCTHW maps MViT returns, :1128-1134). */
int diffsal_im2col3d(const float* x, float* cols, int B, int C, int T, int H, int W, int KT, int KH, int KW, int st, int sh,
                     int sw, int pt, int ph, int pw, int Kp, diffsal_stream_t stream);
int diffsal_pool3d_ln(const float* in, const float* w27, const float* gamma, const float* beta, float* out, int B, int heads,
                      int D, int T, int H, int W, int st, int sh, int sw, long in_stride_b, long in_stride_n, float eps,
                      diffsal_stream_t stream);
int diffsal_maxpool_tokens(const float* in, float* out, int B, int C, int T, int H, int W, int kt, int kh, int kw, int st,
                           int sh, int sw, diffsal_stream_t stream);
int diffsal_relpos_project(const float* q, const float* Rt, const float* Rh, const float* Rw, float* extra, int BH, int D,
                           int qt, int qh, int qw, int kt, int kh, int kw, int E, diffsal_stream_t stream);
int diffsal_tokens_to_channels_first(const float* in, float* out, int B, int C, int L, int off, diffsal_stream_t stream);
/* Training of the video encoder: backward of the pieces above (all gather form, no atomics, fixed summation order).
 * pool3d_ln with gamma = beta = NULL is the convolution alone (its LayerNorm then runs as diffsal_layernorm).
 * pool3d_bwd_data writes d_in through the forward input's strides (straight into the q / k / v slice of the qkv gradient);
 * pool3d_bwd_weight leaves part[chunks][27][D] doubles (chunks = diffsal_pool3d_bwd_weight_chunks()) for
 * diffsal_reduce_partials.  maxpool_tokens_idx also records the arg-max token (first maximum in (t,y,x) order, as
 * torch.nn.MaxPool3d); maxpool_tokens_bwd routes dy back through it.  relpos_project_bwd: dq (+)= dextra . R and per-chunk
 * partials of the three table gradients (chunks = diffsal_relpos_project_bwd_chunks(); finish with
 * diffsal_reduce_partials(part, out, 1, chunks, width, 0) -> dRt | dRh | dRw back to back). */
int diffsal_pool3d_bwd_data(const float* dy, const float* w27, float* din, int B, int heads, int D, int T, int H, int W, int st,
                            int sh, int sw, long in_stride_b, long in_stride_n, diffsal_stream_t stream);
int diffsal_pool3d_bwd_weight_chunks(void);
int diffsal_pool3d_bwd_weight(const float* in, const float* dy, double* part, int B, int heads, int D, int T, int H, int W,
                              int st, int sh, int sw, long in_stride_b, long in_stride_n, diffsal_stream_t stream);
int diffsal_maxpool_tokens_idx(const float* in, float* out, int* idx, int B, int C, int T, int H, int W, int kt, int kh, int kw,
                               int st, int sh, int sw, diffsal_stream_t stream);
int diffsal_maxpool_tokens_bwd(const float* dy, const int* idx, float* din, int B, int C, int T, int H, int W, int kt, int kh,
                               int kw, int st, int sh, int sw, diffsal_stream_t stream);
/* qkv_pool: the three attention_pool convolutions of one block in ONE launch, head dimension D = 96 (mvit.py:446-494, called
 * three times per block at :556-590): qkv is the fused projection [B][1+T*H*W][3][heads][96]; w27 / gamma / beta / eps / out
 * are arrays of three (q, k, v); gamma = beta = eps = NULL: convolutions only.  out[x]: [B*heads][1 + To*Ho*Wo][96] with the
 * strides of q (stride_q[3]) or of k and v (stride_kv[3]).  Bit-identical to three diffsal_pool3d_ln calls in the convolution;
 * the LayerNorm sums run over 8 lanes x 12 channels instead of 32 x 4.
 * qkv_pool_bwd_data: every element of dqkv [B][N][3][heads][96] from the three output gradients (temporal stride 1, equal
 * spatial strides); bit-identical to three diffsal_pool3d_bwd_data calls. */
int diffsal_qkv_pool(const void* qkv, const float* const* w27, const float* const* gamma, const float* const* beta,
                     const float* eps, float* const* out, int B, int heads, int D, int T, int H, int W, const int* stride_q,
                     const int* stride_kv, int dtype /* storage type of qkv; outputs are fp32 */,
                     int w_channel_major /* 1: filters in the parameter's [96][27] layout, 0: tap-major [27][96] */,
                     diffsal_stream_t stream);
int diffsal_qkv_pool_bwd_data(const float* const* dy, const float* const* w27, float* dqkv, int B, int heads, int D, int T,
                              int H, int W, const int* stride_q, const int* stride_kv, int w_channel_major,
                              diffsal_stream_t stream);
/* qkv_pool_bwd_weight: the three filter gradients in one launch: part[3][chunks][27*96] doubles (chunks =
 * diffsal_qkv_pool_bwd_weight_chunks(B, heads, T, H, W, stride_q): 512, or 170 on small stages), finished by diffsal_reduce_partials(part, out, 3, chunks, 27*96, 0). */
int diffsal_qkv_pool_bwd_weight_chunks(int B, int heads, int T, int H, int W, const int* stride_q);
int diffsal_qkv_pool_bwd_weight(const float* qkv, const float* const* dy, double* part, int B, int heads, int D, int T, int H,
                                int W, const int* stride_q, const int* stride_kv, int w_channel_major /* layout of each [27*96] row */,
                                diffsal_stream_t stream);
/* rel_tables: the three gathered relative-position tables of a block (resize_decomposed_rel_pos, mvit.py:330-361) as sparse
 * row maps built once per grid on the host: out_t[m] = w2_t[m][0] * rel_t[idx2_t[m][0]] + w2_t[m][1] * rel_t[idx2_t[m][1]],
 * m < M[t]; arrays of three (t, h, w).  rel_tables_bwd applies the transposed maps in CSR form (row starts [R[t] + 1],
 * columns = m, weights): drel_t[r] = sum of w * dout_t[m] in CSR order. */
int diffsal_rel_tables(const float* const* rel, const int* const* idx2, const float* const* w2, float* const* out, const int* M,
                       int D, diffsal_stream_t stream);
int diffsal_rel_tables_bwd(const float* const* dout, const int* const* csr_ptr, const int* const* csr_col,
                           const float* const* csr_w, float* const* drel, const int* R, int D, diffsal_stream_t stream);
int diffsal_relpos_project_bwd_chunks(void);
int diffsal_relpos_project_bwd(const float* dextra, const float* q, const float* Rt, const float* Rh, const float* Rw, float* dq,
                               int accumulate, double* part /*[chunks][(qt*kt + qh*kh + qw*kw) * D]: dRt | dRh | dRw partials*/,
                               int BH, int D, int qt, int qh, int qw, int kt, int kh, int kw, int E, diffsal_stream_t stream);

/* ---- legacy DDPM-style UNet of R/models/diffusion_decoder/diffusion.py (DiffusionModel :197-357; not instantiated by
 * any configuration of the reference, kept for completeness of the models/diffusion_decoder surface), fp32 NHWC.  Its
 * ResnetBlocks / Downsample / conv_out reuse diffsal_conv_igemm and K3; what it adds:
 *   groupnorm         GroupNorm with the activation optional (act 0: AttnBlock.norm :145,173; act 1 == groupnorm_swish)
 *   softmax_rows      out[r,:] = softmax(scale * x[r,:]) over the HW keys of the full self-attention (:166-168); the two
 *                     products around it are diffsal_conv_igemm calls with k / v^T of one image as the weight operand
 *   upsample_nearest2 Upsample :46-47;  avgpool2: Downsample without conv :69;  sigmoid_gate: feat_interact :318 */
int diffsal_groupnorm(const void* x, const float* gamma, const float* beta, void* out, int B, int HW, int C, int groups,
                      float eps, int act, void* ws, size_t ws_bytes, int dtype, diffsal_stream_t stream);
int diffsal_softmax_rows(const float* x, float* out, long rows, int cols, float scale, diffsal_stream_t stream);
int diffsal_upsample_nearest2(const float* in, float* out, int N, int H, int W, int C, diffsal_stream_t stream);
int diffsal_avgpool2(const float* in, float* out, int N, int H, int W, int C, diffsal_stream_t stream);
int diffsal_sigmoid_gate(const float* y, const float* x, float* out, long n, diffsal_stream_t stream);

/* ---- VGGish feature stack (R/models/vggish.py:70-106): 3x3 convs are diffsal_conv_in (1 input channel, act = ReLU) and
 * diffsal_conv_igemm (bias + ReLU epilogue); this is its MaxPool2d(k, stride) on NHWC (no padding, floor). */
int diffsal_maxpool2d(const void* in, void* out, int N, int H, int W, int C, int k, int stride, int dtype,
                      diffsal_stream_t stream);

/* ---- evaluation metrics on the device: CC, SIM, NSS, KL-div of predicted vs ground-truth saliency maps -----------
 * R/models/sal_losses.py:14-37 (nss2), :63-97 (cc_s2), :100-131 (kldiv2), :134-176 (normalize_map2, similarity2), as
 * called per validation batch by get_kl_cc_sim_loss_wo_weight (R/diffusion_trainer.py:741,797,868).
 * pred, gt: [B][n] fp32 (n = T*H*W values per image).  per_image (optional): [B][4] = (cc, sim, nss, kl);
 * mean_out: [4] batch means in the same order (what the reference functions return).  Two streaming passes, fp64
 * accumulation in a fixed order; ws >= diffsal_saliency_metrics_ws_bytes(B) bytes. */
size_t diffsal_saliency_metrics_ws_bytes(int B);
/* Training: gradient of  w_cc CC + w_sim SIM + w_nss NSS + w_kl KL  (the four batch means above) with respect to pred, from the
 * workspace the forward call left behind; weights4 = four floats on the device (upstream gradients x the configuration's loss
 * weights).  The differentiable terms of get_kl_cc_sim_loss / get_lossv2 (R/models/sal_losses.py:179-259). */
int diffsal_saliency_metrics_bwd(const float* pred, const float* gt, int B, long n, const void* ws, size_t ws_bytes,
                                 const float* weights4, float* dpred, diffsal_stream_t stream);
int diffsal_saliency_metrics(const float* pred, const float* gt, int B, long n, void* ws, size_t ws_bytes,
                             float* per_image, float* mean_out, diffsal_stream_t stream);

/* ---- K15: sampler elementwise update  out = a*x + b*y + c*z  (y, z may be NULL) -----------
 * scalar-coefficient axpys of R/diffusion_trainer.py:459-478 and R/models/dpm_solver/sampler.py:548-593,816-853. */
int diffsal_axpbypcz(const float* x, const float* y, const float* z, float a, float b, float c, float* out,
                     size_t n, diffsal_stream_t stream);

/* ---- K14 tail + K15 fused: the end of a denoising step in one pass over the sampler state.
 *   x0 = bilinear(s_low [N,h,w] -> [N,H,W])          final resize of the sigmoid map, R/.../sal_unet.py:325-327
 *   m  = ex * x + e0 * x0                            wrapper's model-output conversion (x_start -> noise:
 *                                                    ex = 1/sigma_t, e0 = -alpha_t/sigma_t, R/models/dpm_solver/sampler.py:286-292;
 *                                                    ex = 0, e0 = 1 keeps x0)
 *   x_next = A * x + c0 * m + c1 * m_prev            multistep update, sampler.py:548-593, 797-853 (m_prev NULL: order 1)
 * x0_out and x_next may be NULL.  Same per-element arithmetic as diffsal_resize_bilinear + diffsal_axpbypcz (bit-equal). */
int diffsal_resize_update(const float* s_low, const float* x, const float* m_prev, float* x0_out, float* m_out, float* x_next,
                          int N, int h, int w, int H, int W, float ex, float e0, float A, float c0, float c1,
                          diffsal_stream_t stream);

/* ---- K16 tail: loss, gradient clipping and the optimizer, on flat fp32 buffers -------------
 * diffsal_reduce_blocks(): number of doubles the `part` scratch of the two reductions below must hold.
 *
 * mse_loss: loss[0] = loss_scale * sum (pred - target)^2 and, if dpred != NULL, dpred = 2 loss_scale (pred - target).
 *   With loss_scale = mse_weight / batch this is `mse_weight * (pred-gt).square().sum(dim=(1,2,3)).mean(dim=0)`
 *   (R/models/sal_losses.py:189-192) and its gradient.  n % 4 == 0.
 * grad_norm: norm[0] = gscale * ||g||_2 over the flat gradient buffer -- total_norm of
 *   torch.nn.utils.clip_grad_norm_ (R/diffusion_trainer.py:228-233); gscale = 1/world_size folds in the DDP mean.
 * adam_step: torch.optim.Adam.step (R/util/utils.py:116-123, R/diffusion_trainer.py:235; amsgrad=false) on
 *   p, m (exp_avg), v (exp_avg_sq), with the gradient first multiplied by
 *   gscale * min(1, max_norm / (norm[0] + 1e-6)) (norm may be NULL or max_norm <= 0: no clipping).  `step` counts
 *   from 1.  store_clipped_grad != 0 writes the scaled gradient back into g as clip_grad_norm_ does.
 * scale_by: out = x * s[0] with s a device scalar (the incoming d(loss) of the MSE backward).  n % 4 == 0.
 * multi_copy: dst[dst_offsets[i] .. +sizes[i]) = srcs[i][0 .. sizes[i]) for i < count, a few launches for any count
 *   (gathers the per-parameter gradients autograd produced into the flat gradient buffer; the three tables are HOST
 *   arrays, the pointers in them device pointers; dst_offsets multiples of 4 elements).
 * Reductions are fp64 with a fixed order: a step is bit-reproducible.  No host synchronisation. */
int diffsal_reduce_blocks(void);
int diffsal_multi_copy(const float* const* srcs, const long* dst_offsets, const long* sizes, int count, float* dst,
                       diffsal_stream_t stream);
int diffsal_scale_by(const float* x, const float* s, float* out, long n, diffsal_stream_t stream);
int diffsal_mse_loss(const float* pred, const float* target, float* dpred, float* loss, double* part, long n,
                     float loss_scale, diffsal_stream_t stream);
int diffsal_grad_norm(const float* g, long n, float gscale, float* norm, double* part, diffsal_stream_t stream);
int diffsal_adam_step(float* p, float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int step, float gscale, const float* norm, float max_norm,
                      int store_clipped_grad, diffsal_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSAL_H */
