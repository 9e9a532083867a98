// Small / HBM-bound kernels of the SalUNet step: timestep-embedding MLP, conv_in, frame packing,
// bilinear resizes, audio fusion, the pooled-KV attention core, the sigmoid head and the sampler axpy.
#include "common.h"

#include <cstdlib>
#include <cstring>

namespace diffsal {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kernel[160] = "";
void note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

static const char* const kTuneNames[TUNE_COUNT] = {
    "DIFFSAL_NO_PERSIST", "DIFFSAL_NO_XCD_ORDER", "DIFFSAL_NO_HALO", "DIFFSAL_FORCE_HALO", "DIFFSAL_IGEMM_CFG",
    "DIFFSAL_IGEMM16_CFG", "DIFFSAL_PLAN_DEBUG", "DIFFSAL_WGRAD_CFG", "DIFFSAL_WGRAD_SPLITS", "DIFFSAL_WGRAD_VERBOSE",
    "DIFFSAL_NO_FUSED_BLOCK", "DIFFSAL_NO_WINOGRAD", "DIFFSAL_FORCE_WINOGRAD", "DIFFSAL_GN_CHUNKS", "DIFFSAL_GN_APPLY_WGS",
    "DIFFSAL_GEMM_DMA", "DIFFSAL_CONV_DMA", "DIFFSAL_GROUP_GRID", "DIFFSAL_GEMM_DMA16", "DIFFSAL_WGRAD_DMA", "DIFFSAL_NO_GN_SLAB", "DIFFSAL_NO_WINOGRAD4", "DIFFSAL_BATCH_TILE", "DIFFSAL_BATCH_XCD", "DIFFSAL_NO_TAPSUM_ROWS", "DIFFSAL_TAPSUM_ROWS_FORM",
    "DIFFSAL_NO_ATTN_BWD_DS", "DIFFSAL_NO_POOL_RUNS", "DIFFSAL_NO_ATTN_SLOTS", "DIFFSAL_NO_STREAM16", "DIFFSAL_CONV16_TILE", "DIFFSAL_BLOCK16_WAVES", "DIFFSAL_NO_ATTN16_MFMA", "DIFFSAL_CONV16_HALF"};
static int g_tune[TUNE_COUNT];
static const bool g_tune_init = [] {
  for (int k = 0; k < TUNE_COUNT; ++k) {
    const char* e = getenv(kTuneNames[k]);
    g_tune[k] = (e && e[0]) ? atoi(e) : -1;
  }
  return true;
}();
int tune(int key) { return (key >= 0 && key < TUNE_COUNT) ? __atomic_load_n(&g_tune[key], __ATOMIC_RELAXED) : -1; }

// ------------------------------------------------------------------------------------------------
// K1: sinusoidal embedding + dense0 + swish (this kernel), then dense1 through dense_small_kernel.
// R/models/saliency_decoder/sal_unet.py:15-33, :304-307.  One wavefront per output row: coalesced
// weight-row read + 64-lane butterfly; every workgroup rebuilds the tiny [B, ch] embedding in LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void temb_dense0_kernel(const void* __restrict__ t, int t_is_f32, int B, int ch,
                                                          const float* __restrict__ freq, const float* __restrict__ w0,
                                                          const float* __restrict__ b0, float* __restrict__ hid) {
  extern __shared__ float sh[];  // emb[B][ch]
  const int half = ch / 2, tc = 4 * ch;
  for (int i = threadIdx.x; i < B * half; i += 256) {
    const int b = i / half, j = i - b * half;
    const float tv = t_is_f32 ? static_cast<const float*>(t)[b]
                              : static_cast<float>(static_cast<const long long*>(t)[b]);
    const float a = tv * freq[j];
    sh[b * ch + j] = sinf(a);
    sh[b * ch + half + j] = cosf(a);
  }
  if ((ch & 1) && threadIdx.x < B) sh[threadIdx.x * ch + ch - 1] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= tc) return;
  for (int b = 0; b < B; ++b) {
    float s = 0.f;
    for (int k = lane; k < ch; k += 64) s = fmaf(w0[r * ch + k], sh[b * ch + k], s);
    s = group_sum<64>(s);
    if (lane == 0) hid[static_cast<long>(b) * tc + r] = swishf(s + b0[r]);
  }
}

// out[b, n] = W[n,:] . f(in[b,:]) + bias[n]; one wavefront per output row n, all b.
__global__ __launch_bounds__(256) void dense_small_kernel(const float* __restrict__ in, int B, int K, int swish_in,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          int N, float* __restrict__ out) {
  extern __shared__ float sh[];  // f(in) [B][K]
  for (int i = threadIdx.x; i < B * K; i += 256) sh[i] = swish_in ? swishf(in[i]) : in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  for (int b = 0; b < B; ++b) {
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(w[static_cast<long>(n) * K + k], sh[b * K + k], s);
    s = group_sum<64>(s);
    if (lane == 0) out[static_cast<long>(b) * N + n] = s + (bias ? bias[n] : 0.f);
  }
}

// K1 for many clips (17 .. 64 per call): a LANE is a clip.  The forms above walk the clips one after the other (64-lane butterfly per
// clip and output feature: 60-90 us per launch at 64 clips for 6 MFLOP); here a wavefront owns one output feature, its 64 lanes the
// clips, the weight row is wave-uniform (scalar loads) and f(in) sits in LDS transposed [K][65]: one pass over K.
// Same products; the sum over k runs in k order in one lane instead of 64 partial sums and a butterfly: fp32 rounding differs.
template <bool EMB>
__global__ __launch_bounds__(256) void dense_lanes_kernel(const void* __restrict__ t, int t_is_f32, const float* __restrict__ freq,
                                                          const float* __restrict__ in, int B, int K, int swish_in, int swish_out,
                                                          const float* __restrict__ w, const float* __restrict__ bias, int N,
                                                          float* __restrict__ out) {
  extern __shared__ float sh[];  // f(in)^T [K][65]
  if constexpr (EMB) {           // in = the sinusoidal embedding of t (sal_unet.py:15-33), K = ch
    const int half = K / 2;
    for (int i = threadIdx.x; i < 64 * half; i += 256) {
      const int j = i >> 6, b = i & 63;
      float sv = 0.f, cv = 0.f;
      if (b < B) {
        const float tv = t_is_f32 ? static_cast<const float*>(t)[b] : static_cast<float>(static_cast<const long long*>(t)[b]);
        const float a = tv * freq[j];
        sv = sinf(a); cv = cosf(a);
      }
      sh[j * 65 + b] = sv;
      sh[(half + j) * 65 + b] = cv;
    }
    if ((K & 1) && threadIdx.x < 64) sh[(K - 1) * 65 + threadIdx.x] = 0.f;
  } else {
    for (int i = threadIdx.x; i < B * K; i += 256) {
      const int b = i / K, k = i - b * K;
      const float v = in[i];
      sh[k * 65 + b] = swish_in ? swishf(v) : v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int n = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (n >= N) return;
  const float* wr = w + static_cast<long>(n) * K;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(wr[k], sh[k * 65 + lane], s);
  s += bias ? bias[n] : 0.f;
  if (lane < B) out[static_cast<long>(lane) * N + n] = swish_out ? swishf(s) : s;
}

// ------------------------------------------------------------------------------------------------
// K2: conv_in 1 -> C, 3x3, pad 1; NCHW (C=1) in, NHWC out.  R/.../sal_unet.py:240,292
// A thread owns 4 output channels (its 36 weights stay in registers) and walks over pixels; the
// C/4 lanes of a pixel read the same 9 inputs (broadcast) and write one contiguous C-float row.
// With skip_mod = s only rows/columns with index % s != s-1 are produced (the stride-s 3x3 consumer
// never reads the others).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, T* __restrict__ out, int B,
                                                      int H, int W, int C, int skip_mod, int Hn, int Wn, int relu) {
  const int c4n = C >> 2;
  const int ppb = 256 / c4n;  // pixels per pass
  const int c4 = threadIdx.x % c4n, ps = threadIdx.x / c4n;
  if (ps >= ppb) return;
  const int c = c4 * 4;
  float wr[4][9], bi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bi[j] = bias[c + j];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[j][k] = w[(c + j) * 9 + k];
  }
  const int keep = skip_mod > 0 ? skip_mod - 1 : 1;
  const long total = static_cast<long>(B) * Hn * Wn;
  for (long i = static_cast<long>(blockIdx.x) * ppb + ps; i < total; i += static_cast<long>(gridDim.x) * ppb) {
    const int xn = static_cast<int>(i % Wn);
    const int yn = static_cast<int>((i / Wn) % Hn);
    const int b = static_cast<int>(i / (static_cast<long>(Wn) * Hn));
    const int xw = skip_mod > 0 ? (xn / keep) * skip_mod + xn % keep : xn;
    const int yh = skip_mod > 0 ? (yn / keep) * skip_mod + yn % keep : yn;
    const float* img = x + static_cast<long>(b) * H * W;
    float o[4] = {bi[0], bi[1], bi[2], bi[3]};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = yh + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = xw + kx - 1;
        const bool ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
        const float v = ok ? img[static_cast<long>(iy) * W + ix] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(v, wr[j][ky * 3 + kx], o[j]);
      }
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
    }
    st4(out + ((static_cast<long>(b) * H + yh) * W + xw) * C + c, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// K6: NCTHW -> frames-of-NHWC transpose through a 64x65 LDS tile, plus the noise map as last frame.
// vis [B][C][Q], Q = Tv*hw  ->  out [B][Tout*hw][C] (first Q rows); noise [B][hw][C] -> rows Tv*hw...
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pack_frames_body(const float* __restrict__ vis, const T* __restrict__ noise, T* __restrict__ out,
                                                 int C, int Q, int hw, int Tout, int tiles_q, int n_transpose_blocks, int bx,
                                                 int n_blocks, int b, float (*tile)[65]) {
  const long out_b = static_cast<long>(b) * Tout * hw * C;
  if (bx < n_transpose_blocks) {
    const int tq = bx % tiles_q, tcx = bx / tiles_q;
    const int q0 = tq * 64, c0 = tcx * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* src = vis + static_cast<long>(b) * C * Q;
    for (int r = ty; r < 64; r += 4) {  // r: channel within tile, tx: q within tile (contiguous reads)
      const int c = c0 + r, q = q0 + tx;
      tile[r][tx] = (c < C && q < Q) ? src[static_cast<long>(c) * Q + q] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {  // r: q within tile, tx: channel (contiguous writes)
      const int q = q0 + r, c = c0 + tx;
      if (q < Q && c < C) out[out_b + static_cast<long>(q) * C + c] = static_cast<T>(tile[tx][r]);
    }
  } else if (noise) {
    const int nb = n_blocks - n_transpose_blocks;
    const long n4 = static_cast<long>(hw) * C / 4;
    const T* src = noise + static_cast<long>(b) * hw * C;
    T* dst = out + out_b + static_cast<long>(Q) * C;
    for (long i = static_cast<long>(bx - n_transpose_blocks) * 256 + threadIdx.x; i < n4; i += static_cast<long>(nb) * 256)
      st4(dst + i * 4, ld4(src + i * 4));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_frames_kernel(const float* __restrict__ vis, const T* __restrict__ noise,
                                                          T* __restrict__ out, int C, int Q, int hw, int Tout,
                                                          int tiles_q, int tiles_c, int n_transpose_blocks) {
  __shared__ float tile[64][65];
  (void)tiles_c;
  pack_frames_body<T>(vis, noise, out, C, Q, hw, Tout, tiles_q, n_transpose_blocks, blockIdx.x, gridDim.x, blockIdx.y, tile);
}

// The frame tensors of all decoder stages in ONE launch (the three per-stage launches are 16-21 us each, mostly launch
// latency): blockIdx.x runs through the problems' block ranges.
struct PackMulti {
  const float* vis[4];
  const void* noise[4];
  void* out[4];
  int C[4], Q[4], hw[4], Tout[4], tiles_q[4], ntb[4], blocks[4];
  int n;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_frames_multi_kernel(PackMulti m) {
  __shared__ float tile[64][65];
  int bx = blockIdx.x, j = 0;
  while (j + 1 < m.n && bx >= m.blocks[j]) { bx -= m.blocks[j]; ++j; }
  pack_frames_body<T>(m.vis[j], static_cast<const T*>(m.noise[j]), static_cast<T*>(m.out[j]), m.C[j], m.Q[j], m.hw[j], m.Tout[j],
                      m.tiles_q[j], m.ntb[j], bx, m.blocks[j], blockIdx.y, tile);
}

// ------------------------------------------------------------------------------------------------
// bilinear resize (align_corners=False) on NHWC.  R/.../common_block.py:197; sal_unet.py:325-327,482-484
// ------------------------------------------------------------------------------------------------
// One bilinear blend with a fixed operation order (explicit FMAs): shared by the stand-alone single-channel resize and
// the fused step tail so that the two produce identical bits.
__device__ __forceinline__ float bilerp_w(float v00, float v01, float v10, float v11, float hx, float lx, float hy, float ly) {
  const float top = fmaf(lx, v01, hx * v00);
  const float bot = fmaf(lx, v11, hx * v10);
  return fmaf(ly, bot, hy * top);
}

template <int VEC, typename T>
__global__ __launch_bounds__(256) void resize_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int h,
                                                     int w, int H, int W, int C, float sy, float sx) {
  const int cv = C / VEC;
  const long total = static_cast<long>(N) * H * W * cv;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % cv) * VEC;
    long pix = i / cv;
    const int X = static_cast<int>(pix % W);
    pix /= W;
    const int Y = static_cast<int>(pix % H);
    const int n = static_cast<int>(pix / H);
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_coord(Y, sy, h, y0, y1, ly);
    bilin_coord(X, sx, w, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const T* b = in + static_cast<long>(n) * h * w * C + c;
    const long o = ((static_cast<long>(n) * H + Y) * W + X) * C + c;
    if constexpr (VEC == 4) {
      const float4 v00 = ld4(b + (static_cast<long>(y0) * w + x0) * C), v01 = ld4(b + (static_cast<long>(y0) * w + x1) * C);
      const float4 v10 = ld4(b + (static_cast<long>(y1) * w + x0) * C), v11 = ld4(b + (static_cast<long>(y1) * w + x1) * C);
      float4 r;
      r.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
      r.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
      r.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
      r.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
      st4(out + o, r);
    } else {
      const float v00 = static_cast<float>(b[(static_cast<long>(y0) * w + x0) * C]), v01 = static_cast<float>(b[(static_cast<long>(y0) * w + x1) * C]);
      const float v10 = static_cast<float>(b[(static_cast<long>(y1) * w + x0) * C]), v11 = static_cast<float>(b[(static_cast<long>(y1) * w + x1) * C]);
      out[o] = static_cast<T>(bilerp_w(v00, v01, v10, v11, hx, lx, hy, ly));   // == bilerp1 for C = 1
    }
  }
}

struct ResizeSumArgs {
  const void* in[4];  // T* of the launch's storage type
  int h[4], w[4];
  float sy[4], sx[4];
  int n_in;
};

// Patch form of the bilinear resize for integer up-scaling (the only case on the hot path: x2 in UpEmbed,
// x2..x16 in the 4-scale sum).  One wavefront owns a 4x4 patch of output pixels (aligned to 4), lanes stride
// the channels with float4.  A 4-aligned run of 4 outputs touches only NR = 2 (factor >= 8), 3 (factor 4) or
// 4 (factor 2) source rows/columns, so the NR x NR source window is loaded ONCE and reused for all 16 outputs:
// 2 loads per output instead of 16 -- the naive form is bound by L1/L2 request rate, not by HBM.
// All coordinates / weights are wave-uniform scalars.
__device__ __forceinline__ float uniform_f(float v) {   // wave-uniform value -> SGPR
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// nr = source rows / columns a 4-aligned run of 4 outputs touches: 2 (factor >= 8), 3 (factor 4) or 4 (factor 2);
// wave-uniform, so ONE code path serves every scale (three specialised copies tripled the live accumulators).
template <typename T>
__device__ __forceinline__ void patch_accumulate(const T* __restrict__ in, int nr, int n, int h, int w, int C, int Y0,
                                                 int X0, float sy, float sx, int c, float4 (&acc)[4][4]) {
  // horizontal weights of the 4 output columns over the source columns, and (y0, y1, ly) of the 4 output rows: all
  // wave-uniform, kept in SGPRs; the vertical weights of a source row are rebuilt from (y0, y1, ly) when it is used
  float wx[4][4], lyd[4];
  int y0d[4], y1d[4];
  int rx0 = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_coord(Y0 + d, sy, h, y0, y1, ly);
    bilin_coord(X0 + d, sx, w, x0, x1, lx);
    if (d == 0) rx0 = __builtin_amdgcn_readfirstlane(x0);
    y0d[d] = __builtin_amdgcn_readfirstlane(y0);
    y1d[d] = __builtin_amdgcn_readfirstlane(y1);
    lyd[d] = uniform_f(ly);
#pragma unroll
    for (int k = 0; k < 4; ++k) wx[d][k] = uniform_f((rx0 + k == x0 ? 1.f - lx : 0.f) + (rx0 + k == x1 ? lx : 0.f));
  }
  const int ry0 = y0d[0];
  // one source row at a time: nr loads -> 4 horizontally blended values -> spread over the 4 output rows.  Keeps the
  // live set at the 16 accumulators + one row (the whole-window form needed 500 VGPRs: one wave per SIMD).
  const T* base = in + static_cast<long>(n) * h * w * C + c;
#pragma unroll 1
  for (int i = 0; i < nr; ++i) {
    const T* rp = base + static_cast<long>(min(ry0 + i, h - 1)) * w * C;
    float4 row[4];
    row[0] = ld4(rp + static_cast<long>(min(rx0, w - 1)) * C);
    row[1] = ld4(rp + static_cast<long>(min(rx0 + 1, w - 1)) * C);
    row[2] = nr > 2 ? ld4(rp + static_cast<long>(min(rx0 + 2, w - 1)) * C) : make_float4(0, 0, 0, 0);
    row[3] = nr > 3 ? ld4(rp + static_cast<long>(min(rx0 + 3, w - 1)) * C) : make_float4(0, 0, 0, 0);
    float wy[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) wy[d] = (ry0 + i == y0d[d] ? 1.f - lyd[d] : 0.f) + (ry0 + i == y1d[d] ? lyd[d] : 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float4 t = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t.x = fmaf(wx[e][k], row[k].x, t.x); t.y = fmaf(wx[e][k], row[k].y, t.y);
        t.z = fmaf(wx[e][k], row[k].z, t.z); t.w = fmaf(wx[e][k], row[k].w, t.w);
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        acc[d][e].x = fmaf(wy[d], t.x, acc[d][e].x); acc[d][e].y = fmaf(wy[d], t.y, acc[d][e].y);
        acc[d][e].z = fmaf(wy[d], t.z, acc[d][e].z); acc[d][e].w = fmaf(wy[d], t.w, acc[d][e].w);
      }
    }
  }
}

// out = in0^ + in1^ + in2^ + in3^, x^ = bilinear resize of x to (H, W): one write of the big map.
// Requires H, W multiples of 4 and every input an integer factor (2, 4, 8, ...) smaller; n_in = 1 is the
// plain resize.  One wavefront per (4x4 output patch, 256-channel slab).  R/.../sal_unet.py:482-487, common_block.py:197.
template <typename T>
__global__ __launch_bounds__(256) void resize_sum_kernel(ResizeSumArgs a, T* __restrict__ out, int H, int W, int C,
                                                         int w_patches, long n_patches) {
  const int lane = threadIdx.x & 63;
  const int c_slabs = (C + 255) >> 8;
  const long item = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (item >= n_patches * c_slabs) return;
  const long patch = item / c_slabs;
  const int c = static_cast<int>(item - patch * c_slabs) * 256 + lane * 4;
  const int px = static_cast<int>(patch % w_patches);
  const long t = patch / w_patches;
  const int py = static_cast<int>(t % (H >> 2));
  const int n = static_cast<int>(t / (H >> 2));
  const int Y0 = __builtin_amdgcn_readfirstlane(py * 4), X0 = __builtin_amdgcn_readfirstlane(px * 4);
  if (c >= C) return;
  float4 acc[4][4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[d][e] = make_float4(0, 0, 0, 0);
#pragma unroll 1
  for (int s = 0; s < a.n_in; ++s) {
    const int hs = a.h[s], ws = a.w[s];
    const int f = H / hs;  // wave-uniform
    const T* src = static_cast<const T*>(a.in[s]);
    patch_accumulate<T>(src, f >= 8 ? 2 : (f == 4 ? 3 : 4), n, hs, ws, C, Y0, X0, a.sy[s], a.sx[s], c, acc);
  }
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      st4(out + ((static_cast<long>(n) * H + Y0 + d) * W + X0 + e) * C + c, acc[d][e]);
}

// General fallback (any sizes): one wavefront per output pixel.
template <typename T>
__global__ __launch_bounds__(256) void resize_sum_generic_kernel(ResizeSumArgs a, T* __restrict__ out, int H, int W,
                                                                 int C, int w_tiles) {
  int bid = blockIdx.x;
  const int xt = bid % w_tiles; bid /= w_tiles;
  const int Y = bid % H;
  const int n = bid / H;
  const int lane = threadIdx.x & 63;
  const int X = __builtin_amdgcn_readfirstlane(xt * 4 + (threadIdx.x >> 6));
  if (X >= W) return;
  for (int c = lane * 4; c < C; c += 256) {
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < a.n_in) {
        const int h = a.h[s], w = a.w[s];
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_coord(Y, a.sy[s], h, y0, y1, ly);
        bilin_coord(X, a.sx[s], w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const T* b = static_cast<const T*>(a.in[s]) + static_cast<long>(n) * h * w * C + c;
        const float4 v00 = ld4(b + (static_cast<long>(y0) * w + x0) * C), v01 = ld4(b + (static_cast<long>(y0) * w + x1) * C);
        const float4 v10 = ld4(b + (static_cast<long>(y1) * w + x0) * C), v11 = ld4(b + (static_cast<long>(y1) * w + x1) * C);
        acc.x += hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        acc.y += hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        acc.z += hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        acc.w += hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
      }
    }
    st4(out + ((static_cast<long>(n) * H + Y) * W + X) * C + c, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// K7: audio fusion.  R/.../transformer.py:133-146
//   a[b,t,c,y,x] = a_small[(b t), (y/up, x/up), c]          (nearest upsample, Q3)
//   m[b,c,y,x]   = mean_t a * x[b,t,y,x,c];  s = softmax_x(m)     (softmax over W only, Q4)
//   out[b,c,t,y,x] = a * s        written in the reference's NCTHW order (reinterpreted by the caller, Q5)
// One workgroup per (b, y, 32-channel slab): lanes 0..31 of each wave = channels (coalesced 128-B
// reads of x), waves/iterations = x positions; the W-long rows are transposed through LDS so the
// NCTHW stores are W-contiguous.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void audio_fuse_kernel(const TT* __restrict__ a_small, const TT* __restrict__ x,
                                                         TT* __restrict__ out, int T, int H, int W, int C, int h, int w,
                                                         int up, int a_ld) {
  extern __shared__ float sh[];  // s[32][W+1]
  const int WP = W + 1;
  const int cslabs = C / 32;
  int bid = blockIdx.x;
  const int cs = bid % cslabs; bid /= cslabs;
  const int y = bid % H;
  const int b = bid / H;
  const int ys = y / up;
  // pass 1: m[c][x] = mean_t a * x.  A lane owns four channels of one x position (16-byte loads: 8 lanes cover the slab's
  // 128 contiguous bytes of a pixel, 32 positions per sweep); per (c, x) the products are added in frame order.
  {
    const int cq = (threadIdx.x & 7) * 4, xl = threadIdx.x >> 3;   // channel quad within the slab, x lane 0..31
    const int c = cs * 32 + cq;
    for (int xx = xl; xx < W; xx += 32) {
      const int xs = xx / up;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = 0; t < T; ++t) {
        const float4 av = ld4(a_small + ((static_cast<long>(b) * T + t) * h * w + ys * w + xs) * a_ld + c);
        const float4 xv = ld4(x + (((static_cast<long>(b) * T + t) * H + y) * W + xx) * C + c);
        s.x = fmaf(av.x, xv.x, s.x); s.y = fmaf(av.y, xv.y, s.y); s.z = fmaf(av.z, xv.z, s.z); s.w = fmaf(av.w, xv.w, s.w);
      }
      const float ft = static_cast<float>(T);
      sh[(cq + 0) * WP + xx] = s.x / ft; sh[(cq + 1) * WP + xx] = s.y / ft;
      sh[(cq + 2) * WP + xx] = s.z / ft; sh[(cq + 3) * WP + xx] = s.w / ft;
    }
  }
  __syncthreads();
  // pass 2: softmax over x for each of the 32 channel rows: 8 lanes per row
  {
    const int row = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    float mx = -3.0e38f;
    for (int xx = l8; xx < W; xx += 8) mx = fmaxf(mx, sh[row * WP + xx]);
    mx = group_max<8>(mx);
    float sum = 0.f;
    for (int xx = l8; xx < W; xx += 8) {
      const float e = expf(sh[row * WP + xx] - mx);
      sh[row * WP + xx] = e;
      sum += e;
    }
    sum = group_sum<8>(sum);
    const float inv = 1.0f / sum;
    for (int xx = l8; xx < W; xx += 8) sh[row * WP + xx] *= inv;
  }
  __syncthreads();
  // pass 3: out[b, c, t, y, :] = a * s, x fastest across lanes
  // flat over (channel, frame, x) with x fastest across lanes; the index split uses reciprocal multiplies (i < 2^23, exact
  // after the integer fix-up) instead of three integer divisions per 4-byte store
  if ((W & 3) == 0) {
    // four consecutive x per lane: one 16-byte (fp32) / 8-byte (16-bit) store instead of four 4 / 2-byte ones -- the pass writes the
    // whole [B,C,T,H,W] tensor (74 MB at stage 3) and was bound by store instructions, not bytes
    const int W4 = W >> 2, total4 = 32 * T * W4;
    const float inv_w4 = 1.0f / static_cast<float>(W4), inv_t4 = 1.0f / static_cast<float>(T);
    for (int i = threadIdx.x; i < total4; i += 256) {
      int iw = static_cast<int>((static_cast<float>(i) + 0.5f) * inv_w4);
      int x4 = i - iw * W4;
      if (x4 < 0) { --iw; x4 += W4; } else if (x4 >= W4) { ++iw; x4 -= W4; }
      int cc = static_cast<int>((static_cast<float>(iw) + 0.5f) * inv_t4);
      int t = iw - cc * T;
      if (t < 0) { --cc; t += T; } else if (t >= T) { ++cc; t -= T; }
      const int cg = cs * 32 + cc, xx = x4 * 4;
      const TT* arow = a_small + ((static_cast<long>(b) * T + t) * h * w + ys * w) * a_ld + cg;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = static_cast<float>(arow[static_cast<long>((xx + e) / up) * a_ld]) * sh[cc * WP + xx + e];
      st4(out + (((static_cast<long>(b) * C + cg) * T + t) * H + y) * W + xx, make_float4(v[0], v[1], v[2], v[3]));
    }
    return;
  }
  const int total = 32 * T * W;
  const float inv_w = 1.0f / static_cast<float>(W), inv_t = 1.0f / static_cast<float>(T);
  for (int i = threadIdx.x; i < total; i += 256) {
    int iw = static_cast<int>((static_cast<float>(i) + 0.5f) * inv_w);
    int xx = i - iw * W;
    if (xx < 0) { --iw; xx += W; } else if (xx >= W) { ++iw; xx -= W; }
    int cc = static_cast<int>((static_cast<float>(iw) + 0.5f) * inv_t);
    int t = iw - cc * T;
    if (t < 0) { --cc; t += T; } else if (t >= T) { ++cc; t -= T; }
    const int cg = cs * 32 + cc;
    const float av = static_cast<float>(a_small[((static_cast<long>(b) * T + t) * h * w + ys * w + xx / up) * a_ld + cg]);
    out[(((static_cast<long>(b) * C + cg) * T + t) * H + y) * W + xx] = static_cast<TT>(av * sh[cc * WP + xx]);
  }
}

// K7 on 16-bit storage.  The form above moves 8 bytes per lane and a 32-channel slab is 64 bytes of a pixel -- half a 128-byte
// line per request, the other half fetched again by the neighbouring slab's workgroup (1.4-1.7 TB/s at 64 clips).  Here:
//  * a workgroup owns (clip, R rows, CS channels), CS * 2 bytes a multiple of 128 (CS = 64) or the whole pixel (C = 96), and every
//    global access is 16 bytes per lane (8 channels of x / a_small, 8 consecutive x of the [B,C,T,H,W] result);
//  * pass 3 (out = a * s, written W-contiguous) needs a_small channel-major: frame t's slab [rows of the audio map][w][CS] is staged
//    in LDS (two buffers, one barrier per frame, the next frame's piece in registers meanwhile) instead of 2-byte strided loads;
//  * R rows per workgroup so that a (c, t) run of the result is R * W * 2 >= 192 bytes where the map is narrow.
// Arithmetic and summation order are those of audio_fuse_kernel: results are bit-identical.
// n / d for 0 <= n, n d < 2^20, with magic = 2^20 / d + 1 (host): one multiply and one shift instead of a ~25-instruction division
__device__ __forceinline__ int div_magic(int n, unsigned magic) { return static_cast<int>((static_cast<unsigned>(n) * magic) >> 20); }
inline unsigned make_magic(int d) { return (1u << 20) / static_cast<unsigned>(d) + 1u; }

struct AudioFuse16Args {
  int T, H, W, C, h, w, a_ld, R, RS;
  unsigned mW, mR, mXG;           // div_magic constants of W, R, W / XV
};

// CS: channels per workgroup, XV: consecutive x per store, UPS: log2 of the up-sampling factor of the audio map
template <typename TT, int CS, int XV, int UPS>
__global__ __launch_bounds__(256) void audio_fuse16_kernel(const TT* __restrict__ a_small, const TT* __restrict__ x,
                                                           TT* __restrict__ out, AudioFuse16Args g) {
  extern __shared__ float sh[];          // s[CS][R W + 1] | a_t[2][RS w][CS + 2] (16-bit)
  constexpr int OCT = CS / 8, AP = CS + 2;          // AP: pixel pitch of the staged audio slab, an odd number of dwords
  constexpr int KMAX = 6;                           // pass-3 items per thread (host-checked)
  const int T = g.T, H = g.H, W = g.W, C = g.C, h = g.h, w = g.w, a_ld = g.a_ld, R = g.R, RS = g.RS;
  const int RW = R * W, RP = RW + 1;
  TT* a_sh = reinterpret_cast<TT*>(sh + CS * RP);
  const int cslabs = C / CS, rblocks = H / R;
  int bid = blockIdx.x;
  const int cs = bid % cslabs; bid /= cslabs;
  const int yb = bid % rblocks;
  const int b = bid / rblocks;
  const int y0 = yb * R, c0 = cs * CS;
  const int ys0 = y0 >> UPS;
  // ---- pass 1: m[c][r, x] = mean_t a * x; an item = (pixel of the R x W block, channel octet), octets fastest across lanes
  for (int i = threadIdx.x; i < RW * OCT; i += 256) {
    const int p = i / OCT, o = i - p * OCT;
    const int r = div_magic(p, g.mW), xx = p - r * W;
    const int ys = (y0 + r) >> UPS, xs = xx >> UPS;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const TT* ap = a_small + (static_cast<long>(b) * T * h * w + ys * w + xs) * a_ld + c0 + o * 8;
    const TT* xp = x + ((static_cast<long>(b) * T * H + (y0 + r)) * W + xx) * C + c0 + o * 8;
    const long a_fs = static_cast<long>(h) * w * a_ld, x_fs = static_cast<long>(H) * W * C;
#pragma unroll 3
    for (int t = 0; t < T; ++t) {
      const f8v av = ld8(ap + t * a_fs);
      const f8v xv = ld8(xp + t * x_fs);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = fmaf(av.v[e], xv.v[e], s[e]);
    }
    const float ft = static_cast<float>(T);
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[(o * 8 + e) * RP + p] = s[e] / ft;
  }
  __syncthreads();
  // ---- pass 2: softmax over x for each (channel, row): 8 lanes per softmax row
  for (int row = threadIdx.x >> 3; row < CS * R; row += 32) {
    const int c = div_magic(row, g.mR), r = row - c * R, l8 = threadIdx.x & 7;
    float* sr = sh + c * RP + r * W;
    float mx = -3.0e38f;
    for (int xx = l8; xx < W; xx += 8) mx = fmaxf(mx, sr[xx]);
    mx = group_max<8>(mx);
    float sum = 0.f;
    for (int xx = l8; xx < W; xx += 8) {
      const float e = expf(sr[xx] - mx);
      sr[xx] = e;
      sum += e;
    }
    sum = group_sum<8>(sum);
    const float inv = 1.0f / sum;
    for (int xx = l8; xx < W; xx += 8) sr[xx] *= inv;
  }
  // ---- pass 3: out[b, c, t, y0 + r, x] = a * s, one frame at a time.  A thread keeps its items (channel, row, XV consecutive x)
  // for every frame: their softmax weights stay in registers, per frame an item costs its audio reads, XV products and one store
  const int apix = RS * w;                     // audio pixels under the block (rows ys0 .. ys0 + RS - 1, clamped)
  const int n_chunk = apix * OCT;              // 16-byte pieces of a frame's slab
  uint4 stage[3];                              // at most 768 pieces per frame (host-checked)
  auto fetch = [&](int t) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = threadIdx.x + k * 256;
      if (i < n_chunk) {
        const int px = i / OCT, o = i - px * OCT;
        const int ry = px / w, rx = px - ry * w;
        const int ys = min(ys0 + ry, h - 1);
        stage[k] = *reinterpret_cast<const uint4*>(a_small + ((static_cast<long>(b) * T + t) * h * w + ys * w + rx) * a_ld + c0 + o * 8);
      }
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = threadIdx.x + k * 256;
      if (i < n_chunk) {
        const int px = i / OCT, o = i - px * OCT;
        unsigned* d = reinterpret_cast<unsigned*>(a_sh + (buf * apix + px) * AP + o * 8);
        d[0] = stage[k].x; d[1] = stage[k].y; d[2] = stage[k].z; d[3] = stage[k].w;
      }
    }
  };
  fetch(0);
  park(0);
  __syncthreads();                             // also orders pass 2's writes before the reads below
  const int xg = W / XV, items = CS * R * xg;
  float sw[KMAX][XV];
  int a_of[KMAX];
  long o_of[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int i = threadIdx.x + k * 256;
    a_of[k] = -1;
    o_of[k] = 0;
    if (i < items) {
      const int cr = div_magic(i, g.mXG), gq = i - cr * xg;
      const int c = div_magic(cr, g.mR), r = cr - c * R;
      const int xx = gq * XV;
#pragma unroll
      for (int e = 0; e < XV; ++e) sw[k][e] = sh[c * RP + r * W + xx + e];
      a_of[k] = ((((y0 + r) >> UPS) - ys0) * w + (xx >> UPS)) * AP + c;
      o_of[k] = ((static_cast<long>(b) * C + c0 + c) * T * H + y0 + r) * W + xx;
    }
  }
  const long o_fs = static_cast<long>(H) * W;
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) fetch(t + 1);
    const TT* at = a_sh + (t & 1) * apix * AP;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (a_of[k] < 0) continue;
      const TT* ar = at + a_of[k];
      TT* op = out + o_of[k] + t * o_fs;
      if constexpr (XV == 8) {
        f8v v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = to_f32(ar[(e >> UPS) * AP]) * sw[k][e];
        st8(op, v);
      } else {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = to_f32(ar[(e >> UPS) * AP]) * sw[k][e];
        st4(op, make_float4(v[0], v[1], v[2], v[3]));
      }
    }
    if (t + 1 < T) {
      park((t + 1) & 1);
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K11: attention core with a short pooled K/V (Lk <= 32).  R/.../attention.py:97-108
// G lanes share one (query, head): lane g owns the float4 pieces g, g+G, ... of the head's d channels,
// so the G lanes read/write 16*G contiguous bytes of the q / o rows, the Lk partial scores are
// combined with log2(G) xor-shuffles, and the softmax runs redundantly in registers (no LDS, no
// cross-wave traffic).  K and V of the image sit in LDS; lanes with the same g read the same
// address (broadcast), different g read neighbouring 16-byte slots (conflict-free).
// FLOPs are 0.4 % of a step: the kernel is bound by the q / o traffic.
// ------------------------------------------------------------------------------------------------
template <int LK, int G, typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ v, T* __restrict__ o, int Lq,
                                                        int Lk, int C, int heads, float scale) {
  // One workgroup = one (image, head, chunk of queries): only this head's d = C / heads columns of K and V are staged
  // (the all-heads form re-staged 2 x 18 x C floats per 8 queries on the coarse stages: 44 MB of LDS fills per launch).
  extern __shared__ float sh[];  // K[Lk][d] | V[Lk][d]
  const int d = C / heads, nf4 = d >> 2;
  float* Ks = sh;
  float* Vs = sh + Lk * d;
  const int n = blockIdx.y / heads, hd = blockIdx.y - n * heads;
  const int cb = hd * d;
  {  // 8 independent 16-byte loads of K and of V in flight per thread
    constexpr int U = 8;
    const int n4 = Lk * nf4;
    const T* kb = k + static_cast<long>(n) * Lk * C + cb;
    const T* vb = v + static_cast<long>(n) * Lk * C + cb;
    for (int base = threadIdx.x; base < n4; base += 256 * U) {
      float4 kr[U], vr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 256;
        if (i < n4) {
          const int t = i / nf4, c4 = (i - t * nf4) * 4;
          kr[u] = ld4(kb + static_cast<long>(t) * C + c4);
          vr[u] = ld4(vb + static_cast<long>(t) * C + c4);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 256;
        if (i < n4) { st4(Ks + i * 4, kr[u]); st4(Vs + i * 4, vr[u]); }
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int QPW = 64 / G;      // queries per wavefront
  const int g = lane % G;
  const int l = blockIdx.x * (4 * QPW) + wave * QPW + lane / G;
  const bool valid = l < Lq;
  const int lc = valid ? l : Lq - 1;  // out-of-range lanes stay alive for the shuffles
  const T* qr = q + (static_cast<long>(n) * Lq + lc) * C + cb;
  T* orow = o + (static_cast<long>(n) * Lq + lc) * C + cb;
  float sc[LK];
#pragma unroll
  for (int t = 0; t < LK; ++t) sc[t] = 0.f;
  for (int i = g; i < nf4; i += G) {
    const float4 qv = ld4(qr + 4 * i);
#pragma unroll
    for (int t = 0; t < LK; ++t) {
      if (t < Lk) {
        const float4 kv = ld4(Ks + t * d + 4 * i);
        sc[t] = fmaf(qv.x, kv.x, fmaf(qv.y, kv.y, fmaf(qv.z, kv.z, fmaf(qv.w, kv.w, sc[t]))));
      }
    }
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < LK; ++t) {
    if (t < Lk) { sc[t] = group_sum<G>(sc[t]) * scale; mx = fmaxf(mx, sc[t]); }
  }
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < LK; ++t) {
    if (t < Lk) { sc[t] = expf(sc[t] - mx); sum += sc[t]; }
  }
  const float inv = 1.0f / sum;
  for (int i = g; i < nf4; i += G) {
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < LK; ++t) {
      if (t < Lk) {
        const float4 vv = ld4(Vs + t * d + 4 * i);
        const float pw = sc[t] * inv;
        acc.x = fmaf(pw, vv.x, acc.x); acc.y = fmaf(pw, vv.y, acc.y);
        acc.z = fmaf(pw, vv.z, acc.z); acc.w = fmaf(pw, vv.w, acc.w);
      }
    }
    if (valid) st4(orow + 4 * i, acc);
  }
}

// K11 on 16-bit storage, 16-byte form.  The kernel above gives a (query, head) G lanes of float4 pieces -- 8-byte accesses -- and a
// workgroup 4 * 64 / G queries: at the coarse stages (head dim 384 / 192, G = 16) that is 16 queries per 55 KB of staged K / V
// (0.8-1.4 TB/s at 64 clips).  Here a workgroup owns ALL queries of one (image, head): K and V of the head go to LDS once, as fp32 in
// lane order (conflict-free 16-byte reads); a query's head slice is three octets per lane on G = d / 24 lanes (16-byte loads and
// stores); a lane group carries TWO queries per pass so that every K / V read from LDS feeds two dot products.
// Same formulas as attention_kernel (scores * scale, max, exp, normalise, P V), another summation order inside the dot products.
template <typename T, int G, int LK>
__global__ __launch_bounds__(256, 2) void attention16_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                             T* __restrict__ o, int Lq, int Lk, int C, int heads, float scale, int q_chunk) {
  extern __shared__ float sh[];  // Kl[Lk][d] | Vl[Lk][d], a row in lane order: [(i * 2 + hf) * G + gl][4] = channels (gl + i G) 8 + hf 4 ..
  constexpr int QPP = 256 / G;     // lane groups per workgroup
  const int d = C / heads;
  float* Kl = sh;
  float* Vl = sh + Lk * d;
  const int n = blockIdx.x / heads, hd = blockIdx.x - n * heads;
  const int cb = hd * d;
  for (int it = threadIdx.x; it < Lk * (d >> 3); it += 256) {
    const int t = it / (d >> 3), oc = it - t * (d >> 3);
    const int i = oc / G, gl = oc - i * G;
    const f8v kv = ld8(k + (static_cast<long>(n) * Lk + t) * C + cb + oc * 8);
    const f8v vv = ld8(v + (static_cast<long>(n) * Lk + t) * C + cb + oc * 8);
    float* kd = Kl + t * d + ((i * 2) * G + gl) * 4;
    float* vd = Vl + t * d + ((i * 2) * G + gl) * 4;
    st4(kd, make_float4(kv.v[0], kv.v[1], kv.v[2], kv.v[3])); st4(kd + G * 4, make_float4(kv.v[4], kv.v[5], kv.v[6], kv.v[7]));
    st4(vd, make_float4(vv.v[0], vv.v[1], vv.v[2], vv.v[3])); st4(vd + G * 4, make_float4(vv.v[4], vv.v[5], vv.v[6], vv.v[7]));
  }
  __syncthreads();
  const int gl = threadIdx.x % G;
  // blockIdx.y: a chunk of q_chunk queries (few images: the (image, head) pairs alone would leave most CUs idle)
  const int q_lo = blockIdx.y * q_chunk, q_hi = min(Lq, q_lo + q_chunk);
  const int gr = q_lo + threadIdx.x / G;
  // the queries of the NEXT pass are requested (raw 16-byte pieces) before this pass is computed: a pass is a dependent chain of
  // LDS reads and FMAs behind its loads, and a workgroup makes only a handful of passes
  uint4 nraw[6];
  auto fetch = [&](int l0) __attribute__((always_inline)) {
    const int l1 = l0 + QPP < q_hi ? l0 + QPP : l0;
    const T* q0 = q + (static_cast<long>(n) * Lq + l0) * C + cb;
    const T* q1 = q + (static_cast<long>(n) * Lq + l1) * C + cb;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      nraw[i] = *reinterpret_cast<const uint4*>(q0 + (gl + i * G) * 8);
      nraw[3 + i] = *reinterpret_cast<const uint4*>(q1 + (gl + i * G) * 8);
    }
  };
  if (gr < q_hi) fetch(gr);
  for (int l0 = gr; l0 < q_hi; l0 += 2 * QPP) {          // trip count uniform within a lane group
    const int l1 = l0 + QPP;
    const bool two = l1 < q_hi;
    float qa[24], qb[24];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f8v a = ld8(reinterpret_cast<const T*>(&nraw[i])), b = ld8(reinterpret_cast<const T*>(&nraw[3 + i]));
#pragma unroll
      for (int e = 0; e < 8; ++e) { qa[i * 8 + e] = a.v[e]; qb[i * 8 + e] = b.v[e]; }
    }
    if (l0 + 2 * QPP < q_hi) fetch(l0 + 2 * QPP);
    float sa[LK], sb[LK];
#pragma unroll
    for (int t = 0; t < LK; ++t) {
      sa[t] = 0.f; sb[t] = 0.f;
      if (t < Lk) {
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          const float4 kk = ld4(Kl + t * d + (p * G + gl) * 4);
          sa[t] = fmaf(qa[p * 4 + 0], kk.x, fmaf(qa[p * 4 + 1], kk.y, fmaf(qa[p * 4 + 2], kk.z, fmaf(qa[p * 4 + 3], kk.w, sa[t]))));
          sb[t] = fmaf(qb[p * 4 + 0], kk.x, fmaf(qb[p * 4 + 1], kk.y, fmaf(qb[p * 4 + 2], kk.z, fmaf(qb[p * 4 + 3], kk.w, sb[t]))));
        }
      }
    }
    float ma = -3.0e38f, mb = -3.0e38f;
#pragma unroll
    for (int t = 0; t < LK; ++t) {
      if (t < Lk) {
        sa[t] = group_sum<G>(sa[t]) * scale; ma = fmaxf(ma, sa[t]);
        sb[t] = group_sum<G>(sb[t]) * scale; mb = fmaxf(mb, sb[t]);
      }
    }
    float suma = 0.f, sumb = 0.f;
#pragma unroll
    for (int t = 0; t < LK; ++t) {
      if (t < Lk) { sa[t] = expf(sa[t] - ma); suma += sa[t]; sb[t] = expf(sb[t] - mb); sumb += sb[t]; }
    }
    const float ia = 1.0f / suma, ib = 1.0f / sumb;
    float oa[24], ob[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) { oa[j] = 0.f; ob[j] = 0.f; }
#pragma unroll
    for (int t = 0; t < LK; ++t) {
      if (t < Lk) {
        const float pa = sa[t] * ia, pb = sb[t] * ib;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          const float4 vv = ld4(Vl + t * d + (p * G + gl) * 4);
          oa[p * 4 + 0] = fmaf(pa, vv.x, oa[p * 4 + 0]); oa[p * 4 + 1] = fmaf(pa, vv.y, oa[p * 4 + 1]);
          oa[p * 4 + 2] = fmaf(pa, vv.z, oa[p * 4 + 2]); oa[p * 4 + 3] = fmaf(pa, vv.w, oa[p * 4 + 3]);
          ob[p * 4 + 0] = fmaf(pb, vv.x, ob[p * 4 + 0]); ob[p * 4 + 1] = fmaf(pb, vv.y, ob[p * 4 + 1]);
          ob[p * 4 + 2] = fmaf(pb, vv.z, ob[p * 4 + 2]); ob[p * 4 + 3] = fmaf(pb, vv.w, ob[p * 4 + 3]);
        }
      }
    }
    T* o0 = o + (static_cast<long>(n) * Lq + l0) * C + cb;
    T* o1 = o + (static_cast<long>(n) * Lq + l1) * C + cb;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f8v a, b;
#pragma unroll
      for (int e = 0; e < 8; ++e) { a.v[e] = oa[i * 8 + e]; b.v[e] = ob[i * 8 + e]; }
      st8(o0 + (gl + i * G) * 8, a);
      if (two) st8(o1 + (gl + i * G) * 8, b);
    }
  }
}

// K14 tail: per-pixel dot with w[C] + sigmoid; G lanes per pixel.
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, long NHW,
                                                   int C) {
  const int gl = threadIdx.x & 7, gr = threadIdx.x >> 3;  // 8 lanes per pixel, 32 pixels per block
  for (long p = static_cast<long>(blockIdx.x) * 32 + gr; p < NHW; p += static_cast<long>(gridDim.x) * 32) {
    float s = 0.f;
    for (int c = gl * 4; c < C; c += 32) {
      const float4 a = ld4(x + p * C + c), ww = ld4(w + c);
      s += (a.x * ww.x + a.y * ww.y) + (a.z * ww.z + a.w * ww.w);
    }
    s = group_sum<8>(s);
    if (gl == 0) out[p] = sigmoidf_(s + bias[0]);
  }
}

// a x + b y + c z with ONE rounding sequence shared by the stand-alone sampler update and the fused step epilogue below,
// so that the two paths agree bit for bit.
__device__ __forceinline__ float lincomb3(float a, float x, float b, float y, bool has_y, float c, float z, bool has_z) {
  float r = a * x;
  if (has_y) r = fmaf(b, y, r);
  if (has_z) r = fmaf(c, z, r);
  return r;
}

__global__ __launch_bounds__(256) void axpbypcz_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ z, float a, float b, float c,
                                                       float* __restrict__ out, size_t n) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256)
    out[i] = lincomb3(a, x[i], b, y ? y[i] : 0.f, y != nullptr, c, z ? z[i] : 0.f, z != nullptr);
}

// One bilinear sample of a single-channel map (align_corners = False), the arithmetic of resize_kernel<1>.
__device__ __forceinline__ float bilerp1(const float* __restrict__ img, int h, int w, int Y, int X, float sy, float sx) {
  int y0, y1, x0, x1;
  float ly, lx;
  bilin_coord(Y, sy, h, y0, y1, ly);
  bilin_coord(X, sx, w, x0, x1, lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v00 = img[static_cast<long>(y0) * w + x0], v01 = img[static_cast<long>(y0) * w + x1];
  const float v10 = img[static_cast<long>(y1) * w + x0], v11 = img[static_cast<long>(y1) * w + x1];
  return bilerp_w(v00, v01, v10, v11, hx, lx, hy, ly);
}

// ------------------------------------------------------------------------------------------------
// Fused tail of a denoising step: the final x2 bilinear resize of the sigmoid map (sal_unet.py:325-327),
// the x0 -> model-output conversion of the solver's wrapper (sampler.py:286-292: noise = (x - alpha x0) / sigma) and the
// multistep update x_next = A x + c0 m + c1 m_prev (sampler.py:548-593, 797-853) in ONE pass over the 1.4 MB state:
//   x0 = resize(s);  m = ex * x + e0 * x0;  x_next = A * x + c0 * m + c1 * m_prev
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_update_kernel(const float* __restrict__ s_low, const float* __restrict__ x,
                                                            const float* __restrict__ m_prev, float* __restrict__ x0_out,
                                                            float* __restrict__ m_out, float* __restrict__ x_next, int h,
                                                            int w, int H, int W, float sy, float sx, float ex, float e0,
                                                            float A, float c0, float c1, long total) {
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int X = static_cast<int>(i % W);
    const long r = i / W;
    const int Y = static_cast<int>(r % H);
    const long n = r / H;
    const float x0 = bilerp1(s_low + n * h * w, h, w, Y, X, sy, sx);
    const float xv = x[i];
    const float m = ex == 0.f ? e0 * x0 : lincomb3(ex, xv, e0, x0, true, 0.f, 0.f, false);
    if (x0_out) x0_out[i] = x0;
    m_out[i] = m;
    if (x_next) x_next[i] = lincomb3(A, xv, c0, m, true, c1, m_prev ? m_prev[i] : 0.f, m_prev != nullptr);
  }
}

// storage-type conversion (weights once per parameter version; fp32 <-> bf16 / fp16), round to nearest even
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long n) {
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long>(gridDim.x) * 256)
    dst[i] = static_cast<D>(static_cast<float>(src[i]));
}

// MaxPool2d(k, stride s, no padding, floor) on NHWC.  R/models/vggish.py:76-84 (VGG feature stack).
template <typename T>
__global__ __launch_bounds__(256) void maxpool2d_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C,
                                                        int Ho, int Wo, int k, int s, long total4) {
  const int c4n = C >> 2;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total4; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    long pix = i / c4n;
    const int xo = static_cast<int>(pix % Wo); pix /= Wo;
    const int yo = static_cast<int>(pix % Ho);
    const long n = pix / Ho;
    float4 m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const float4 v = ld4(in + ((n * H + yo * s + dy) * W + xo * s + dx) * C + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    st4(out + ((n * Ho + yo) * Wo + xo) * C + c, m);
  }
}

static int ew_grid(long total_threads) {
  long g = (total_threads + 255) / 256;
  return static_cast<int>(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace diffsal

using namespace diffsal;

extern "C" size_t diffsal_workspace_bytes(int op, const diffsal_conv_desc* d, const long* dims, int n_dims) {
  auto D = [&](int i) { return static_cast<int>(dims[i]); };
  const bool have = dims != nullptr;
  switch (op) {
    case DIFFSAL_WS_GROUPNORM: return have && n_dims == 2 ? diffsal_groupnorm_ws_bytes(D(0), D(1)) : 0;
    case DIFFSAL_WS_CONV_IGEMM: return d ? diffsal_conv_igemm_ws_bytes(d) : 0;
    case DIFFSAL_WS_CONV_WINO: return d ? diffsal_conv_wino_ws_bytes(d) : 0;
    case DIFFSAL_WS_CONV_WINO4: return d ? diffsal_conv_wino4_ws_bytes(d) : 0;
    case DIFFSAL_WS_CONV_WINO4_STATS: return d && have && n_dims == 1 ? diffsal_conv_wino4_stats_bytes(d, D(0)) : 0;
    case DIFFSAL_WS_CONV_WGRAD: return d ? diffsal_conv_wgrad_ws_bytes(d) : 0;
    case DIFFSAL_WS_WGRAD_SEGMENTED: return have && n_dims == 4 ? diffsal_wgrad_segmented_ws_bytes(D(0), D(1), D(2), D(3)) : 0;
    case DIFFSAL_WS_TAPSUM_BWD: return have && n_dims == 4 ? static_cast<size_t>(diffsal_tapsum_bwd_ws_bytes(D(0), D(1), D(2), D(3))) : 0;
    case DIFFSAL_WS_SALIENCY_METRICS: return have && n_dims == 1 ? diffsal_saliency_metrics_ws_bytes(D(0)) : 0;
    case DIFFSAL_WS_ATTENTION_TAIL:
      return have && n_dims == 5 ? sizeof(float) * diffsal_attention_general_tail_floats(D(0), D(1), D(2), D(3), D(4)) : 0;
    case DIFFSAL_WS_ATTENTION_BWD_QTAIL:
      return have && n_dims == 6 ? sizeof(float) * diffsal_attention_general_bwd_qtail_floats(D(0), D(1), D(2), D(3), D(4), D(5)) : 0;
    case DIFFSAL_WS_ATTENTION_BWD_DS:
      return have && n_dims == 4 ? sizeof(float) * diffsal_attention_general_bwd_ds_floats(D(0), D(1), D(2), D(3)) : 0;
    default: return 0;
  }
}

extern "C" int diffsal_version(void) { return 42; }  // = _lib.ABI_VERSION
extern "C" const char* diffsal_last_error(void) { return g_err; }
extern "C" const char* diffsal_last_gemm_kernel(void) { return g_kernel; }

extern "C" int diffsal_set_tuning(const char* name, int value) {
  DS_REQUIRE(name, DIFFSAL_E_ARG, "set_tuning: null name");
  for (int k = 0; k < TUNE_COUNT; ++k)
    if (strcmp(name, kTuneNames[k]) == 0) {
      __atomic_store_n(&g_tune[k], value, __ATOMIC_RELAXED);
      return DIFFSAL_OK;
    }
  set_error("set_tuning: unknown switch '%s'", name);
  return DIFFSAL_E_ARG;
}
extern "C" int diffsal_get_tuning(const char* name) {
  if (name)
    for (int k = 0; k < TUNE_COUNT; ++k)
      if (strcmp(name, kTuneNames[k]) == 0) return tune(k);
  return -1;
}

extern "C" int diffsal_temb_mlp(const void* t, int t_is_f32, int B, int ch, const float* freq, const float* w0,
                                const float* b0, const float* w1, const float* b1, float* hidden_ws, float* temb_out,
                                diffsal_stream_t stream) {
  DS_REQUIRE(t && freq && w0 && b0 && w1 && b1 && hidden_ws && temb_out, DIFFSAL_E_ARG, "temb_mlp: null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tc = 4 * ch;
  if (B > 16 && B <= 64 && static_cast<size_t>(tc) * 65 * sizeof(float) <= 160 * 1024 && ch >= 4) {     // a lane per clip
    const size_t l0 = static_cast<size_t>(ch) * 65 * sizeof(float), l1 = static_cast<size_t>(tc) * 65 * sizeof(float);
    DS_RAISE_DYNAMIC_LDS((dense_lanes_kernel<false>), 160 * 1024);
    hipLaunchKernelGGL((dense_lanes_kernel<true>), dim3((tc + 3) / 4), dim3(256), l0, s, t, t_is_f32, freq, static_cast<const float*>(nullptr),
                       B, ch, 0, 1, w0, b0, tc, hidden_ws);
    int rc = check_launch("temb_mlp(dense0)");
    if (rc) return rc;
    hipLaunchKernelGGL((dense_lanes_kernel<false>), dim3((tc + 3) / 4), dim3(256), l1, s, static_cast<const void*>(nullptr), 0,
                       static_cast<const float*>(nullptr), hidden_ws, B, tc, 0, 0, w1, b1, tc, temb_out);
    return check_launch("temb_mlp(dense1)");
  }
  DS_REQUIRE(B > 0 && ch >= 4 && ch <= 1024 && static_cast<long>(B) * ch * 16 <= 64 * 1024, DIFFSAL_E_SHAPE,
             "temb_mlp: bad shape B=%d ch=%d", B, ch);
  hipLaunchKernelGGL(temb_dense0_kernel, dim3((tc + 3) / 4), dim3(256), static_cast<size_t>(B) * ch * sizeof(float), s,
                     t, t_is_f32, B, ch, freq, w0, b0, hidden_ws);
  int rc = check_launch("temb_mlp(dense0)");
  if (rc) return rc;
  hipLaunchKernelGGL(dense_small_kernel, dim3((tc + 3) / 4), dim3(256), static_cast<size_t>(B) * tc * sizeof(float), s,
                     hidden_ws, B, tc, 0, w1, b1, tc, temb_out);
  return check_launch("temb_mlp(dense1)");
}

extern "C" int diffsal_dense_small(const float* in, int B, int K, int swish_in, const float* w, const float* bias,
                                   int N, float* out, diffsal_stream_t stream) {
  DS_REQUIRE(in && w && out, DIFFSAL_E_ARG, "dense_small: null argument");
  if (B > 16 && B <= 64 && K > 0 && N > 0 && static_cast<size_t>(K) * 65 * sizeof(float) <= 160 * 1024) {     // a lane per clip
    DS_RAISE_DYNAMIC_LDS((dense_lanes_kernel<false>), 160 * 1024);
    hipLaunchKernelGGL((dense_lanes_kernel<false>), dim3((N + 3) / 4), dim3(256), static_cast<size_t>(K) * 65 * sizeof(float),
                       static_cast<hipStream_t>(stream), static_cast<const void*>(nullptr), 0, static_cast<const float*>(nullptr), in, B, K,
                       swish_in, 0, w, bias, N, out);
    return check_launch("dense_small");
  }
  DS_REQUIRE(B > 0 && K > 0 && N > 0 && static_cast<long>(B) * K * 4 <= 64 * 1024, DIFFSAL_E_SHAPE,
             "dense_small: bad shape B=%d K=%d N=%d", B, K, N);
  hipLaunchKernelGGL(dense_small_kernel, dim3((N + 3) / 4), dim3(256), static_cast<size_t>(B) * K * sizeof(float),
                     static_cast<hipStream_t>(stream), in, B, K, swish_in, w, bias, N, out);
  return check_launch("dense_small");
}

extern "C" int diffsal_conv_in(const float* x, const float* w, const float* bias, void* out, int B, int H, int W,
                               int C, int skip_mod, int act, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(act == DIFFSAL_ACT_NONE || act == DIFFSAL_ACT_RELU, DIFFSAL_E_ARG, "conv_in: act %d (none or ReLU)", act);
  DS_REQUIRE(x && w && bias && out, DIFFSAL_E_ARG, "conv_in: null argument");
  DS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "conv_in: bad shape");
  DS_REQUIRE(aligned16(out), DIFFSAL_E_ALIGN, "conv_in: misaligned output");
  DS_REQUIRE(C <= 1024 && skip_mod != 1, DIFFSAL_E_SHAPE, "conv_in: C=%d skip_mod=%d unsupported", C, skip_mod);
  const int keep = skip_mod > 0 ? skip_mod - 1 : 1;
  const int Hn = skip_mod > 0 ? (H / skip_mod) * keep + (H % skip_mod < keep ? H % skip_mod : keep) : H;
  const int Wn = skip_mod > 0 ? (W / skip_mod) * keep + (W % skip_mod < keep ? W % skip_mod : keep) : W;
  const int ppb = 256 / (C / 4);
  long g = (static_cast<long>(B) * Hn * Wn + ppb - 1) / ppb;
  g = g > 8192 ? 8192 : g;
#define CALL(T)                                                                                                     \
  hipLaunchKernelGGL((conv_in_kernel<T>), dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), x, \
                     w, bias, static_cast<T*>(out), B, H, W, C, skip_mod, Hn, Wn, act)
  DS_DTYPE_DISPATCH(dtype, "conv_in", CALL);
#undef CALL
  return check_launch("conv_in");
}

extern "C" int diffsal_pack_frames(const float* vis, const void* noise, void* out, int B, int C, int Tv, int Tout,
                                   int hw, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(vis && out, DIFFSAL_E_ARG, "pack_frames: null argument");
  DS_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && Tv > 0 && hw > 0 && Tout >= Tv + (noise ? 1 : 0), DIFFSAL_E_SHAPE,
             "pack_frames: bad shape C=%d Tv=%d Tout=%d hw=%d", C, Tv, Tout, hw);
  DS_REQUIRE(aligned16(out) && (!noise || aligned16(noise)), DIFFSAL_E_ALIGN, "pack_frames: misaligned pointer");
  const int Q = Tv * hw;
  const int tiles_q = (Q + 63) / 64, tiles_c = (C + 63) / 64;
  const int ntb = tiles_q * tiles_c;
  const int copy_blocks = noise ? ew_grid(static_cast<long>(hw) * C / 4) : 0;
#define CALL(T)                                                                                                    \
  hipLaunchKernelGGL((pack_frames_kernel<T>), dim3(ntb + copy_blocks, B), dim3(256), 0, static_cast<hipStream_t>(stream), \
                     vis, static_cast<const T*>(noise), static_cast<T*>(out), C, Q, hw, Tout, tiles_q, tiles_c, ntb)
  DS_DTYPE_DISPATCH(dtype, "pack_frames", CALL);
#undef CALL
  return check_launch("pack_frames");
}

extern "C" int diffsal_pack_frames_multi(const float* const* vis, const void* const* noise, void* const* out, int n, int B,
                                         const int* C, const int* Tv, const int* Tout, const int* hw, int dtype,
                                         diffsal_stream_t stream) {
  DS_REQUIRE(vis && noise && out && C && Tv && Tout && hw, DIFFSAL_E_ARG, "pack_frames_multi: null argument");
  DS_REQUIRE(n >= 1 && n <= 4 && B > 0, DIFFSAL_E_SHAPE, "pack_frames_multi: 1..4 problems, got %d", n);
  PackMulti m;
  m.n = n;
  long total = 0;
  for (int j = 0; j < 4; ++j) {
    const int jj = j < n ? j : 0;
    DS_REQUIRE(vis[jj] && out[jj], DIFFSAL_E_ARG, "pack_frames_multi: null tensor %d", jj);
    DS_REQUIRE(C[jj] > 0 && C[jj] % 4 == 0 && Tv[jj] > 0 && hw[jj] > 0 && Tout[jj] >= Tv[jj] + (noise[jj] ? 1 : 0), DIFFSAL_E_SHAPE,
               "pack_frames_multi: bad shape C=%d Tv=%d Tout=%d hw=%d", C[jj], Tv[jj], Tout[jj], hw[jj]);
    DS_REQUIRE(aligned16(out[jj]) && (!noise[jj] || aligned16(noise[jj])), DIFFSAL_E_ALIGN, "pack_frames_multi: misaligned pointer");
    m.vis[j] = vis[jj]; m.noise[j] = noise[jj]; m.out[j] = out[jj];
    m.C[j] = C[jj]; m.Q[j] = Tv[jj] * hw[jj]; m.hw[j] = hw[jj]; m.Tout[j] = Tout[jj];
    m.tiles_q[j] = (m.Q[j] + 63) / 64;
    m.ntb[j] = m.tiles_q[j] * ((C[jj] + 63) / 64);
    m.blocks[j] = m.ntb[j] + (noise[jj] ? ew_grid(static_cast<long>(hw[jj]) * C[jj] / 4) : 0);
    if (j < n) total += m.blocks[j];
  }
  DS_REQUIRE(total < (1L << 31), DIFFSAL_E_SHAPE, "pack_frames_multi: too many blocks");
#define CALL(T)                                                                                                        \
  hipLaunchKernelGGL((pack_frames_multi_kernel<T>), dim3(static_cast<unsigned>(total), B), dim3(256), 0,              \
                     static_cast<hipStream_t>(stream), m)
  DS_DTYPE_DISPATCH(dtype, "pack_frames_multi", CALL);
#undef CALL
  return check_launch("pack_frames_multi");
}

template <typename T>
static void resize_bilinear_t(const T* in, T* out, int N, int h, int w, int H, int W, int C, hipStream_t s) {
  const float sy = static_cast<float>(h) / static_cast<float>(H), sx = static_cast<float>(w) / static_cast<float>(W);
  const int f = H / h;
  // x2 is cheaper with one thread per output (4 loads, mostly L1 hits); the patch form pays from x4 up
  if (C % 4 == 0 && aligned16(in) && aligned16(out) && H % 4 == 0 && W % 4 == 0 && f >= 2 && (f & (f - 1)) == 0 &&
      h * f == H && w * f == W) {
    ResizeSumArgs a;
    a.n_in = 1;
    for (int i = 0; i < 4; ++i) { a.in[i] = in; a.h[i] = h; a.w[i] = w; a.sy[i] = sy; a.sx[i] = sx; }
    const long n_patches = static_cast<long>(N) * (H / 4) * (W / 4);
    const long waves = n_patches * ((C + 255) / 256);
    hipLaunchKernelGGL((resize_sum_kernel<T>), dim3(static_cast<int>((waves + 3) / 4)), dim3(256), 0, s, a, out, H, W,
                       C, W / 4, n_patches);
  } else if (C % 4 == 0 && aligned16(in) && aligned16(out)) {
    const long total = static_cast<long>(N) * H * W * (C / 4);
    hipLaunchKernelGGL((resize_kernel<4, T>), dim3(ew_grid(total)), dim3(256), 0, s, in, out, N, h, w, H, W, C, sy, sx);
  } else {
    const long total = static_cast<long>(N) * H * W * C;
    hipLaunchKernelGGL((resize_kernel<1, T>), dim3(ew_grid(total)), dim3(256), 0, s, in, out, N, h, w, H, W, C, sy, sx);
  }
}

// ------------------------------------------------------------------------------------------------
// K2 fused: conv_in (1 -> C, 3x3, pad 1) followed, with nothing in between, by Downsample4x4's 3x3 stride-4 convolution
// (R/.../sal_unet.py:240,292 and :67-84) is ONE linear map of the single-channel input: a 5x5 stride-4 convolution whose
// weights  W_eff[co][dy][dx] = sum_ci sum_{ky2+ky1=dy, kx2+kx1=dx} W2[co,ci,ky2,kx2] W1[ci,ky1,kx1]  and bias
// b_eff = b2 + sum W2 b1  the host composes once per parameter update (in fp64).  Exact for H, W multiples of 4: the
// stride-4 windows then never reach Downsample4x4's zero row / column, every conv_in output they read is a genuine one, and
// conv_in's own padding is the zero ring of x on the top / left.  Replaces a 132 MB intermediate (fp32) and a 3.6 GFLOP GEMM
// by 25 multiply-adds per output value.  One thread per (output pixel, channel quad); the 25 inputs of a pixel are
// broadcast loads, the weights live in LDS.
// ------------------------------------------------------------------------------------------------
namespace diffsal {
template <typename T>
__global__ __launch_bounds__(256) void conv_in_s4_kernel(const float* __restrict__ x, const float* __restrict__ w25,
                                                         const float* __restrict__ bias, T* __restrict__ out, int H, int W,
                                                         int C, long total4) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];   // [25][C] then bias [C]
  for (int i = threadIdx.x; i < 25 * C; i += 256) wsm[i] = w25[i];
  for (int i = threadIdx.x; i < C; i += 256) wsm[25 * C + i] = bias[i];
  __syncthreads();
  const int c4n = C >> 2, Ho = H >> 2, Wo = W >> 2;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total4; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    long r = i / c4n;
    const int ox = static_cast<int>(r % Wo); r /= Wo;
    const int oy = static_cast<int>(r % Ho);
    const long n = r / Ho;
    const float* xb = x + n * H * W;
    float4 acc = ld4(wsm + 25 * C + c);
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
      const int iy = 4 * oy - 1 + dy;
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const int ix = 4 * ox - 1 + dx;
        const bool ok = iy >= 0 && ix >= 0;                       // bottom / right never leave the image (H, W % 4 == 0)
        float v = xb[static_cast<long>(ok ? iy : 0) * W + (ok ? ix : 0)];   // unconditional load from a clamped address
        v = ok ? v : 0.f;
        const float4 ww = ld4(wsm + (dy * 5 + dx) * C + c);
        acc.x = fmaf(v, ww.x, acc.x); acc.y = fmaf(v, ww.y, acc.y); acc.z = fmaf(v, ww.z, acc.z); acc.w = fmaf(v, ww.w, acc.w);
      }
    }
    st4(out + i * 4, acc);
  }
}
}  // namespace diffsal
using namespace diffsal;

extern "C" int diffsal_conv_in_s4(const float* x, const float* w25, const float* bias, void* out, int B, int H, int W, int C,
                                  int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(x && w25 && bias && out, DIFFSAL_E_ARG, "conv_in_s4: null argument");
  DS_REQUIRE(B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && C > 0 && C % 4 == 0 && 26 * C * 4 <= 64 * 1024, DIFFSAL_E_SHAPE,
             "conv_in_s4: bad shape B=%d H=%d W=%d C=%d (H, W multiples of 4)", B, H, W, C);
  DS_REQUIRE(aligned16(out) && aligned16(w25) && aligned16(bias), DIFFSAL_E_ALIGN, "conv_in_s4: misaligned pointer");
  const long total4 = static_cast<long>(B) * (H / 4) * (W / 4) * (C / 4);
#define CALL(T)                                                                                                        \
  hipLaunchKernelGGL((conv_in_s4_kernel<T>), dim3(ew_grid(total4)), dim3(256), 26 * C * sizeof(float),                   \
                     static_cast<hipStream_t>(stream), x, w25, bias, static_cast<T*>(out), H, W, C, total4)
  DS_DTYPE_DISPATCH(dtype, "conv_in_s4", CALL);
#undef CALL
  return check_launch("conv_in_s4");
}

// ------------------------------------------------------------------------------------------------
// Pieces of the legacy DDPM-style UNet (R/models/diffusion_decoder/diffusion.py), fp32, NHWC:
//   softmax_rows      out[r, :] = softmax(scale * x[r, :])               AttnBlock :166-168 (full HW x HW attention)
//   upsample_nearest2 out[n, 2y+dy, 2x+dx, :] = in[n, y, x, :]           Upsample :46-47
//   avgpool2          out[n, y, x, :] = mean of the 2x2 window           Downsample without conv :69
//   sigmoid_gate      out = sigmoid(y) * x                               feat_interact :318
// ------------------------------------------------------------------------------------------------
namespace diffsal {
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ out, long rows,
                                                           int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);   // one wavefront per row
  if (row >= rows) return;
  const float* xr = x + row * cols;
  float* orow = out + row * cols;
  float mx = -3.0e38f;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xr[c] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, kWave));
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) sum += expf(xr[c] * scale - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, kWave);
  const float inv = 1.0f / sum;
  for (int c = lane; c < cols; c += 64) orow[c] = expf(xr[c] * scale - mx) * inv;
}

__global__ __launch_bounds__(256) void upsample_nearest2_kernel(const float* __restrict__ in, float* __restrict__ out, int H,
                                                                int W, int C, long total4) {
  const int c4n = C >> 2;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total4; i += static_cast<long>(gridDim.x) * 256) {
    const int c4 = static_cast<int>(i % c4n);
    long r = i / c4n;
    const int X = static_cast<int>(r % (2 * W)); r /= 2 * W;
    const int Y = static_cast<int>(r % (2 * H));
    const long n = r / (2 * H);
    st4(out + i * 4, ld4(in + ((n * H + (Y >> 1)) * W + (X >> 1)) * C + c4 * 4));
  }
}

__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                       int C, long total4) {
  const int c4n = C >> 2, Ho = H >> 1, Wo = W >> 1;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total4; i += static_cast<long>(gridDim.x) * 256) {
    const int c4 = static_cast<int>(i % c4n);
    long r = i / c4n;
    const int X = static_cast<int>(r % Wo); r /= Wo;
    const int Y = static_cast<int>(r % Ho);
    const long n = r / Ho;
    const float* b = in + ((n * H + 2 * Y) * W + 2 * X) * C + c4 * 4;
    const float4 a = ld4(b), bb = ld4(b + C), c = ld4(b + static_cast<long>(W) * C), d = ld4(b + static_cast<long>(W) * C + C);
    // torch.avg_pool2d sums the window in (row, column) order and multiplies by 1 / 4
    st4(out + i * 4, make_float4(((a.x + bb.x) + c.x + d.x) * 0.25f, ((a.y + bb.y) + c.y + d.y) * 0.25f,
                                 ((a.z + bb.z) + c.z + d.z) * 0.25f, ((a.w + bb.w) + c.w + d.w) * 0.25f));
  }
}

__global__ __launch_bounds__(256) void sigmoid_gate_kernel(const float* __restrict__ y, const float* __restrict__ x,
                                                           float* __restrict__ out, long n4) {
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * 256) {
    const float4 a = ld4(y + i * 4), b = ld4(x + i * 4);
    st4(out + i * 4, make_float4(sigmoidf_(a.x) * b.x, sigmoidf_(a.y) * b.y, sigmoidf_(a.z) * b.z, sigmoidf_(a.w) * b.w));
  }
}

}  // namespace diffsal
using namespace diffsal;

extern "C" int diffsal_softmax_rows(const float* x, float* out, long rows, int cols, float scale, diffsal_stream_t stream) {
  DS_REQUIRE(x && out && rows > 0 && cols > 0, DIFFSAL_E_ARG, "softmax_rows: bad argument");
  DS_REQUIRE((rows + 3) / 4 < (1L << 31), DIFFSAL_E_SHAPE, "softmax_rows: too many rows");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, out, rows, cols, scale);
  return check_launch("softmax_rows");
}

extern "C" int diffsal_upsample_nearest2(const float* in, float* out, int N, int H, int W, int C, diffsal_stream_t stream) {
  DS_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "upsample_nearest2: bad shape");
  DS_REQUIRE(aligned16(in) && aligned16(out), DIFFSAL_E_ALIGN, "upsample_nearest2: misaligned pointer");
  const long total4 = static_cast<long>(N) * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(upsample_nearest2_kernel, dim3(ew_grid(total4)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, H, W, C,
                     total4);
  return check_launch("upsample_nearest2");
}

extern "C" int diffsal_avgpool2(const float* in, float* out, int N, int H, int W, int C, diffsal_stream_t stream) {
  DS_REQUIRE(in && out && N > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "avgpool2: bad shape");
  DS_REQUIRE(aligned16(in) && aligned16(out), DIFFSAL_E_ALIGN, "avgpool2: misaligned pointer");
  const long total4 = static_cast<long>(N) * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool2_kernel, dim3(ew_grid(total4)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, H, W, C, total4);
  return check_launch("avgpool2");
}

extern "C" int diffsal_sigmoid_gate(const float* y, const float* x, float* out, long n, diffsal_stream_t stream) {
  DS_REQUIRE(y && x && out && n > 0 && n % 4 == 0, DIFFSAL_E_SHAPE, "sigmoid_gate: bad argument (n %% 4 == 0)");
  DS_REQUIRE(aligned16(y) && aligned16(x) && aligned16(out), DIFFSAL_E_ALIGN, "sigmoid_gate: misaligned pointer");
  hipLaunchKernelGGL(sigmoid_gate_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), y, x, out, n / 4);
  return check_launch("sigmoid_gate");
}

extern "C" int diffsal_resize_bilinear(const void* in, void* out, int N, int h, int w, int H, int W, int C, int dtype,
                                       diffsal_stream_t stream) {
  DS_REQUIRE(in && out, DIFFSAL_E_ARG, "resize_bilinear: null argument");
  DS_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, DIFFSAL_E_SHAPE, "resize_bilinear: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(T) resize_bilinear_t<T>(static_cast<const T*>(in), static_cast<T*>(out), N, h, w, H, W, C, s)
  DS_DTYPE_DISPATCH(dtype, "resize_bilinear", CALL);
#undef CALL
  return check_launch("resize_bilinear");
}

extern "C" int diffsal_resize_sum(const void* const* ins, const int* hs, const int* ws, int n_in, void* out, int N,
                                  int H, int W, int C, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(ins && hs && ws && out, DIFFSAL_E_ARG, "resize_sum: null argument");
  DS_REQUIRE(n_in >= 1 && n_in <= 4 && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE,
             "resize_sum: bad shape n_in=%d C=%d", n_in, C);
  ResizeSumArgs a;
  a.n_in = n_in;
  for (int i = 0; i < 4; ++i) {
    a.in[i] = i < n_in ? ins[i] : nullptr;
    a.h[i] = i < n_in ? hs[i] : 1;
    a.w[i] = i < n_in ? ws[i] : 1;
    a.sy[i] = static_cast<float>(a.h[i]) / static_cast<float>(H);
    a.sx[i] = static_cast<float>(a.w[i]) / static_cast<float>(W);
    if (i < n_in) DS_REQUIRE(ins[i] && aligned16(ins[i]) && hs[i] > 0 && ws[i] > 0, DIFFSAL_E_ARG, "resize_sum: bad input %d", i);
  }
  DS_REQUIRE(aligned16(out), DIFFSAL_E_ALIGN, "resize_sum: misaligned output");
  bool patchable = (H % 4 == 0) && (W % 4 == 0);
  for (int i = 0; i < n_in; ++i) {
    const int f = H / hs[i];
    patchable = patchable && f >= 2 && (f & (f - 1)) == 0 && hs[i] * f == H && ws[i] * f == W;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (patchable) {
    const long n_patches = static_cast<long>(N) * (H / 4) * (W / 4);
    const long waves = n_patches * ((C + 255) / 256);
#define CALL(T)                                                                                                        \
  hipLaunchKernelGGL((resize_sum_kernel<T>), dim3(static_cast<int>((waves + 3) / 4)), dim3(256), 0, s, a,              \
                     static_cast<T*>(out), H, W, C, W / 4, n_patches)
    DS_DTYPE_DISPATCH(dtype, "resize_sum", CALL);
#undef CALL
  } else {
    const int w_tiles = (W + 3) / 4;
    const long blocks = static_cast<long>(N) * H * w_tiles;
    DS_REQUIRE(blocks < (1L << 31), DIFFSAL_E_SHAPE, "resize_sum: output too large");
#define CALL(T)                                                                                                        \
  hipLaunchKernelGGL((resize_sum_generic_kernel<T>), dim3(static_cast<int>(blocks)), dim3(256), 0, s, a,               \
                     static_cast<T*>(out), H, W, C, w_tiles)
    DS_DTYPE_DISPATCH(dtype, "resize_sum", CALL);
#undef CALL
  }
  return check_launch("resize_sum");
}

extern "C" int diffsal_audio_fuse(const void* a_small, int a_ld, const void* x, void* out, int B, int T, int H, int W, int C,
                                  int h, int w, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(a_small && x && out, DIFFSAL_E_ARG, "audio_fuse: null argument");
  DS_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && h > 0 && w > 0 && a_ld >= C && a_ld % 8 == 0, DIFFSAL_E_SHAPE,
             "audio_fuse: bad shape C=%d a_ld=%d", C, a_ld);
  DS_REQUIRE(aligned16(a_small) && aligned16(x), DIFFSAL_E_ALIGN, "audio_fuse: misaligned pointer");
  int up = 1;
  if (h != H && w != W) {  // quirk Q3: upsample only when BOTH differ; factor H // h
    up = H / h;
    DS_REQUIRE(up >= 1 && h * up == H && w * up == W, DIFFSAL_E_SHAPE,
               "audio_fuse: audio map %dx%d does not upsample to %dx%d by an integer factor", h, w, H, W);
  } else {
    DS_REQUIRE(h == H && w == W, DIFFSAL_E_SHAPE, "audio_fuse: audio map %dx%d incompatible with %dx%d", h, w, H, W);
  }
  const int ups = up == 1 ? 0 : up == 2 ? 1 : up == 4 ? 2 : up == 8 ? 3 : -1;
  if (dtype != DIFFSAL_F32 && tune(TUNE_NO_STREAM16) != 1 && (C % 64 == 0 || C == 96) && W % 4 == 0 && aligned16(out) && a_ld % 8 == 0 && ups >= 0) {
    // 16-bit storage: 16-byte accesses, whole 128-byte lines per slab, R rows per workgroup (audio_fuse16_kernel)
    const int CS = C % 64 == 0 ? 64 : 96;
    const int XV = W % 8 == 0 ? 8 : 4;
    int R = H;
    for (int r = 1; r <= H; ++r)
      if (H % r == 0 && r * W >= 96) { R = r; break; }
    const int RS = R % up == 0 ? R / up : (up % R == 0 ? 1 : (R - 1) / up + 2);     // audio rows under a block of R rows
    const size_t lds16 = static_cast<size_t>(CS) * (R * W + 1) * sizeof(float) + static_cast<size_t>(2) * RS * w * (CS + 2) * 2;
    const long items = static_cast<long>(CS) * R * (W / XV);
    if (lds16 <= 64 * 1024 && RS * w * (CS / 8) <= 768 && items <= 6 * 256 && static_cast<long>(R) * W * W < (1 << 20) && items * (W / XV) < (1 << 20)) {
      const int grid = B * (H / R) * (C / CS);
      hipStream_t s = static_cast<hipStream_t>(stream);
      AudioFuse16Args g{T, H, W, C, h, w, a_ld, R, RS, make_magic(W), make_magic(R), make_magic(W / XV)};
#define CALL16(TT, CSV, XVV, UPSV)                                                                                      \
  hipLaunchKernelGGL((audio_fuse16_kernel<TT, CSV, XVV, UPSV>), dim3(grid), dim3(256), lds16, s, static_cast<const TT*>(a_small), \
                     static_cast<const TT*>(x), static_cast<TT*>(out), g)
#define CALL16_U(TT, CSV, XVV)                                                                             \
  do {                                                                                                     \
    if (ups == 0) CALL16(TT, CSV, XVV, 0); else if (ups == 1) CALL16(TT, CSV, XVV, 1);                     \
    else if (ups == 2) CALL16(TT, CSV, XVV, 2); else CALL16(TT, CSV, XVV, 3);                              \
  } while (0)
#define CALL16_T(TT)                                                                                       \
  do {                                                                                                     \
    if (CS == 64) { if (XV == 8) CALL16_U(TT, 64, 8); else CALL16_U(TT, 64, 4); }                          \
    else { if (XV == 8) CALL16_U(TT, 96, 8); else CALL16_U(TT, 96, 4); }                                   \
  } while (0)
      if (dtype == DIFFSAL_BF16) CALL16_T(bf16_t); else CALL16_T(f16_t);
#undef CALL16_T
#undef CALL16_U
#undef CALL16
      return check_launch("audio_fuse(16-bit)");
    }
  }
  const size_t lds = static_cast<size_t>(32) * (W + 1) * sizeof(float);
#define CALL(TT)                                                                                                     \
  hipLaunchKernelGGL((audio_fuse_kernel<TT>), dim3(B * H * (C / 32)), dim3(256), lds, static_cast<hipStream_t>(stream), \
                     static_cast<const TT*>(a_small), static_cast<const TT*>(x), static_cast<TT*>(out), T, H, W, C, h, w, up, a_ld)
  DS_DTYPE_DISPATCH(dtype, "audio_fuse", CALL);
#undef CALL
  return check_launch("audio_fuse");
}

template <int LK, int G, typename T>
static void launch_attention(const T* q, const T* k, const T* v, T* o, int N, int Lq, int Lk, int C,
                             int heads, float scale, hipStream_t s) {
  const size_t lds = static_cast<size_t>(2) * Lk * (C / heads) * sizeof(float);
  if (lds > 64 * 1024) DS_RAISE_DYNAMIC_LDS((attention_kernel<LK, G, T>), 160 * 1024);
  const int qpb = 4 * (64 / G);  // queries per workgroup (all four wavefronts serve the same head)
  hipLaunchKernelGGL((attention_kernel<LK, G, T>), dim3((Lq + qpb - 1) / qpb, N * heads), dim3(256), lds, s, q, k, v, o,
                     Lq, Lk, C, heads, scale);
}

template <int LK, typename T>
static void launch_attention_g(int G, const T* q, const T* k, const T* v, T* o, int N, int Lq, int Lk,
                               int C, int heads, float scale, hipStream_t s) {
  switch (G) {
    case 16: launch_attention<LK, 16, T>(q, k, v, o, N, Lq, Lk, C, heads, scale, s); break;
    case 8: launch_attention<LK, 8, T>(q, k, v, o, N, Lq, Lk, C, heads, scale, s); break;
    case 4: launch_attention<LK, 4, T>(q, k, v, o, N, Lq, Lk, C, heads, scale, s); break;
    default: launch_attention<LK, 1, T>(q, k, v, o, N, Lq, Lk, C, heads, scale, s); break;
  }
}

template <typename T>
static void attention_t(const T* q, const T* k, const T* v, T* o, int N, int Lq, int Lk, int C, int heads, float scale,
                        hipStream_t s) {
  // lanes per (query, head): about 3 float4 pieces of the head dim per lane, at most 16 lanes
  const int nf4 = C / heads / 4;
  int G = 1;
  while (G < 16 && nf4 / (2 * G) >= 3) G *= 2;
  if (G == 2) G = 4;
  if (Lk <= 4) launch_attention_g<4, T>(G, q, k, v, o, N, Lq, Lk, C, heads, scale, s);
  else if (Lk <= 8) launch_attention_g<8, T>(G, q, k, v, o, N, Lq, Lk, C, heads, scale, s);
  else if (Lk <= 18) launch_attention_g<18, T>(G, q, k, v, o, N, Lq, Lk, C, heads, scale, s);
  else launch_attention_g<32, T>(G, q, k, v, o, N, Lq, Lk, C, heads, scale, s);
}

namespace diffsal {
int try_attention16_mfma(const void* q, const void* k, const void* v, void* o, int N, int Lq, int Lk, int C, int heads, float scale,
                         int dtype, hipStream_t s);
}
extern "C" int diffsal_attention(const void* q, const void* k, const void* v, void* o, int N, int Lq, int Lk,
                                 int C, int heads, float scale, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(q && k && v && o, DIFFSAL_E_ARG, "attention: null argument");
  DS_REQUIRE(N > 0 && Lq > 0 && Lk > 0 && Lk <= 32, DIFFSAL_E_SHAPE, "attention: bad shape Lq=%d Lk=%d", Lq, Lk);
  DS_REQUIRE(heads >= 1 && static_cast<long>(N) * heads < 65536, DIFFSAL_E_SHAPE, "attention: heads=%d", heads);
  DS_REQUIRE(C % heads == 0 && (C / heads) % 4 == 0, DIFFSAL_E_SHAPE,
             "attention: C=%d heads=%d: head dim must be a multiple of 4", C, heads);
  DS_REQUIRE(static_cast<size_t>(2) * Lk * (C / heads) * sizeof(float) <= 160 * 1024, DIFFSAL_E_SHAPE,
             "attention: K/V tile of one head exceeds LDS (Lk=%d C=%d heads=%d)", Lk, C, heads);
  DS_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o), DIFFSAL_E_ALIGN,
             "attention: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    // 16-bit storage, head dim 192 / 384 (stages 1 / 0): the matrix-core form (csrc/attn16_mfma.hip)
    const int rc = diffsal::try_attention16_mfma(q, k, v, o, N, Lq, Lk, C, heads, scale, dtype, s);
    if (rc != 0) return rc < 0 ? rc : DIFFSAL_OK;
  }
  {
    // 16-bit storage, head dims 48 .. 384, up to 18 keys: the 16-byte form, one workgroup per (image, head)
    const int d = C / heads, G = d / 24;
    const size_t lds16 = static_cast<size_t>(2) * Lk * d * sizeof(float);
    if (dtype != DIFFSAL_F32 && tune(TUNE_NO_STREAM16) != 1 && d % 24 == 0 && (G == 2 || G == 4 || G == 8 || G == 16) && Lk <= 18 && C % 8 == 0 &&
        lds16 <= 64 * 1024) {
      // query chunks: whole passes of 2 * 256 / G queries, as many as it takes to put ~2 workgroups on every CU
      const int pass_q = 2 * 256 / G, passes = (Lq + pass_q - 1) / pass_q;
      int chunks = (512 + N * heads - 1) / (N * heads);
      chunks = chunks < 1 ? 1 : (chunks > passes ? passes : chunks);
      const int q_chunk = (passes + chunks - 1) / chunks * pass_q;
      const dim3 grid(static_cast<unsigned>(N * heads), static_cast<unsigned>((Lq + q_chunk - 1) / q_chunk));
#define CALL16(TT, GV) \
  hipLaunchKernelGGL((attention16_kernel<TT, GV, 18>), grid, dim3(256), lds16, s, static_cast<const TT*>(q), static_cast<const TT*>(k), \
                     static_cast<const TT*>(v), static_cast<TT*>(o), Lq, Lk, C, heads, scale, q_chunk)
#define CALL16_G(TT) \
  do { if (G == 2) CALL16(TT, 2); else if (G == 4) CALL16(TT, 4); else if (G == 8) CALL16(TT, 8); else CALL16(TT, 16); } while (0)
      if (dtype == DIFFSAL_BF16) CALL16_G(bf16_t); else CALL16_G(f16_t);
#undef CALL16_G
#undef CALL16
      return check_launch("attention(16-bit)");
    }
  }
#define CALL(T) \
  attention_t<T>(static_cast<const T*>(q), static_cast<const T*>(k), static_cast<const T*>(v), static_cast<T*>(o), N, Lq, Lk, C, heads, scale, s)
  DS_DTYPE_DISPATCH(dtype, "attention", CALL);
#undef CALL
  return check_launch("attention");
}

extern "C" int diffsal_head_sigmoid(const void* x, const float* w, const float* bias, float* out, int NHW, int C,
                                    int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(x && w && bias && out, DIFFSAL_E_ARG, "head_sigmoid: null argument");
  DS_REQUIRE(NHW > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "head_sigmoid: bad shape");
  DS_REQUIRE(aligned16(x) && aligned16(w), DIFFSAL_E_ALIGN, "head_sigmoid: misaligned pointer");
  long g = (static_cast<long>(NHW) + 31) / 32;
  g = g > 4096 ? 4096 : g;
#define CALL(T)                                                                                               \
  hipLaunchKernelGGL((head_kernel<T>), dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), \
                     static_cast<const T*>(x), w, bias, out, static_cast<long>(NHW), C)
  DS_DTYPE_DISPATCH(dtype, "head_sigmoid", CALL);
#undef CALL
  return check_launch("head_sigmoid");
}

extern "C" int diffsal_axpbypcz(const float* x, const float* y, const float* z, float a, float b, float c, float* out,
                                size_t n, diffsal_stream_t stream) {
  DS_REQUIRE(x && out, DIFFSAL_E_ARG, "axpbypcz: null argument");
  if (n == 0) return DIFFSAL_OK;
  hipLaunchKernelGGL(axpbypcz_kernel, dim3(ew_grid(static_cast<long>(n))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, z, a, b, c, out, n);
  return check_launch("axpbypcz");
}

extern "C" int diffsal_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, diffsal_stream_t stream) {
  DS_REQUIRE(src && dst, DIFFSAL_E_ARG, "cast: null argument");
  if (n <= 0) return DIFFSAL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int g = ew_grid(n);
#define CALL_D(D)                                                                                      \
  do {                                                                                                 \
    if (src_dtype == DIFFSAL_F32)                                                                      \
      hipLaunchKernelGGL((cast_kernel<float, D>), dim3(g), dim3(256), 0, s, static_cast<const float*>(src), static_cast<D*>(dst), n); \
    else if (src_dtype == DIFFSAL_BF16)                                                                \
      hipLaunchKernelGGL((cast_kernel<bf16_t, D>), dim3(g), dim3(256), 0, s, static_cast<const bf16_t*>(src), static_cast<D*>(dst), n); \
    else if (src_dtype == DIFFSAL_F16)                                                                 \
      hipLaunchKernelGGL((cast_kernel<f16_t, D>), dim3(g), dim3(256), 0, s, static_cast<const f16_t*>(src), static_cast<D*>(dst), n); \
    else { set_error("cast: src dtype %d", src_dtype); return DIFFSAL_E_ARG; }                         \
  } while (0)
  DS_DTYPE_DISPATCH(dst_dtype, "cast", CALL_D);
#undef CALL_D
  return check_launch("cast");
}

extern "C" int diffsal_maxpool2d(const void* in, void* out, int N, int H, int W, int C, int k, int stride, int dtype,
                                 diffsal_stream_t stream) {
  DS_REQUIRE(in && out, DIFFSAL_E_ARG, "maxpool2d: null argument");
  DS_REQUIRE(N > 0 && H >= k && W >= k && C > 0 && C % 4 == 0 && k > 0 && stride > 0, DIFFSAL_E_SHAPE, "maxpool2d: bad shape");
  DS_REQUIRE(aligned16(in) && aligned16(out), DIFFSAL_E_ALIGN, "maxpool2d: misaligned pointer");
  const int Ho = (H - k) / stride + 1, Wo = (W - k) / stride + 1;
  const long total4 = static_cast<long>(N) * Ho * Wo * (C / 4);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(T)                                                                                                        \
  hipLaunchKernelGGL((maxpool2d_kernel<T>), dim3(ew_grid(total4)), dim3(256), 0, s, static_cast<const T*>(in),          \
                     static_cast<T*>(out), H, W, C, Ho, Wo, k, stride, total4)
  DS_DTYPE_DISPATCH(dtype, "maxpool2d", CALL);
#undef CALL
  return check_launch("maxpool2d");
}

extern "C" int diffsal_resize_update(const float* s_low, const float* x, const float* m_prev, float* x0_out, float* m_out,
                                     float* x_next, int N, int h, int w, int H, int W, float ex, float e0, float A, float c0,
                                     float c1, diffsal_stream_t stream) {
  DS_REQUIRE(s_low && x && m_out, DIFFSAL_E_ARG, "resize_update: null argument");
  DS_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0, DIFFSAL_E_SHAPE, "resize_update: bad shape");
  const long total = static_cast<long>(N) * H * W;
  hipLaunchKernelGGL(resize_update_kernel, dim3(ew_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), s_low, x,
                     m_prev, x0_out, m_out, x_next, h, w, H, W, static_cast<float>(h) / static_cast<float>(H),
                     static_cast<float>(w) / static_cast<float>(W), ex, e0, A, c0, c1, total);
  return check_launch("resize_update");
}
