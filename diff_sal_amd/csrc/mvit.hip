// Kernels of the MViTv2-S video encoder (R/models/mvit.py) that are not plain GEMMs / LayerNorms (those reuse
// diffsal_conv_igemm / diffsal_layernorm): patch-embedding im2col, the depthwise-Conv3d poolings of q / k / v fused with
// their LayerNorm, the max-pool of the skip path, the relative-position projections that feed diffsal_attention_general,
// and the token -> NCTHW output transpose.  Tokens are [B, 1 + T*H*W, C] with the class token in row 0 (mvit.py:1097-1099).
#include "common.h"

namespace diffsal {

// ------------------------------------------------------------------------------------------------
// PatchEmbed3D: Conv3d(3 -> 96, kernel (3,7,7), stride (2,4,4), padding (1,3,3)), mvit.py:983-989 / :157-163.
// cols[m][k] with m = (b, to, ho, wo), k = (c, kt, ky, kx) -- the order of weight.reshape(Cout, Cin*KT*KH*KW) --
// zero-padded to Kp columns (a multiple of 32); the projection itself is one GEMM.  x: [B, C, T, H, W].
// One wavefront per output pixel, lanes over k: reads along kx are contiguous runs of the input row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3d_kernel(const float* __restrict__ x, float* __restrict__ cols, int C, int T,
                                                       int H, int W, int To, int Ho, int Wo, int KT, int KH, int KW, int st,
                                                       int sh, int sw, int pt, int ph, int pw, int Kp, long M) {
  const int lane = threadIdx.x & 63;
  const long m = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int wo = static_cast<int>(m % Wo);
  long r = m / Wo;
  const int ho = static_cast<int>(r % Ho); r /= Ho;
  const int to = static_cast<int>(r % To);
  const int b = static_cast<int>(r / To);
  const int K = C * KT * KH * KW;
  const float* xb = x + static_cast<long>(b) * C * T * H * W;
  for (int k = lane; k < Kp; k += 64) {
    float v = 0.f;
    if (k < K) {
      const int kx = k % KW;
      int q = k / KW;
      const int ky = q % KH; q /= KH;
      const int kt = q % KT;
      const int c = q / KT;
      const int it = to * st - pt + kt, iy = ho * sh - ph + ky, ix = wo * sw - pw + kx;
      if (it >= 0 && it < T && iy >= 0 && iy < H && ix >= 0 && ix < W)
        v = xb[((static_cast<long>(c) * T + it) * H + iy) * W + ix];
    }
    cols[m * Kp + k] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// attention_pool (mvit.py:446-494) for one of q / k / v: depthwise Conv3d 3x3x3 (pad 1, stride (st, sh, sw), no bias,
// one filter per channel of the HEAD, shared by all heads) on the video tokens, class token passed through, then
// LayerNorm over the head dimension D on every token (class token included).
// in : element (b, n, head, d) at  in + b*in_sb + n*in_sn + head*D + d   (a slice of the fused qkv GEMM output)
// out: [B, heads, 1 + To*Ho*Wo, D]
// A group of G lanes owns one output token (float4 per lane); D <= 4*G.
// ------------------------------------------------------------------------------------------------
// a / b for 0 <= a < 2^23, 0 < b: one float multiply by the reciprocal (inv_b = 1.0f / b, computed once per kernel) and an
// exact integer fix-up, instead of the ~35-instruction integer division sequence.
__device__ __forceinline__ int div_fast(int a, int b, float inv_b) {
  int q = static_cast<int>((static_cast<float>(a) + 0.5f) * inv_b);
  const int r = a - q * b;
  q += (r >= b) - (r < 0);
  return q;
}

template <int G>
__global__ __launch_bounds__(256) void pool3d_ln_kernel(const float* __restrict__ in, const float* __restrict__ w27,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ out, long in_sb, long in_sn, int heads, int D,
                                                        int T, int H, int W, int To, int Ho, int Wo, int st, int sh, int sw,
                                                        float eps, long rows) {
  constexpr int ROWS = 256 / G;
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  const int Lo = To * Ho * Wo;
  const int c = gl * 4;
  const bool act = c < D;
  const float inv_row = 1.0f / static_cast<float>(Lo + 1), inv_heads = 1.0f / static_cast<float>(heads);
  const float inv_wo = 1.0f / static_cast<float>(Wo), inv_ho = 1.0f / static_cast<float>(Ho);
  for (long row = static_cast<long>(blockIdx.x) * ROWS + gr; row < rows; row += static_cast<long>(gridDim.x) * ROWS) {
    const int bh = div_fast(static_cast<int>(row), Lo + 1, inv_row);
    const int n = static_cast<int>(row) - bh * (Lo + 1);
    const int b = div_fast(bh, heads, inv_heads);
    const int head = bh - b * heads;
    const float* base = in + b * in_sb + static_cast<long>(head) * D + c;
    float4 acc = make_float4(0, 0, 0, 0);
    if (act) {
      if (n == 0) {
        acc = ld4(base);
      } else {
        const int l = n - 1;
        const int lw = div_fast(l, Wo, inv_wo);
        const int wo = l - lw * Wo;
        const int to = div_fast(lw, Ho, inv_ho);
        const int ho = lw - to * Ho;
        // (a branch-free form -- absent taps read the class token with weight 0, as in the data-gradient kernel -- was
        // measured slower here: 2.8 vs 2.35 ms per training step; nearly every tap of an output is present)
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          const int it = to * st - 1 + kt;
          if (it < 0 || it >= T) continue;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int iy = ho * sh - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const int ix = wo * sw - 1 + kx;
              if (ix < 0 || ix >= W) continue;
              const float4 a = ld4(base + (1 + (static_cast<long>(it) * H + iy) * W + ix) * in_sn);
              const float4 ww = ld4(w27 + ((kt * 3 + ky) * 3 + kx) * D + c);
              acc.x = fmaf(a.x, ww.x, acc.x); acc.y = fmaf(a.y, ww.y, acc.y);
              acc.z = fmaf(a.z, ww.z, acc.z); acc.w = fmaf(a.w, ww.w, acc.w);
            }
          }
        }
      }
    }
    if (gamma == nullptr) {          // training path: convolution only, LayerNorm is its own differentiable operator
      if (act) st4(out + row * D + c, acc);
      continue;
    }
    // LayerNorm over D (two-pass in registers, lane group butterfly)
    float s = act ? (acc.x + acc.y) + (acc.z + acc.w) : 0.f;
    s = group_sum<G>(s);
    const float mean = s / static_cast<float>(D);
    float qv = 0.f;
    if (act) {
      const float a0 = acc.x - mean, a1 = acc.y - mean, a2 = acc.z - mean, a3 = acc.w - mean;
      qv = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    qv = group_sum<G>(qv);
    const float rstd = 1.0f / sqrtf(qv / static_cast<float>(D) + eps);
    if (act) {
      const float4 g = ld4(gamma + c), be = ld4(beta + c);
      float4 o;
      o.x = (acc.x - mean) * rstd * g.x + be.x; o.y = (acc.y - mean) * rstd * g.y + be.y;
      o.z = (acc.z - mean) * rstd * g.z + be.z; o.w = (acc.w - mean) * rstd * g.w + be.w;
      st4(out + row * D + c, o);
    }
  }
}

// MaxPool3d of the skip path (mvit.py:765-777, 788-792): kernel (kt,kh,kw) = stride + 1 where stride > 1, padding k/2,
// on tokens [B, 1 + T*H*W, C]; class token passed through.  One float4 of channels per thread.
__global__ __launch_bounds__(256) void maxpool_tokens_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                             int T, int H, int W, int To, int Ho, int Wo, int kt, int kh,
                                                             int kw, int st, int sh, int sw, long total4) {
  const int c4n = C >> 2;
  const int Lo = To * Ho * Wo, Li = T * H * W;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total4; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    const long tok = i / c4n;
    const int n = static_cast<int>(tok % (Lo + 1));
    const int b = static_cast<int>(tok / (Lo + 1));
    const float* ib = in + static_cast<long>(b) * (Li + 1) * C + c;
    float4 m;
    if (n == 0) {
      m = ld4(ib);
    } else {
      const int l = n - 1;
      const int wo = l % Wo, ho = (l / Wo) % Ho, to = l / (Wo * Ho);
      m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
      for (int a = 0; a < kt; ++a) {
        const int it = to * st - kt / 2 + a;
        if (it < 0 || it >= T) continue;
        for (int e = 0; e < kh; ++e) {
          const int iy = ho * sh - kh / 2 + e;
          if (iy < 0 || iy >= H) continue;
          for (int f = 0; f < kw; ++f) {
            const int ix = wo * sw - kw / 2 + f;
            if (ix < 0 || ix >= W) continue;
            const float4 v = ld4(ib + (1 + (static_cast<long>(it) * H + iy) * W + ix) * C);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
          }
        }
      }
    }
    st4(out + tok * C + c, m);
  }
}

// ------------------------------------------------------------------------------------------------
// Relative-position projections of add_decomposed_rel_pos (mvit.py:363-410): for query (t, y, x) of head vector q,
//   extra[0 .. kt)        = q . Rt[t][j]        Rt: [qt][kt][D]
//   extra[8 .. 8 + kh)    = q . Rh[y][j]        Rh: [qh][kh][D]
//   extra[24 .. 24 + kw)  = q . Rw[x][j]        Rw: [qw][kw][D]
// (unused slots and the class-token row are zero).  q: [B*heads, 1 + qt*qh*qw, D] (the pooled, normalised, UNSCALED q).
// One wavefront per query: lane e < 48 forms one dot product.
// ------------------------------------------------------------------------------------------------
// Column layouts: E = 48: t [0, 8), h [8, 24), w [24, 48);  E = 32 (key grids up to 8 x 8 x 16 -- every MViTv2-S stage at
// 224 x 384): t [0, 8), h [8, 16), w [16, 32): a ninth less contraction length in the attention kernels' QK^T.
constexpr int REL_T0 = 0, REL_H0 = 8;
__host__ __device__ constexpr int rel_w0(int E) { return E == 32 ? 16 : 24; }

__global__ __launch_bounds__(256) void relpos_project_kernel(const float* __restrict__ q, const float* __restrict__ Rt,
                                                             const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                             float* __restrict__ extra, int D, int qt, int qh, int qw, int kt,
                                                             int kh, int kw, long rows, int REL_E) {
  extern __shared__ float sq[];   // [4][D]
  const int REL_W0 = rel_w0(REL_E);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = static_cast<long>(blockIdx.x) * 4 + wave;
  const int L = qt * qh * qw;
  const bool live = row < rows;
  const long rc = live ? row : rows - 1;
  const int n = static_cast<int>(rc % (L + 1));
  for (int d = lane; d < D; d += 64) sq[wave * D + d] = q[rc * D + d];
  __syncthreads();
  if (!live || lane >= REL_E) return;
  float r = 0.f;
  if (n > 0) {
    const int l = n - 1;
    const int x = l % qw, y = (l / qw) % qh, t = l / (qw * qh);
    const float* R = nullptr;
    if (lane < REL_H0) { if (lane < kt) R = Rt + (static_cast<long>(t) * kt + lane) * D; }
    else if (lane < REL_W0) { if (lane - REL_H0 < kh) R = Rh + (static_cast<long>(y) * kh + (lane - REL_H0)) * D; }
    else { if (lane - REL_W0 < kw) R = Rw + (static_cast<long>(x) * kw + (lane - REL_W0)) * D; }
    if (R) {
      const float* qs = sq + wave * D;
      for (int d = 0; d < D; d += 4) {
        const float4 a = ld4(R + d);
        r = fmaf(a.x, qs[d], fmaf(a.y, qs[d + 1], fmaf(a.z, qs[d + 2], fmaf(a.w, qs[d + 3], r))));
      }
    }
  }
  extra[row * REL_E + lane] = r;
}

// tokens [B, off + L, C] (rows off.. of each batch) -> [B, C, L]: the NCTHW feature maps MViT returns (mvit.py:1128-1134).
__global__ __launch_bounds__(256) void tokens_to_channels_first_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                       int C, int L, int off, int tiles_l) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y;
  const int tl = blockIdx.x % tiles_l, tc = blockIdx.x / tiles_l;
  const int l0 = tl * 64, c0 = tc * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = in + (static_cast<long>(b) * (off + L) + off) * C;
  for (int r = ty; r < 64; r += 4) {   // r: token within tile, tx: channel (contiguous reads)
    const int l = l0 + r, c = c0 + tx;
    tile[r][tx] = (l < L && c < C) ? src[static_cast<long>(l) * C + c] : 0.f;
  }
  __syncthreads();
  float* dst = out + static_cast<long>(b) * C * L;
  for (int r = ty; r < 64; r += 4) {   // r: channel within tile, tx: token (contiguous writes)
    const int c = c0 + r, l = l0 + tx;
    if (c < C && l < L) dst[static_cast<long>(c) * L + l] = tile[tx][r];
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of the depthwise pooling convolution above (conv-only form).  Gather form, no atomics:
//   data  : d_in[b, n, head, :] = sum over the taps whose output position exists of dy[out] * w[tap]   (class token: copy)
//   weight: per-workgroup partial sums part[chunk][27][D] (double), reduced in a fixed order by diffsal_reduce_partials.
// d_in is written through the same (batch, token) strides as the forward's input view, i.e. straight into the q / k / v
// slice of the fused-qkv gradient buffer.
// ------------------------------------------------------------------------------------------------
// NCS: spatial taps that can reach an input pixel per axis -- 3 (stride 1), 2 (stride 2), 1 (stride >= 3): the candidates are
// k = (i + 1) % s + j s, j < NCS, so a stride-8 key / value pooling does 3 loads per input token instead of 27 (24 of them
// dummies).  NCS = 0: every one of the 27 taps is tried (unequal spatial strides).  Valid taps are visited in the same order
// in both forms, so the results are bit-identical.
template <int G, int NCS>
__global__ __launch_bounds__(256) void pool3d_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w27,
                                                              float* __restrict__ din, long in_sb, long in_sn, int heads,
                                                              int D, int T, int H, int W, int To, int Ho, int Wo, int st,
                                                              int sh, int sw, long rows) {
  constexpr int ROWS = 256 / G;
  extern __shared__ __attribute__((aligned(16))) float w_s[];   // [27][D]: read per tap from LDS (in registers they cost 108 VGPRs)
  for (int i = threadIdx.x; i < 27 * D; i += 256) w_s[i] = w27[i];
  __syncthreads();
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  const int Li = T * H * W, Lo = To * Ho * Wo;
  const int c = gl * 4;
  if (c >= D) return;
  const float inv_row = 1.0f / static_cast<float>(Li + 1), inv_heads = 1.0f / static_cast<float>(heads);
  const float inv_w = 1.0f / static_cast<float>(W), inv_h = 1.0f / static_cast<float>(H);
  for (long row = static_cast<long>(blockIdx.x) * ROWS + gr; row < rows; row += static_cast<long>(gridDim.x) * ROWS) {
    const int bh = div_fast(static_cast<int>(row), Li + 1, inv_row);
    const int n = static_cast<int>(row) - bh * (Li + 1);
    const int b = div_fast(bh, heads, inv_heads);
    const int head = bh - b * heads;
    const float* dyb = dy + static_cast<long>(bh) * (Lo + 1) * D + c;
    float4 acc = make_float4(0, 0, 0, 0);
    if (n == 0) {
      acc = ld4(dyb);
    } else {
      const int l = n - 1;
      const int lw = div_fast(l, W, inv_w);
      const int ix = l - lw * W;
      const int it = div_fast(lw, H, inv_h);
      const int iy = lw - it * H;
      // branch-free: an absent tap reads the class-token row with weight 0 (27 independent loads in flight instead of
      // a branch and a drain per tap); one kernel plane (9 taps, 18 loads) at a time keeps 3-4 waves per SIMD resident
#pragma unroll 1
      for (int kt = 0; kt < 3; ++kt) {
        const int nt = it + 1 - kt;
        const bool vt = nt >= 0 && nt % st == 0 && nt / st < To;
        if constexpr (NCS == 0) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int ny = iy + 1 - ky;
            const bool vy = vt && ny >= 0 && ny % sh == 0 && ny / sh < Ho;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const int nx = ix + 1 - kx;
              const bool v = vy && nx >= 0 && nx % sw == 0 && nx / sw < Wo;
              const long o = v ? 1 + (static_cast<long>(nt / st) * Ho + ny / sh) * Wo + nx / sw : 0;
              const float4 g = ld4(dyb + o * D);
              float4 ww = ld4(w_s + ((kt * 3 + ky) * 3 + kx) * D + c);
              if (!v) ww = make_float4(0.f, 0.f, 0.f, 0.f);
              acc.x = fmaf(g.x, ww.x, acc.x); acc.y = fmaf(g.y, ww.y, acc.y);
              acc.z = fmaf(g.z, ww.z, acc.z); acc.w = fmaf(g.w, ww.w, acc.w);
            }
          }
        } else {
          const int k0y = (iy + 1) % sh, k0x = (ix + 1) % sw;
#pragma unroll
          for (int jy = 0; jy < NCS; ++jy) {
            const int ky = k0y + jy * sh, dy_ = iy + 1 - ky;
            const int ny = dy_ / sh;
            const bool vy = vt && ky <= 2 && dy_ >= 0 && ny < Ho;
#pragma unroll
            for (int jx = 0; jx < NCS; ++jx) {
              const int kx = k0x + jx * sw, dx_ = ix + 1 - kx;
              const int nx = dx_ / sw;
              const bool v = vy && kx <= 2 && dx_ >= 0 && nx < Wo;
              const long o = v ? 1 + (static_cast<long>(nt / st) * Ho + ny) * Wo + nx : 0;
              const float4 g = ld4(dyb + o * D);
              float4 ww = ld4(w_s + (v ? ((kt * 3 + ky) * 3 + kx) * D : 0) + c);
              if (!v) ww = make_float4(0.f, 0.f, 0.f, 0.f);
              acc.x = fmaf(g.x, ww.x, acc.x); acc.y = fmaf(g.y, ww.y, acc.y);
              acc.z = fmaf(g.z, ww.z, acc.z); acc.w = fmaf(g.w, ww.w, acc.w);
            }
          }
        }
      }
    }
    st4(din + b * in_sb + static_cast<long>(n) * in_sn + static_cast<long>(head) * D + c, acc);
  }
}

constexpr int POOL_WCHUNKS = 1024;   // 4 workgroups per CU: the gather is latency-bound

// one workgroup per chunk of output tokens; thread = (tap group, channel quad); LDS tree over the token lanes
__global__ __launch_bounds__(256) void pool3d_bwd_weight_kernel(const float* __restrict__ in, const float* __restrict__ dy,
                                                                double* __restrict__ part, long in_sb, long in_sn, int heads,
                                                                int D, int T, int H, int W, int To, int Ho, int Wo, int st,
                                                                int sh, int sw, long rows /* B*heads*Lo video tokens */) {
  extern __shared__ double shw[];   // [27][D]
  const int c4n = D >> 2;
  const int grp = threadIdx.x / c4n, c = (threadIdx.x % c4n) * 4;   // (token lane, kernel plane kt), channel quad
  const int TL = 256 / (3 * c4n);                                  // token lanes; each owns three threads per quad (kt = 0..2)
  const int tl = grp / 3, kt = grp - tl * 3;
  const bool live = tl < TL;
  const int Lo = To * Ho * Wo;
  const long lo = rows * blockIdx.x / POOL_WCHUNKS, hi = rows * (blockIdx.x + 1) / POOL_WCHUNKS;
  float4 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = make_float4(0, 0, 0, 0);
  if (live) {
    for (long r = lo + tl; r < hi; r += TL) {
      const int l = static_cast<int>(r % Lo);
      const long bh = r / Lo;
      const int head = static_cast<int>(bh % heads), b = static_cast<int>(bh / heads);
      const int wo = l % Wo, ho = (l / Wo) % Ho, to = l / (Wo * Ho);
      const float4 g = ld4(dy + (bh * (Lo + 1) + 1 + l) * D + c);
      const float* base = in + b * in_sb + static_cast<long>(head) * D + c;
      // branch-free: a tap outside the volume reads the class-token row and contributes g * 0
      const int it = to * st - 1 + kt;
      const bool vt = it >= 0 && it < T;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = ho * sh - 1 + ky;
        const bool vy = vt && iy >= 0 && iy < H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = wo * sw - 1 + kx;
          const bool v = vy && ix >= 0 && ix < W;
          const long o = v ? 1 + (static_cast<long>(it) * H + iy) * W + ix : 0;
          float4 a = ld4(base + o * in_sn);
          if (!v) a = make_float4(0.f, 0.f, 0.f, 0.f);
          float4& acc_t = acc[ky * 3 + kx];
          acc_t.x = fmaf(a.x, g.x, acc_t.x); acc_t.y = fmaf(a.y, g.y, acc_t.y);
          acc_t.z = fmaf(a.z, g.z, acc_t.z); acc_t.w = fmaf(a.w, g.w, acc_t.w);
        }
      }
    }
  }
  for (int i = threadIdx.x; i < 27 * D; i += 256) shw[i] = 0.0;
  __syncthreads();
  for (int turn = 0; turn < TL; ++turn) {          // fixed order over the token lanes: deterministic
    if (live && tl == turn) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        double* d = shw + (kt * 9 + i) * D + c;
        d[0] += acc[i].x; d[1] += acc[i].y; d[2] += acc[i].z; d[3] += acc[i].w;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 27 * D; i += 256) part[static_cast<long>(blockIdx.x) * 27 * D + i] = shw[i];
}

// MaxPool of the skip path with the arg-max recorded (training): idx[b, n, c] = input token index (1-based video token,
// 0 for the class token); ties keep the first window position in (t, y, x) order, as torch.nn.MaxPool3d does.
__global__ __launch_bounds__(256) void maxpool_tokens_idx_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 int* __restrict__ idx, int C, int T, int H, int W, int To,
                                                                 int Ho, int Wo, int kt, int kh, int kw, int st, int sh, int sw,
                                                                 long total) {
  const int Lo = To * Ho * Wo, Li = T * H * W;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % C);
    const long tok = i / C;
    const int n = static_cast<int>(tok % (Lo + 1));
    const int b = static_cast<int>(tok / (Lo + 1));
    const float* ib = in + static_cast<long>(b) * (Li + 1) * C + c;
    float m;
    int am = 0;
    if (n == 0) {
      m = ib[0];
    } else {
      const int l = n - 1;
      const int wo = l % Wo, ho = (l / Wo) % Ho, to = l / (Wo * Ho);
      m = -3.0e38f;
      for (int a = 0; a < kt; ++a) {
        const int it = to * st - kt / 2 + a;
        if (it < 0 || it >= T) continue;
        for (int e = 0; e < kh; ++e) {
          const int iy = ho * sh - kh / 2 + e;
          if (iy < 0 || iy >= H) continue;
          for (int f = 0; f < kw; ++f) {
            const int ix = wo * sw - kw / 2 + f;
            if (ix < 0 || ix >= W) continue;
            const int tokin = 1 + (it * H + iy) * W + ix;
            const float v = ib[static_cast<long>(tokin) * C];
            if (v > m) { m = v; am = tokin; }
          }
        }
      }
    }
    out[i] = m;
    idx[i] = am;
  }
}

// d_in[b, tok, c] = sum of dy over the (at most kt*kh*kw / stride) outputs whose recorded arg-max is this token.
// Thread = (token, channel quad): the token's coordinates and its candidate windows are worked out once per 16 bytes.
__global__ __launch_bounds__(256) void maxpool_tokens_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                                 float* __restrict__ din, int C, int T, int H, int W, int To,
                                                                 int Ho, int Wo, int kt, int kh, int kw, int st, int sh, int sw,
                                                                 long total4) {
  const int Lo = To * Ho * Wo, Li = T * H * W;
  const int c4n = C >> 2;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total4; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    const long tok = i / c4n;
    const int n = static_cast<int>(tok % (Li + 1));
    const int b = static_cast<int>(tok / (Li + 1));
    const long ob = static_cast<long>(b) * (Lo + 1) * C + c;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n == 0) {
      g = ld4(dy + ob);
    } else {
      const int l = n - 1;
      const int ix = l % W, r = l / W, iy = r % H, it = r / H;
      for (int a = 0; a < kt; ++a) {          // outputs whose window position a covers this input: to*st - kt/2 + a == it
        const int nt = it + kt / 2 - a;
        if (nt < 0 || nt % st != 0 || nt / st >= To) continue;
        for (int e = 0; e < kh; ++e) {
          const int ny = iy + kh / 2 - e;
          if (ny < 0 || ny % sh != 0 || ny / sh >= Ho) continue;
          for (int f = 0; f < kw; ++f) {
            const int nx = ix + kw / 2 - f;
            if (nx < 0 || nx % sw != 0 || nx / sw >= Wo) continue;
            const long o = ob + (1 + (static_cast<long>(nt / st) * Ho + ny / sh) * Wo + nx / sw) * C;
            const int4 am = *reinterpret_cast<const int4*>(idx + o);
            const float4 d4 = ld4(dy + o);
            g.x += am.x == n ? d4.x : 0.f; g.y += am.y == n ? d4.y : 0.f;
            g.z += am.z == n ? d4.z : 0.f; g.w += am.w == n ? d4.w : 0.f;
          }
        }
      }
    }
    st4(din + tok * C + c, g);
  }
}

// Backward of relpos_project, query side: dq[row][d] (+)= sum_e dE[row][e] * R_e[d]  (class-token rows: nothing).
// One wavefront per query, lane = channel (D <= 128: two channels per lane at most).
__global__ __launch_bounds__(256) void relpos_bwd_q_kernel(const float* __restrict__ dE, const float* __restrict__ Rt,
                                                           const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                           float* __restrict__ dq, int D, int qt, int qh, int qw, int kt, int kh,
                                                           int kw, int accumulate, long rows, int REL_E) {
  const int REL_W0 = rel_w0(REL_E);
  const int lane = threadIdx.x & 63;
  const long row = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int L = qt * qh * qw;
  const int n = static_cast<int>(row % (L + 1));
  for (int d = lane; d < D; d += 64) {
    float r = 0.f;
    if (n > 0) {
      const int l = n - 1;
      const int x = l % qw, y = (l / qw) % qh, t = l / (qw * qh);
      const float* e = dE + row * REL_E;
      for (int j = 0; j < kt; ++j) r = fmaf(e[REL_T0 + j], Rt[(static_cast<long>(t) * kt + j) * D + d], r);
      for (int j = 0; j < kh; ++j) r = fmaf(e[REL_H0 + j], Rh[(static_cast<long>(y) * kh + j) * D + d], r);
      for (int j = 0; j < kw; ++j) r = fmaf(e[REL_W0 + j], Rw[(static_cast<long>(x) * kw + j) * D + d], r);
    }
    if (accumulate) dq[row * D + d] += r; else dq[row * D + d] = r;
  }
}

// Backward of relpos_project, table side: for axis a in {t, y, x}: dR[i][j][d] = sum over the queries whose coordinate on
// that axis is i of dE[query][slot0 + j] * q[query][d].  One workgroup per (axis coordinate i, chunk of the queries);
// thread = (j, channel quad); per-chunk partial sums part[chunk][table offset] (double) are combined in a fixed order by
// diffsal_reduce_partials: deterministic, no atomics.  Table offsets: Rt at 0, Rh after qt*kt*D, Rw after that.
constexpr int REL_CHUNKS = 32;

__global__ __launch_bounds__(256) void relpos_bwd_tables_kernel(const float* __restrict__ dE, const float* __restrict__ q,
                                                                double* __restrict__ part, int BH, int D, int qt, int qh,
                                                                int qw, int kt, int kh, int kw, int REL_E) {
  const int REL_W0 = rel_w0(REL_E);
  int i = blockIdx.x, axis = 0;
  if (i >= qt) { i -= qt; axis = 1; }
  if (axis == 1 && i >= qh) { i -= qh; axis = 2; }
  const int chunk = blockIdx.y;
  const int kk = axis == 0 ? kt : (axis == 1 ? kh : kw);
  const int slot0 = axis == 0 ? REL_T0 : (axis == 1 ? REL_H0 : REL_W0);
  const long width = (static_cast<long>(qt) * kt + static_cast<long>(qh) * kh + static_cast<long>(qw) * kw) * D;
  const long tab0 = axis == 0 ? 0 : (axis == 1 ? static_cast<long>(qt) * kt * D
                                               : (static_cast<long>(qt) * kt + static_cast<long>(qh) * kh) * D);
  const int c4n = D >> 2;
  const int L = qt * qh * qw;
  const int n_a = axis == 0 ? qh * qw : (axis == 1 ? qt * qw : qt * qh);       // queries per image with this coordinate
  const long n_all = static_cast<long>(BH) * n_a;
  const long lo = n_all * chunk / REL_CHUNKS, hi = n_all * (chunk + 1) / REL_CHUNKS;
  // a thread owns a channel quad and FOUR table columns j: the query quad is loaded once for the four products (the gather is
  // bound by L1 / L2 reads of q, which the one-column form re-read kk times per axis).  There are only kk4 * D/4 such items
  // (48-144 at D = 96), so the 256 threads form G = 256 / items groups that each walk 1/G of the chunk's queries -- the walk
  // is load latency -- and the groups' sums are added in group order through LDS (fixed order: deterministic).
  __shared__ float4 sh[256][4];
  const int kk4 = (kk + 3) >> 2;
  const int n_items = kk4 * c4n;
  const int per_pass = n_items < 256 ? n_items : 256;
  const int G = 256 / per_pass;
  const int sub = threadIdx.x / per_pass, it_local = threadIdx.x - sub * per_pass;
  const int nb = axis == 2 ? qh : qw;
  for (int item0 = 0; item0 < n_items; item0 += per_pass) {
    const int item = item0 + it_local;
    const bool live = sub < G && item < n_items;
    const int j0 = live ? (item / c4n) * 4 : 0, c = live ? (item % c4n) * 4 : 0;
    float4 acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[jj] = make_float4(0, 0, 0, 0);
    if (live) {
      // the query walk: (image, two free coordinates) advance as counters, no divisions in the loop.  a = slow free
      // coordinate, b2 = fast one (axis 0: (y, x); axis 1: (t, x); axis 2: (t, y))
      const long lo_s = lo + (hi - lo) * sub / G, hi_s = lo + (hi - lo) * (sub + 1) / G;
      int bh = static_cast<int>(lo_s / n_a);
      int m0 = static_cast<int>(lo_s - static_cast<long>(bh) * n_a);
      int a = m0 / nb, b2 = m0 - a * nb;
#pragma unroll 4
      for (long u = lo_s; u < hi_s; ++u) {
        const int t = axis == 0 ? i : a;
        const int y = axis == 0 ? a : (axis == 1 ? i : b2);
        const int x = axis == 2 ? i : b2;
        const long row = static_cast<long>(bh) * (L + 1) + 1 + (static_cast<long>(t) * qh + y) * qw + x;
        if (++b2 == nb) {
          b2 = 0;
          if (++a * nb == n_a) { a = 0; ++bh; }
        }
        const float4 qv = ld4(q + row * D + c);
        const float* er = dE + row * REL_E + slot0 + j0;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float e = er[j0 + jj < kk ? jj : 0];
          acc[jj].x = fmaf(e, qv.x, acc[jj].x); acc[jj].y = fmaf(e, qv.y, acc[jj].y);
          acc[jj].z = fmaf(e, qv.z, acc[jj].z); acc[jj].w = fmaf(e, qv.w, acc[jj].w);
        }
      }
    }
    if (G > 1) {
      __syncthreads();               // the previous pass's readers are done
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) sh[threadIdx.x][jj] = acc[jj];
      __syncthreads();
    }
    if (live && sub == 0) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if (j0 + jj >= kk) continue;
        double v[4] = {acc[jj].x, acc[jj].y, acc[jj].z, acc[jj].w};
        for (int g2 = 1; g2 < G; ++g2) {
          const float4 o4 = sh[g2 * per_pass + it_local][jj];
          v[0] += o4.x; v[1] += o4.y; v[2] += o4.z; v[3] += o4.w;
        }
        double* o = part + chunk * width + tab0 + (static_cast<long>(i) * kk + j0 + jj) * D + c;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
      }
    }
  }
}

// mvit_pool.hip: head dimension 96 (8 lanes per query, 12 channels per lane)
int relpos_project96(const float* q, const float* Rt, const float* Rh, const float* Rw, float* extra, long rows, int qt, int qh,
                     int qw, int kt, int kh, int kw, int E, hipStream_t s);
int relpos_bwd_q96(const float* dE, const float* Rt, const float* Rh, const float* Rw, float* dq, long rows, int qt, int qh,
                   int qw, int kt, int kh, int kw, int accumulate, int E, hipStream_t s);

static int rows_grid(long rows, int per_block) {
  long g = (rows + per_block - 1) / per_block;
  return static_cast<int>(g > 65535 * 16 ? 65535 * 16 : g);
}

}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_im2col3d(const float* x, float* cols, int B, int C, int T, int H, int W, int KT, int KH, int KW, int st,
                                int sh, int sw, int pt, int ph, int pw, int Kp, diffsal_stream_t stream) {
  DS_REQUIRE(x && cols, DIFFSAL_E_ARG, "im2col3d: null argument");
  DS_REQUIRE(B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && KT > 0 && KH > 0 && KW > 0 && st > 0 && sh > 0 && sw > 0 &&
                 Kp >= C * KT * KH * KW && Kp % 32 == 0,
             DIFFSAL_E_SHAPE, "im2col3d: bad shape (Kp=%d must cover C*KT*KH*KW and be a multiple of 32)", Kp);
  const int To = (T + 2 * pt - KT) / st + 1, Ho = (H + 2 * ph - KH) / sh + 1, Wo = (W + 2 * pw - KW) / sw + 1;
  DS_REQUIRE(To > 0 && Ho > 0 && Wo > 0, DIFFSAL_E_SHAPE, "im2col3d: empty output");
  const long M = static_cast<long>(B) * To * Ho * Wo;
  DS_REQUIRE((M + 3) / 4 < (1L << 31), DIFFSAL_E_SHAPE, "im2col3d: too many rows");
  hipLaunchKernelGGL(im2col3d_kernel, dim3(static_cast<unsigned>((M + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, cols, C, T, H, W, To, Ho, Wo, KT, KH, KW, st, sh, sw, pt, ph, pw, Kp, M);
  return check_launch("im2col3d");
}

extern "C" int diffsal_pool3d_ln(const float* in, const float* w27, const float* gamma, const float* beta, float* out, int B,
                                 int heads, int D, int T, int H, int W, int st, int sh, int sw, long in_stride_b,
                                 long in_stride_n, float eps, diffsal_stream_t stream) {
  DS_REQUIRE(in && w27 && out && (gamma == nullptr) == (beta == nullptr), DIFFSAL_E_ARG, "pool3d_ln: null argument");
  DS_REQUIRE(B > 0 && heads > 0 && D > 0 && D % 4 == 0 && D <= 256 && T > 0 && H > 0 && W > 0 && st > 0 && sh > 0 && sw > 0,
             DIFFSAL_E_SHAPE, "pool3d_ln: bad shape D=%d", D);
  DS_REQUIRE(aligned16(in) && aligned16(out) && aligned16(w27) && (!gamma || (aligned16(gamma) && aligned16(beta))) &&
                 in_stride_b % 4 == 0 && in_stride_n % 4 == 0,
             DIFFSAL_E_ALIGN, "pool3d_ln: misaligned pointer / stride");
  const int To = (T - 1) / st + 1, Ho = (H - 1) / sh + 1, Wo = (W - 1) / sw + 1;   // (X + 2 - 3) / s + 1
  const long rows = static_cast<long>(B) * heads * (static_cast<long>(To) * Ho * Wo + 1);
  DS_REQUIRE(rows < (1L << 23), DIFFSAL_E_SHAPE, "pool3d_ln: %ld output rows (the index arithmetic covers < 2^23)", rows);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(G)                                                                                                          \
  hipLaunchKernelGGL((pool3d_ln_kernel<G>), dim3(rows_grid(rows, 256 / G)), dim3(256), 0, s, in, w27, gamma, beta, out,    \
                     in_stride_b, in_stride_n, heads, D, T, H, W, To, Ho, Wo, st, sh, sw, eps, rows)
  if (D <= 32) { CALL(8); } else if (D <= 64) { CALL(16); } else if (D <= 128) { CALL(32); } else { CALL(64); }
#undef CALL
  return check_launch("pool3d_ln");
}

extern "C" int diffsal_maxpool_tokens(const float* in, float* out, int B, int C, int T, int H, int W, int kt, int kh, int kw,
                                      int st, int sh, int sw, diffsal_stream_t stream) {
  DS_REQUIRE(in && out, DIFFSAL_E_ARG, "maxpool_tokens: null argument");
  DS_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && T > 0 && H > 0 && W > 0 && kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0,
             DIFFSAL_E_SHAPE, "maxpool_tokens: bad shape");
  DS_REQUIRE(aligned16(in) && aligned16(out), DIFFSAL_E_ALIGN, "maxpool_tokens: misaligned pointer");
  const int To = (T + 2 * (kt / 2) - kt) / st + 1, Ho = (H + 2 * (kh / 2) - kh) / sh + 1, Wo = (W + 2 * (kw / 2) - kw) / sw + 1;
  const long total4 = static_cast<long>(B) * (static_cast<long>(To) * Ho * Wo + 1) * (C / 4);
  long g = (total4 + 255) / 256;
  g = g > 16384 ? 16384 : g;
  hipLaunchKernelGGL(maxpool_tokens_kernel, dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out,
                     C, T, H, W, To, Ho, Wo, kt, kh, kw, st, sh, sw, total4);
  return check_launch("maxpool_tokens");
}

extern "C" int diffsal_relpos_project(const float* q, const float* Rt, const float* Rh, const float* Rw, float* extra, int BH,
                                      int D, int qt, int qh, int qw, int kt, int kh, int kw, int E, diffsal_stream_t stream) {
  DS_REQUIRE(q && Rt && Rh && Rw && extra, DIFFSAL_E_ARG, "relpos_project: null argument");
  DS_REQUIRE(E == 48 || E == 32, DIFFSAL_E_SHAPE, "relpos_project: E=%d (column layouts: 48 or 32)", E);
  const int REL_E = E, REL_W0 = rel_w0(E);
  DS_REQUIRE(BH > 0 && D > 0 && D % 4 == 0 && D <= 1024 && qt > 0 && qh > 0 && qw > 0 && kt > 0 && kt <= REL_H0 - REL_T0 &&
                 kh > 0 && kh <= REL_W0 - REL_H0 && kw > 0 && kw <= REL_E - REL_W0,
             DIFFSAL_E_SHAPE, "relpos_project: key grid %dx%dx%d exceeds the slots of the E=%d layout", kt, kh, kw, E);
  DS_REQUIRE(aligned16(Rt) && aligned16(Rh) && aligned16(Rw), DIFFSAL_E_ALIGN, "relpos_project: misaligned table");
  const long rows = static_cast<long>(BH) * (static_cast<long>(qt) * qh * qw + 1);
  DS_REQUIRE((rows + 3) / 4 < (1L << 31), DIFFSAL_E_SHAPE, "relpos_project: too many rows");
  if (D == 96 && rows < (1L << 31) && aligned16(q) && aligned16(extra))
    return relpos_project96(q, Rt, Rh, Rw, extra, rows, qt, qh, qw, kt, kh, kw, E, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(relpos_project_kernel, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 4 * D * sizeof(float),
                     static_cast<hipStream_t>(stream), q, Rt, Rh, Rw, extra, D, qt, qh, qw, kt, kh, kw, rows, E);
  return check_launch("relpos_project");
}

extern "C" int diffsal_tokens_to_channels_first(const float* in, float* out, int B, int C, int L, int off,
                                                diffsal_stream_t stream) {
  DS_REQUIRE(in && out, DIFFSAL_E_ARG, "tokens_to_channels_first: null argument");
  DS_REQUIRE(B > 0 && C > 0 && L > 0 && off >= 0, DIFFSAL_E_SHAPE, "tokens_to_channels_first: bad shape");
  const int tiles_l = (L + 63) / 64, tiles_c = (C + 63) / 64;
  hipLaunchKernelGGL(tokens_to_channels_first_kernel, dim3(tiles_l * tiles_c, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, C, L, off, tiles_l);
  return check_launch("tokens_to_channels_first");
}

extern "C" int diffsal_pool3d_bwd_data(const float* dy, const float* w27, float* din, int B, int heads, int D, int T, int H,
                                       int W, int st, int sh, int sw, long in_stride_b, long in_stride_n,
                                       diffsal_stream_t stream) {
  DS_REQUIRE(dy && w27 && din, DIFFSAL_E_ARG, "pool3d_bwd_data: null argument");
  DS_REQUIRE(B > 0 && heads > 0 && D > 0 && D % 4 == 0 && D <= 256 && T > 0 && H > 0 && W > 0 && st > 0 && sh > 0 && sw > 0,
             DIFFSAL_E_SHAPE, "pool3d_bwd_data: bad shape");
  DS_REQUIRE(aligned16(dy) && aligned16(w27) && aligned16(din) && in_stride_b % 4 == 0 && in_stride_n % 4 == 0, DIFFSAL_E_ALIGN,
             "pool3d_bwd_data: misaligned pointer / stride");
  const int To = (T - 1) / st + 1, Ho = (H - 1) / sh + 1, Wo = (W - 1) / sw + 1;
  const long rows = static_cast<long>(B) * heads * (static_cast<long>(T) * H * W + 1);
  DS_REQUIRE(rows < (1L << 23), DIFFSAL_E_SHAPE, "pool3d_bwd_data: %ld input rows (the index arithmetic covers < 2^23)", rows);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL2(G, NCS)                                                                                                  \
  hipLaunchKernelGGL((pool3d_bwd_data_kernel<G, NCS>), dim3(rows_grid(rows, 256 / G)), dim3(256), 27 * D * sizeof(float), s, dy, \
                     w27, din, in_stride_b, in_stride_n, heads, D, T, H, W, To, Ho, Wo, st, sh, sw, rows)
#define CALL(G)                                                                                                        \
  do {                                                                                                                 \
    if (sh != sw) CALL2(G, 0);                                                                                         \
    else if (sh == 1) CALL2(G, 3);                                                                                     \
    else if (sh == 2) CALL2(G, 2);                                                                                     \
    else CALL2(G, 1);                                                                                                  \
  } while (0)
  if (D <= 32) { CALL(8); } else if (D <= 64) { CALL(16); } else if (D <= 128) { CALL(32); } else { CALL(64); }
#undef CALL
#undef CALL2
  return check_launch("pool3d_bwd_data");
}

extern "C" int diffsal_pool3d_bwd_weight_chunks(void) { return POOL_WCHUNKS; }

extern "C" int diffsal_pool3d_bwd_weight(const float* in, const float* dy, double* part, int B, int heads, int D, int T, int H,
                                         int W, int st, int sh, int sw, long in_stride_b, long in_stride_n,
                                         diffsal_stream_t stream) {
  DS_REQUIRE(in && dy && part, DIFFSAL_E_ARG, "pool3d_bwd_weight: null argument");
  DS_REQUIRE(B > 0 && heads > 0 && D > 0 && D % 4 == 0 && D <= 256 && T > 0 && H > 0 && W > 0, DIFFSAL_E_SHAPE,
             "pool3d_bwd_weight: bad shape");
  DS_REQUIRE(27 * D * sizeof(double) <= 64 * 1024, DIFFSAL_E_SHAPE, "pool3d_bwd_weight: D=%d too wide", D);
  const int To = (T - 1) / st + 1, Ho = (H - 1) / sh + 1, Wo = (W - 1) / sw + 1;
  const long rows = static_cast<long>(B) * heads * To * Ho * Wo;
  hipLaunchKernelGGL(pool3d_bwd_weight_kernel, dim3(POOL_WCHUNKS), dim3(256), 27 * D * sizeof(double),
                     static_cast<hipStream_t>(stream), in, dy, part, in_stride_b, in_stride_n, heads, D, T, H, W, To, Ho, Wo,
                     st, sh, sw, rows);
  return check_launch("pool3d_bwd_weight");
}

extern "C" int diffsal_maxpool_tokens_idx(const float* in, float* out, int* idx, int B, int C, int T, int H, int W, int kt,
                                          int kh, int kw, int st, int sh, int sw, diffsal_stream_t stream) {
  DS_REQUIRE(in && out && idx, DIFFSAL_E_ARG, "maxpool_tokens_idx: null argument");
  DS_REQUIRE(B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0,
             DIFFSAL_E_SHAPE, "maxpool_tokens_idx: bad shape");
  const int To = (T + 2 * (kt / 2) - kt) / st + 1, Ho = (H + 2 * (kh / 2) - kh) / sh + 1, Wo = (W + 2 * (kw / 2) - kw) / sw + 1;
  const long total = static_cast<long>(B) * (static_cast<long>(To) * Ho * Wo + 1) * C;
  long g = (total + 255) / 256;
  g = g > 32768 ? 32768 : g;
  hipLaunchKernelGGL(maxpool_tokens_idx_kernel, dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     out, idx, C, T, H, W, To, Ho, Wo, kt, kh, kw, st, sh, sw, total);
  return check_launch("maxpool_tokens_idx");
}

extern "C" int diffsal_maxpool_tokens_bwd(const float* dy, const int* idx, float* din, int B, int C, int T, int H, int W, int kt,
                                          int kh, int kw, int st, int sh, int sw, diffsal_stream_t stream) {
  DS_REQUIRE(dy && idx && din, DIFFSAL_E_ARG, "maxpool_tokens_bwd: null argument");
  DS_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && T > 0 && H > 0 && W > 0 && kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0,
             DIFFSAL_E_SHAPE, "maxpool_tokens_bwd: bad shape (C must be a multiple of 4)");
  DS_REQUIRE(aligned16(dy) && aligned16(idx) && aligned16(din), DIFFSAL_E_ALIGN, "maxpool_tokens_bwd: misaligned pointer");
  const int To = (T + 2 * (kt / 2) - kt) / st + 1, Ho = (H + 2 * (kh / 2) - kh) / sh + 1, Wo = (W + 2 * (kw / 2) - kw) / sw + 1;
  const long total4 = static_cast<long>(B) * (static_cast<long>(T) * H * W + 1) * (C / 4);
  long g = (total4 + 255) / 256;
  g = g > 32768 ? 32768 : g;
  hipLaunchKernelGGL(maxpool_tokens_bwd_kernel, dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), dy,
                     idx, din, C, T, H, W, To, Ho, Wo, kt, kh, kw, st, sh, sw, total4);
  return check_launch("maxpool_tokens_bwd");
}

extern "C" int diffsal_relpos_project_bwd_chunks(void) { return REL_CHUNKS; }

extern "C" int diffsal_relpos_project_bwd(const float* dextra, const float* q, const float* Rt, const float* Rh, const float* Rw,
                                          float* dq, int accumulate, double* part, int BH, int D, int qt, int qh, int qw, int kt,
                                          int kh, int kw, int E, diffsal_stream_t stream) {
  DS_REQUIRE(dextra && q && Rt && Rh && Rw && dq && part, DIFFSAL_E_ARG, "relpos_project_bwd: null argument");
  DS_REQUIRE(E == 48 || E == 32, DIFFSAL_E_SHAPE, "relpos_project_bwd: E=%d (column layouts: 48 or 32)", E);
  const int REL_E = E, REL_W0 = rel_w0(E);
  DS_REQUIRE(BH > 0 && D > 0 && D % 4 == 0 && D <= 1024 && qt > 0 && qh > 0 && qw > 0 && kt > 0 && kt <= REL_H0 - REL_T0 &&
                 kh > 0 && kh <= REL_W0 - REL_H0 && kw > 0 && kw <= REL_E - REL_W0,
             DIFFSAL_E_SHAPE, "relpos_project_bwd: bad shape");
  const long rows = static_cast<long>(BH) * (static_cast<long>(qt) * qh * qw + 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if (D == 96 && rows < (1L << 31) && aligned16(dextra) && aligned16(dq) && aligned16(Rt) && aligned16(Rh) && aligned16(Rw)) {
    rc = relpos_bwd_q96(dextra, Rt, Rh, Rw, dq, rows, qt, qh, qw, kt, kh, kw, accumulate, E, s);
  } else {
    hipLaunchKernelGGL(relpos_bwd_q_kernel, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, s, dextra, Rt, Rh, Rw, dq,
                       D, qt, qh, qw, kt, kh, kw, accumulate, rows, E);
    rc = check_launch("relpos_project_bwd(q)");
  }
  if (rc) return rc;
  hipLaunchKernelGGL(relpos_bwd_tables_kernel, dim3(qt + qh + qw, REL_CHUNKS), dim3(256), 0, s, dextra, q, part, BH, D, qt, qh,
                     qw, kt, kh, kw, E);
  return check_launch("relpos_project_bwd(tables)");
}
