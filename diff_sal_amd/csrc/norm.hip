// Normalisation-family kernels (HBM-bound): GroupNorm+swish on NHWC images, LayerNorm over C on
// tokens, and the depthwise q/k/v projections fused with their LayerNorm.
//
// Layout rule: channels are the fastest dimension everywhere, so a group of G = pow2 lanes owns one
// token/pixel and reads it with coalesced float4 loads; reductions over C are xor-butterflies inside
// that lane group (64-wide wavefronts hold 64/G tokens at once).
#include "common.h"

namespace diffsal {

// ------------------------------------------------------------------------------------------------
// GroupNorm(32, eps) + swish.   R/models/saliency_decoder/sal_unet.py:36-44
// pass 1: per (image, pixel-chunk) partial sums per group  -> ws[B][chunks][groups][2] (double)
// pass 2: finalise mean/rstd per (image, group), normalise, affine, swish.
// ------------------------------------------------------------------------------------------------
constexpr int GN_CHUNKS_MAX = 128;   // workspace is sized for this many pixel chunks per image
static int gn_chunks() { const int t = tune(TUNE_GN_CHUNKS); return t > 0 && t <= GN_CHUNKS_MAX ? t : 32; }

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* __restrict__ ws,
                                                       int HW, int C, int groups) {
  extern __shared__ double sh[];  // [C][2]
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int GN_CHUNKS = gridDim.x;
  const int c4n = C >> 2;                     // float4 per pixel
  const int pix_per_pass = 256 / c4n > 0 ? 256 / c4n : 1;
  const int my_c4 = threadIdx.x % c4n;
  const int my_p = threadIdx.x / c4n;
  const bool active = threadIdx.x < pix_per_pass * c4n;
  const int p_begin = static_cast<long>(HW) * chunk / GN_CHUNKS;
  const int p_end = static_cast<long>(HW) * (chunk + 1) / GN_CHUNKS;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (active) {
    const T* base = x + (static_cast<long>(b) * HW) * C + my_c4 * 4;
    constexpr int U = 8;   // independent loads in flight (the serial form was one memory round trip per pixel pass)
    for (int p0 = p_begin + my_p; p0 < p_end; p0 += pix_per_pass * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = p0 + u * pix_per_pass;
        v[u] = p < p_end ? ld4(base + static_cast<long>(p) * C) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s[0] += v[u].x; q[0] += v[u].x * v[u].x;
        s[1] += v[u].y; q[1] += v[u].y * v[u].y;
        s[2] += v[u].z; q[2] += v[u].z * v[u].z;
        s[3] += v[u].w; q[3] += v[u].w * v[u].w;
      }
    }
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) sh[i] = 0.0;
  __syncthreads();
  if (active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      atomicAdd(&sh[(my_c4 * 4 + i) * 2 + 0], static_cast<double>(s[i]));
      atomicAdd(&sh[(my_c4 * 4 + i) * 2 + 1], static_cast<double>(q[i]));
    }
  }
  __syncthreads();
  const int cpg = C / groups;
  for (int g = threadIdx.x; g < groups; g += 256) {
    double a = 0, bq = 0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += sh[c * 2]; bq += sh[c * 2 + 1]; }
    double* o = ws + ((static_cast<long>(b) * GN_CHUNKS + chunk) * groups + g) * 2;
    o[0] = a; o[1] = bq;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const double* __restrict__ ws,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ out, int HW, int C, int groups, float eps, int swish,
                                                       int GN_CHUNKS) {
  extern __shared__ float shf[];  // [C] scale, [C] shift, [groups] mean, [groups] rstd
  float* g_mean = shf + 2 * C;
  float* g_rstd = g_mean + groups;
  const int b = blockIdx.y;
  const int cpg = C / groups;
  // statistics of the image's groups, ONCE per group (not per channel: at C = 768 and 32 chunks that was 192 dependent
  // 8-byte loads per thread, 20 us of every workgroup's life -- more than the normalisation itself): 8 lanes per group take
  // every 8th chunk, their sums are combined in a fixed order, lane 0 finishes mean / rstd in fp64
  for (int g0 = 0; g0 < groups; g0 += 32) {
    const int g = g0 + (threadIdx.x >> 3), kl = threadIdx.x & 7;
    double s = 0, q = 0;
    if (g < groups)
      for (int k = kl; k < GN_CHUNKS; k += 8) {
        const double* o = ws + ((static_cast<long>(b) * GN_CHUNKS + k) * groups + g) * 2;
        s += o[0]; q += o[1];
      }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      s += __shfl_xor(s, off, 8);
      q += __shfl_xor(q, off, 8);
    }
    if (kl == 0 && g < groups) {
      const double n = static_cast<double>(HW) * cpg;
      const double mean = s / n;
      double var = q / n - mean * mean;
      var = var < 0 ? 0 : var;
      g_mean[g] = static_cast<float>(mean);
      g_rstd[g] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float sc = g_rstd[g] * gamma[c];
    shf[c] = sc;
    shf[C + c] = beta[c] - g_mean[g] * sc;
  }
  __syncthreads();
  const int c4n = C >> 2;
  const long total4 = static_cast<long>(HW) * c4n;
  const T* xb = x + static_cast<long>(b) * HW * C;
  T* ob = out + static_cast<long>(b) * HW * C;
  // channel quad of element i = i mod c4n, carried along instead of a 64-bit modulo per element
  const long i0 = static_cast<long>(blockIdx.x) * 256 + threadIdx.x, step = static_cast<long>(gridDim.x) * 256;
  int cq = static_cast<int>(i0 % c4n);
  const int cstep = static_cast<int>(step % c4n);
  for (long i = i0; i < total4; i += step) {
    const int c = cq * 4;
    cq += cstep;
    if (cq >= c4n) cq -= c4n;
    float4 v = ld4(xb + i * 4);
    v.x = v.x * shf[c + 0] + shf[C + c + 0];
    v.y = v.y * shf[c + 1] + shf[C + c + 1];
    v.z = v.z * shf[c + 2] + shf[C + c + 2];
    v.w = v.w * shf[c + 3] + shf[C + c + 3];
    if (swish) { v.x = swishf(v.x); v.y = swishf(v.y); v.z = swishf(v.z); v.w = swishf(v.w); }
    st4(ob + i * 4, v);
  }
}

// ------------------------------------------------------------------------------------------------
// Row helpers: a lane group of G lanes owns one token of C channels (C % 4 == 0), NV = ceil(C/4/G)
// float4 per lane.  Two-pass (mean, then centred variance) in registers.
// ------------------------------------------------------------------------------------------------
template <int G, int NV, typename T>
__device__ __forceinline__ void ln_rows_finish(float4 (&v)[NV], int gl, int C, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float eps, T* __restrict__ orow) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (gl + i * G) * 4;
    if (c < C) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  s = group_sum<G>(s);
  const float mean = s / static_cast<float>(C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (gl + i * G) * 4;
    if (c < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  q = group_sum<G>(q);
  const float rstd = 1.0f / sqrtf(q / static_cast<float>(C) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (gl + i * G) * 4;
    if (c < C) {
      const float4 g = ld4(gamma + c), b = ld4(beta + c);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      st4(orow + c, o);
    }
  }
}

// The same LayerNorm applied in place to a token held in registers (the block's `norm` folded into the kernels that read
// its output): identical arithmetic to ln_rows_finish, and the value is rounded through the storage type T exactly as the
// stand-alone LayerNorm's store + reload would, so the fused and the two-kernel forms agree bit for bit.
template <int G, int NV, typename T>
__device__ __forceinline__ void ln_rows_inplace(float4 (&v)[NV], int gl, int C, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (gl + i * G) * 4;
    if (c < C) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  s = group_sum<G>(s);
  const float mean = s / static_cast<float>(C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (gl + i * G) * 4;
    if (c < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  q = group_sum<G>(q);
  const float rstd = 1.0f / sqrtf(q / static_cast<float>(C) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (gl + i * G) * 4;
    if (c < C) {
      const float4 g = ld4(gamma + c), b = ld4(beta + c);
      v[i].x = static_cast<float>(static_cast<T>((v[i].x - mean) * rstd * g.x + b.x));
      v[i].y = static_cast<float>(static_cast<T>((v[i].y - mean) * rstd * g.y + b.y));
      v[i].z = static_cast<float>(static_cast<T>((v[i].z - mean) * rstd * g.z + b.z));
      v[i].w = static_cast<float>(static_cast<T>((v[i].w - mean) * rstd * g.w + b.w));
    }
  }
}

// LayerNorm over C.  R/.../transformer.py:110,121; sal_unet.py:447,473
template <int G, int NV, typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out,
                                                        int M, int C, float eps) {
  constexpr int ROWS = 256 / G;
  const int gl = threadIdx.x % G;
  const int gr = threadIdx.x / G;
  for (long row = static_cast<long>(blockIdx.x) * ROWS + gr; row < M; row += static_cast<long>(gridDim.x) * ROWS) {
    const T* xr = x + row * C;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (gl + i * G) * 4;
      v[i] = c < C ? ld4(xr + c) : make_float4(0, 0, 0, 0);
    }
    ln_rows_finish<G, NV, T>(v, gl, C, gamma, beta, eps, out + row * C);
  }
}

// Up to three LayerNorms of one width in one launch (blockIdx.y = tensor): MViT's norm_q / norm_k / norm_v on the pooled tensors
// (R/models/mvit.py:562-570) -- the key / value tensors are 673 tokens per head, launches of their own cost more than their work.
struct LnMulti {
  const float* x[3]; const float* gamma[3]; const float* beta[3]; float* out[3];
  const float* dy[3]; double* part;      // backward
  int M[3];
  float eps[3];
};

template <int G, int NV>
__global__ __launch_bounds__(256) void layernorm_multi_kernel(LnMulti p, int C) {
  constexpr int ROWS = 256 / G;
  const int t = blockIdx.y;
  const int gl = threadIdx.x % G;
  const int gr = threadIdx.x / G;
  const float* __restrict__ x = p.x[t];
  float* __restrict__ out = p.out[t];
  for (long row = static_cast<long>(blockIdx.x) * ROWS + gr; row < p.M[t]; row += static_cast<long>(gridDim.x) * ROWS) {
    const float* xr = x + row * C;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (gl + i * G) * 4;
      v[i] = c < C ? ld4(xr + c) : make_float4(0, 0, 0, 0);
    }
    ln_rows_finish<G, NV, float>(v, gl, C, p.gamma[t], p.beta[t], p.eps[t], out + row * C);
  }
}

// depthwise 3x3 (pad 1, stride 1) + LayerNorm.  R/.../attention.py:36-47,94 (quirk Q8: centre slice)
// PRELN: the input is the block's un-normalised frames and every loaded token goes through the block's LayerNorm (pg, pb,
// peps) first -- see ln_rows_inplace.  Validity and trip counts are uniform within a lane group, so the group reductions
// inside always see a whole group.
template <int G, int NV, typename T, bool PRELN = false>
__device__ __forceinline__ void dwconv3_ln_body(const T* __restrict__ x, const float* __restrict__ w9,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                T* __restrict__ out, int N, int H, int W, int C, float eps, int bx, int nb,
                                                const float* __restrict__ pg = nullptr, const float* __restrict__ pb = nullptr,
                                                float peps = 0.f) {
  constexpr int ROWS = 256 / G;
  const int gl = threadIdx.x % G;
  const int gr = threadIdx.x / G;
  const long M = static_cast<long>(N) * H * W;
  for (long row = static_cast<long>(bx) * ROWS + gr; row < M; row += static_cast<long>(nb) * ROWS) {
    const int xw = static_cast<int>(row % W);
    const int yh = static_cast<int>((row / W) % H);
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = yh + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = xw + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const T* xr = x + (row + static_cast<long>(ky - 1) * W + (kx - 1)) * C;
        const float* wr = w9 + (ky * 3 + kx) * C;
        if constexpr (PRELN) {
          float4 a[NV];
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const int c = (gl + i * G) * 4;
            a[i] = c < C ? ld4(xr + c) : make_float4(0, 0, 0, 0);
          }
          ln_rows_inplace<G, NV, T>(a, gl, C, pg, pb, peps);
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const int c = (gl + i * G) * 4;
            if (c < C) {
              const float4 ww = ld4(wr + c);
              v[i].x = fmaf(a[i].x, ww.x, v[i].x);
              v[i].y = fmaf(a[i].y, ww.y, v[i].y);
              v[i].z = fmaf(a[i].z, ww.z, v[i].z);
              v[i].w = fmaf(a[i].w, ww.w, v[i].w);
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const int c = (gl + i * G) * 4;
            if (c < C) {
              const float4 a = ld4(xr + c), ww = ld4(wr + c);
              v[i].x = fmaf(a.x, ww.x, v[i].x);
              v[i].y = fmaf(a.y, ww.y, v[i].y);
              v[i].z = fmaf(a.z, ww.z, v[i].z);
              v[i].w = fmaf(a.w, ww.w, v[i].w);
            }
          }
        }
      }
    }
    ln_rows_finish<G, NV, T>(v, gl, C, gamma, beta, eps, out + row * C);
  }
}

template <int G, int NV, typename T>
__global__ __launch_bounds__(256) void dwconv3_ln_kernel(const T* __restrict__ x, const float* __restrict__ w9,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ out, int N, int H, int W, int C, float eps) {
  dwconv3_ln_body<G, NV, T>(x, w9, gamma, beta, out, N, H, W, C, eps, blockIdx.x, gridDim.x);
}

// Strip form of the kernel above for C <= 4 G (one float4 per lane): a lane group walks SEG output pixels along a row
// with a sliding 3x3 window in registers -- 3 new 16-byte loads per output instead of 9, and the nine weight vectors are
// loaded once per thread instead of once per output (18 -> ~4 vector loads per output: the kernel was bound by L1
// request rate, not by HBM).
template <int G, int SEG, typename T, bool PRELN = false>
__device__ __forceinline__ void dwconv3_ln_strip_body(const T* __restrict__ x, const float* __restrict__ w9,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      T* __restrict__ out, int N, int H, int W, int C, float eps, int bx, int nb,
                                                      const float* __restrict__ pg = nullptr,
                                                      const float* __restrict__ pb = nullptr, float peps = 0.f) {
  // PRELN: a loaded token is normalised (the block's LayerNorm) before it enters the window; a padding position stays zero
  auto fetch = [&](const T* p, bool valid) __attribute__((always_inline)) -> float4 {
    float4 t[1];
    t[0] = valid ? ld4(p) : make_float4(0, 0, 0, 0);
    if constexpr (PRELN) {
      ln_rows_inplace<G, 1, T>(t, threadIdx.x % G, C, pg, pb, peps);
      if (!valid) t[0] = make_float4(0, 0, 0, 0);
    }
    return t[0];
  };
  constexpr int GROUPS = 256 / G;
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  const int c = gl * 4;
  const bool act = c < C;
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = act ? ld4(w9 + t * C + c) : make_float4(0, 0, 0, 0);
  const int strips_w = (W + SEG - 1) / SEG;
  const long n_strips = static_cast<long>(N) * H * strips_w;
  const float4 zero = make_float4(0, 0, 0, 0);
  for (long sidx = static_cast<long>(bx) * GROUPS + gr; sidx < n_strips; sidx += static_cast<long>(nb) * GROUPS) {
    const int xs = static_cast<int>(sidx % strips_w) * SEG;
    const long ny = sidx / strips_w;
    const int y = static_cast<int>(ny % H);
    const T* rowp[3];
    bool rv[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
      rv[ky] = act && iy >= 0 && iy < H;
      rowp[ky] = x + ((ny + (ky - 1)) * W) * C + c;     // row iy of the same image (only dereferenced when valid)
    }
    float4 c0[3], c1[3], c2[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      c0[ky] = fetch(rowp[ky] + static_cast<long>(xs - 1) * C, rv[ky] && xs - 1 >= 0);
      c1[ky] = fetch(rowp[ky] + static_cast<long>(xs) * C, rv[ky]);
    }
    const int xe = min(xs + SEG, W);
    for (int xx = xs; xx < xe; ++xx) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) c2[ky] = fetch(rowp[ky] + static_cast<long>(xx + 1) * C, rv[ky] && xx + 1 < W);
      float4 v[1];
      v[0] = zero;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float4 a0 = c0[ky], a1 = c1[ky], a2 = c2[ky];
        const float4 w0 = w[ky * 3 + 0], w1 = w[ky * 3 + 1], w2 = w[ky * 3 + 2];
        v[0].x = fmaf(a2.x, w2.x, fmaf(a1.x, w1.x, fmaf(a0.x, w0.x, v[0].x)));
        v[0].y = fmaf(a2.y, w2.y, fmaf(a1.y, w1.y, fmaf(a0.y, w0.y, v[0].y)));
        v[0].z = fmaf(a2.z, w2.z, fmaf(a1.z, w1.z, fmaf(a0.z, w0.z, v[0].z)));
        v[0].w = fmaf(a2.w, w2.w, fmaf(a1.w, w1.w, fmaf(a0.w, w0.w, v[0].w)));
      }
      ln_rows_finish<G, 1, T>(v, gl, C, gamma, beta, eps, out + (ny * W + xx) * C);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) { c0[ky] = c1[ky]; c1[ky] = c2[ky]; }
    }
  }
}

template <int G, int SEG, typename T>
__global__ __launch_bounds__(256) void dwconv3_ln_strip_kernel(const T* __restrict__ x, const float* __restrict__ w9,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ out, int N, int H, int W, int C, float eps) {
  dwconv3_ln_strip_body<G, SEG, T>(x, w9, gamma, beta, out, N, H, W, C, eps, blockIdx.x, gridDim.x);
}

// depthwise k x k, stride k, no padding, + LayerNorm, for K and V at once: one workgroup per pooled token.
// R/.../attention.py:49-76,88-95.  Threads = (256/G position lanes) x (G channel lanes).
template <int G, int NV, typename T, bool PRELN = false, bool PROJ = false>
__device__ __forceinline__ void dwpool_ln_kv_body(const T* __restrict__ xk, const T* __restrict__ xv,
                                                  const float* __restrict__ wk, const float* __restrict__ wv,
                                                  const float* __restrict__ gk, const float* __restrict__ bk,
                                                  const float* __restrict__ gv, const float* __restrict__ bv,
                                                  T* __restrict__ ok, T* __restrict__ ov, int H, int W,
                                                  int C, int k, int gh, int gw, float eps, int tok,
                                                  const float* __restrict__ pg = nullptr, const float* __restrict__ pb = nullptr,
                                                  float peps = 0.f, bool ln_k = false, bool ln_v = false,
                                                  const T* __restrict__ pwk = nullptr, const float* __restrict__ pbk = nullptr,
                                                  const T* __restrict__ pwv = nullptr, const float* __restrict__ pbv = nullptr) {
  constexpr int PL = 256 / G;
  extern __shared__ float shp[];  // [2][PL][C] (+ [2][C] of T behind it when the projections are folded in)
  const int gl = threadIdx.x % G;
  const int pl = threadIdx.x / G;
  // tok = n * gh*gw + gy*gw + gx
  const int n = tok / (gh * gw);
  const int g = tok - n * gh * gw;
  const int gy = g / gw, gx = g - gy * gw;
  float4 ak[NV], av[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { ak[i] = make_float4(0, 0, 0, 0); av[i] = make_float4(0, 0, 0, 0); }
  const long img = static_cast<long>(n) * H * W;
  if constexpr (PRELN) {
    // The position loop has the same trip count for every lane of a position group, so the reductions see whole groups.
    // The loads of position pos + PL are issued before position pos is normalised and accumulated (one iteration is a
    // load -> reduce -> reduce -> FMA chain; without the look-ahead its global latency is paid k*k / PL times in a row), and
    // when key and value read the same tensor through the same LayerNorm it is loaded and normalised once.
    const bool same = (xk == xv) && (ln_k == ln_v);
    auto pos_off = [&](int pos) {
      const int dy = pos / k, dx = pos - dy * k;
      return (img + static_cast<long>(gy * k + dy) * W + (gx * k + dx)) * C;
    };
    auto fetch = [&](int pos, float4 (&a)[NV], float4 (&b)[NV]) {
      const long off = pos_off(pos);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (gl + i * G) * 4;
        a[i] = c < C ? ld4(xk + off + c) : make_float4(0, 0, 0, 0);
        if (!same) b[i] = c < C ? ld4(xv + off + c) : make_float4(0, 0, 0, 0);
      }
    };
    float4 a[NV], b[NV], an[NV], bn[NV];
    if (pl < k * k) fetch(pl, a, b);
    for (int pos = pl; pos < k * k; pos += PL) {
      const bool more = pos + PL < k * k;
      if (more) fetch(pos + PL, an, bn);
      if (ln_k) ln_rows_inplace<G, NV, T>(a, gl, C, pg, pb, peps);
      if (same) {
#pragma unroll
        for (int i = 0; i < NV; ++i) b[i] = a[i];
      } else if (ln_v) {
        ln_rows_inplace<G, NV, T>(b, gl, C, pg, pb, peps);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (gl + i * G) * 4;
        if (c < C) {
          const float4 w1 = ld4(wk + static_cast<long>(pos) * C + c), w2 = ld4(wv + static_cast<long>(pos) * C + c);
          ak[i].x = fmaf(a[i].x, w1.x, ak[i].x); ak[i].y = fmaf(a[i].y, w1.y, ak[i].y);
          ak[i].z = fmaf(a[i].z, w1.z, ak[i].z); ak[i].w = fmaf(a[i].w, w1.w, ak[i].w);
          av[i].x = fmaf(b[i].x, w2.x, av[i].x); av[i].y = fmaf(b[i].y, w2.y, av[i].y);
          av[i].z = fmaf(b[i].z, w2.z, av[i].z); av[i].w = fmaf(b[i].w, w2.w, av[i].w);
        }
      }
      if (more) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { a[i] = an[i]; b[i] = bn[i]; }
      }
    }
  }
  if constexpr (!PRELN) {
    for (int pos = pl; pos < k * k; pos += PL) {
      const int dy = pos / k, dx = pos - dy * k;
      const long off = (img + static_cast<long>(gy * k + dy) * W + (gx * k + dx)) * C;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (gl + i * G) * 4;
        if (c < C) {
          const float4 a = ld4(xk + off + c), w1 = ld4(wk + static_cast<long>(pos) * C + c);
          ak[i].x = fmaf(a.x, w1.x, ak[i].x); ak[i].y = fmaf(a.y, w1.y, ak[i].y);
          ak[i].z = fmaf(a.z, w1.z, ak[i].z); ak[i].w = fmaf(a.w, w1.w, ak[i].w);
          const float4 b = ld4(xv + off + c), w2 = ld4(wv + static_cast<long>(pos) * C + c);
          av[i].x = fmaf(b.x, w2.x, av[i].x); av[i].y = fmaf(b.y, w2.y, av[i].y);
          av[i].z = fmaf(b.z, w2.z, av[i].z); av[i].w = fmaf(b.w, w2.w, av[i].w);
        }
      }
    }
  }
  // cross position-lane reduction through LDS (fixed order => deterministic)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (gl + i * G) * 4;
    if (c < C) {
      st4(&shp[(0 * PL + pl) * C + c], ak[i]);
      st4(&shp[(1 * PL + pl) * C + c], av[i]);
    }
  }
  __syncthreads();
  if (pl < 2) {  // lane group 0 finishes K, lane group 1 finishes V
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (gl + i * G) * 4;
      v[i] = make_float4(0, 0, 0, 0);
      if (c < C) {
        for (int j = 0; j < PL; ++j) {
          const float4 t = ld4(&shp[(pl * PL + j) * C + c]);
          v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
        }
      }
    }
    if constexpr (PROJ) {
      // proj_k / proj_v (attention.py:79-80,98-99) folded in: the normalised token is parked in LDS, rounded to the storage
      // type as the stand-alone launch would have stored it
      T* park = reinterpret_cast<T*>(shp + 2 * PL * C);
      if (pl == 0) ln_rows_finish<G, NV, T>(v, gl, C, gk, bk, eps, park);
      else ln_rows_finish<G, NV, T>(v, gl, C, gv, bv, eps, park + C);
    } else {
      if (pl == 0) ln_rows_finish<G, NV, T>(v, gl, C, gk, bk, eps, ok + static_cast<long>(tok) * C);
      else ln_rows_finish<G, NV, T>(v, gl, C, gv, bv, eps, ov + static_cast<long>(tok) * C);
    }
  }
  if constexpr (PROJ) {
    __syncthreads();
    const T* park = reinterpret_cast<const T*>(shp + 2 * PL * C);
    // one output feature of K or V per group of G lanes (G = C / 12: the lanes of a group read consecutive 16-byte pieces of
    // one weight row -- coalesced, three independent loads per lane -- and combine on the VALU); 256 / G features per pass
    constexpr int FPP = 256 / G;
    const int fg = threadIdx.x / G, fl = threadIdx.x % G;
    for (int f0 = 0; f0 < 2 * C; f0 += FPP) {
      const int idx = min(f0 + fg, 2 * C - 1);                    // (2 C is a multiple of FPP for C = 96, 192)
      const int which = idx >= C, nn = idx - which * C;
      const T* wrow = (which ? pwv : pwk) + static_cast<long>(nn) * C;
      const T* xin = park + which * C;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (fl + i * G) * 4;
        const float4 w4 = ld4(wrow + c), x4 = ld4(xin + c);
        acc = fmaf(w4.x, x4.x, fmaf(w4.y, x4.y, fmaf(w4.z, x4.z, fmaf(w4.w, x4.w, acc))));
      }
      acc = group_sum<G>(acc);
      if (fl == 0 && f0 + fg < 2 * C) (which ? ov : ok)[static_cast<long>(tok) * C + nn] = static_cast<T>(acc + (which ? pbv : pbk)[nn]);
    }
  }
}

template <int G, int NV, typename T>
__global__ __launch_bounds__(256) void dwpool_ln_kv_kernel(const T* __restrict__ xk, const T* __restrict__ xv,
                                                           const float* __restrict__ wk, const float* __restrict__ wv,
                                                           const float* __restrict__ gk, const float* __restrict__ bk,
                                                           const float* __restrict__ gv, const float* __restrict__ bv,
                                                           T* __restrict__ ok, T* __restrict__ ov, int H, int W,
                                                           int C, int k, int gh, int gw, float eps) {
  dwpool_ln_kv_body<G, NV, T>(xk, xv, wk, wv, gk, bk, gv, bv, ok, ov, H, W, C, k, gh, gw, eps, blockIdx.x);
}

// The query branch (depthwise 3x3 + LayerNorm on every token) and the pooled key / value branch of a transformer block read
// the same normalised frames and do not depend on each other: one grid, the first nq workgroups run the query body, the
// rest one pooled token each.  (Two launches of 10-60 us, mostly latency at the coarse stages, become one.)
struct QkvPrepArgs {
  const void* xq; const float* w9; const float* gq; const float* bq; void* oq;
  const void* xk; const void* xv; const float* wk; const float* wv;
  const float* gk; const float* bk; const float* gv; const float* bv; void* ok; void* ov;
  int N, H, W, C, k, gh, gw, nq;
  float eps;
  // PRELN: the block's LayerNorm (transformer.py:110: x = self.norm(x)) applied to the tokens as they are loaded -- the
  // query input and the value input always, the key input when it is the same normalised tensor (visual-only)
  const float* pg; const float* pb; float peps; int ln_k;
  // pooled branch alone, small C: proj_k / proj_v folded in ([C][C] weights in the storage type, fp32 biases); null = off
  const void* pwk; const float* pbk; const void* pwv; const float* pbv;
};

template <bool STRIP, int G, int NV, typename T, bool PRELN, bool PROJ = false>
__global__ __launch_bounds__(256) void qkv_prep_kernel(QkvPrepArgs a) {
  const int b = blockIdx.x;
  if (b < a.nq) {
    if constexpr (STRIP)
      dwconv3_ln_strip_body<G, 8, T, PRELN>(static_cast<const T*>(a.xq), a.w9, a.gq, a.bq, static_cast<T*>(a.oq), a.N, a.H, a.W,
                                            a.C, a.eps, b, a.nq, a.pg, a.pb, a.peps);
    else
      dwconv3_ln_body<G, NV, T, PRELN>(static_cast<const T*>(a.xq), a.w9, a.gq, a.bq, static_cast<T*>(a.oq), a.N, a.H, a.W, a.C,
                                       a.eps, b, a.nq, a.pg, a.pb, a.peps);
  } else {
    dwpool_ln_kv_body<G, NV, T, PRELN, PROJ>(static_cast<const T*>(a.xk), static_cast<const T*>(a.xv), a.wk, a.wv, a.gk, a.bk, a.gv,
                                             a.bv, static_cast<T*>(a.ok), static_cast<T*>(a.ov), a.H, a.W, a.C, a.k, a.gh, a.gw, a.eps,
                                             b - a.nq, a.pg, a.pb, a.peps, a.ln_k != 0, true, static_cast<const T*>(a.pwk), a.pbk,
                                             static_cast<const T*>(a.pwv), a.pbv);
  }
}

// ------------------------------------------------------------------------------------------------
// K9 on 16-bit storage, 16-byte form (qkv_prep16_kernel).  The kernels above give a lane four channels: 8-byte accesses on
// 16-bit storage, one LayerNorm lane group per 4 G channels, the nine fp32 weight vectors of the depthwise filter re-read
// from L1 for every token (1.3-2.2 TB/s at 64 clips).  Here a token of C = 24 G channels is held by G lanes as THREE octets
// per lane (channel of octet i, element e: (gl + i G) 8 + e): every access of the token tensors is 16 bytes, a 64-lane
// wavefront holds 64 / G tokens, the per-channel constants (depthwise filter, LayerNorm affine) sit in LDS in lane order
// (conflict-free 16-byte reads), and the pooled branch adds its 64 / G position lanes up inside the wavefront (DPP row shifts +
// permlane swaps that keep the lane-in-group) before a [2][4][C] hand-off through LDS.  Same formulas as the forms above
// (two-pass LayerNorm, fmaf accumulation in tap / position order); the order of the additions inside a LayerNorm sum and of
// the pooled positions differs, so results agree to fp32 rounding, not bit for bit.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ float dpp_row_shr(float v) {      // lane l <- lane l - N inside its row of 16, 0 where there is none
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xF, 0xF, true));
}
// sum over the 64 / G lane groups of a wavefront of the values with the same lane-in-group; valid in the LAST G lanes
template <int G>
__device__ __forceinline__ float sum_over_groups(float v) {
  if constexpr (G <= 4) v += dpp_row_shr<4>(v);
  if constexpr (G <= 8) v += dpp_row_shr<8>(v);
  if constexpr (G <= 16) v += lane_xor16(v);
  v += lane_xor32(v);
  return v;
}

// per-channel fp32 vector -> LDS in lane order: dst[((i * 2 + hf) * G + gl) * 4 + e] = src[(gl + i G) 8 + hf 4 + e]
template <int G>
__device__ __forceinline__ void lane_order_fill(float* __restrict__ dst, const float* __restrict__ src, int C) {
  for (int c4 = threadIdx.x; c4 < C / 4; c4 += 256) {
    const int oct = c4 >> 1, hf = c4 & 1, i = oct / G, gl = oct - i * G;
    st4(dst + ((i * 2 + hf) * G + gl) * 4, ld4(src + c4 * 4));
  }
}
struct LaneVec { float v[24]; };
template <int G>
__device__ __forceinline__ LaneVec lane_order_read(const float* __restrict__ src, int gl) {
  LaneVec r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const float4 t = ld4(src + ((i * 2 + hf) * G + gl) * 4);
      r.v[i * 8 + hf * 4 + 0] = t.x; r.v[i * 8 + hf * 4 + 1] = t.y; r.v[i * 8 + hf * 4 + 2] = t.z; r.v[i * 8 + hf * 4 + 3] = t.w;
    }
  return r;
}

// two-pass LayerNorm of a token held as 24 values per lane by G lanes
template <int G>
__device__ __forceinline__ void ln24_stats(const float (&v)[24], int C, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 24; ++j) s += v[j];
  s = group_sum<G>(s);
  mean = s / static_cast<float>(C);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 24; ++j) { const float d = v[j] - mean; q = fmaf(d, d, q); }
  q = group_sum<G>(q);
  rstd = 1.0f / sqrtf(q / static_cast<float>(C) + eps);
}

template <typename T, int G>
__device__ __forceinline__ void load24(const T* __restrict__ p, int gl, float (&v)[24]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const f8v t = ld8(p + (gl + i * G) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[i * 8 + e] = t.v[e];
  }
}
template <typename T, int G>
__device__ __forceinline__ void store24(T* __restrict__ p, int gl, const float (&v)[24]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f8v t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t.v[e] = v[i * 8 + e];
    st8(p + (gl + i * G) * 8, t);
  }
}

// K8 on 16-bit storage with 16-byte accesses: a token of C = 24 G channels on G lanes, three octets per lane (layernorm_kernel
// above moves 8 bytes per lane there).  Two-pass statistics in registers as everywhere; the sums run in another order.
template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm16_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ out, int M, int C, float eps) {
  constexpr int ROWS = 256 / G;
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  float g[24], b[24];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int c = (gl + i * G) * 8 + hf * 4, j = i * 8 + hf * 4;
      const float4 g4 = ld4(gamma + c), b4 = ld4(beta + c);
      g[j] = g4.x; g[j + 1] = g4.y; g[j + 2] = g4.z; g[j + 3] = g4.w;
      b[j] = b4.x; b[j + 1] = b4.y; b[j + 2] = b4.z; b[j + 3] = b4.w;
    }
  for (long row = static_cast<long>(blockIdx.x) * ROWS + gr; row < M; row += static_cast<long>(gridDim.x) * ROWS) {
    float v[24];
    load24<T, G>(x + row * C, gl, v);
    float mean, rstd;
    ln24_stats<G>(v, C, eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < 24; ++j) v[j] = (v[j] - mean) * rstd * g[j] + b[j];
    store24<T, G>(out + row * C, gl, v);
  }
}

template <typename T, int G, bool PRELN>
__global__ __launch_bounds__(256, 3) void qkv_prep16_kernel(QkvPrepArgs a) {
  extern __shared__ float sh16[];
  constexpr int TPB = 256 / G;                  // tokens (query branch) / window positions (pooled branch) per pass
  const int C = a.C, gl = threadIdx.x % G, gr = threadIdx.x / G;
  if (static_cast<int>(blockIdx.x) < a.nq) {
    // ---- query branch: depthwise 3x3 (pad 1) + LayerNorm on every token (attention.py:36-47,94)
    float* wl = sh16;                            // [9][C] filter, then gamma, beta, each in lane order
    for (int t = 0; t < 9; ++t) lane_order_fill<G>(wl + t * C, a.w9 + t * C, C);
    lane_order_fill<G>(wl + 9 * C, a.gq, C);
    lane_order_fill<G>(wl + 10 * C, a.bq, C);
    __syncthreads();
    const T* __restrict__ x = static_cast<const T*>(a.xq);
    T* __restrict__ out = static_cast<T*>(a.oq);
    const int H = a.H, W = a.W;
    const int M = a.N * H * W;
    // XCD-aware walk (a.nq is a multiple of 8; workgroups are dealt round-robin over the XCDs): XCD x owns a contiguous range of token
    // chunks, so the rows above and below a token -- W tokens away -- are fetched into the same L2 instead of into three
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = a.nq >> 3;
    const int n_chunk = (M + TPB - 1) / TPB, cpx = (n_chunk + 7) >> 3;
    for (int chunk = slot; chunk < cpx; chunk += wpx) {
      const int row = (xcd * cpx + chunk) * TPB + gr;
      if (row >= M) continue;
      const int xw = row % W, yh = (row / W) % H;
      float acc[24];
#pragma unroll
      for (int j = 0; j < 24; ++j) acc[j] = 0.f;
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky)         // (one kernel row per trip: unrolled over all nine taps hipcc hoists 27 + 54 loads and spills)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tap = ky * 3 + kx;
        const int iy = yh + ky - 1, ix = xw + kx - 1;
        // a padding tap reads the centre pixel and enters with weight 0 (no branch: lane groups of one wavefront differ here, and
        // the divergent form measured 30 % slower)
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const T* xr = x + static_cast<long>(ok ? row + (ky - 1) * W + (kx - 1) : row) * C;
        const float m = ok ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const f8v t = ld8(xr + (gl + i * G) * 8);
          float4 w0 = ld4(wl + tap * C + ((i * 2 + 0) * G + gl) * 4), w1 = ld4(wl + tap * C + ((i * 2 + 1) * G + gl) * 4);
          w0.x *= m; w0.y *= m; w0.z *= m; w0.w *= m; w1.x *= m; w1.y *= m; w1.z *= m; w1.w *= m;
          acc[i * 8 + 0] = fmaf(t.v[0], w0.x, acc[i * 8 + 0]); acc[i * 8 + 1] = fmaf(t.v[1], w0.y, acc[i * 8 + 1]);
          acc[i * 8 + 2] = fmaf(t.v[2], w0.z, acc[i * 8 + 2]); acc[i * 8 + 3] = fmaf(t.v[3], w0.w, acc[i * 8 + 3]);
          acc[i * 8 + 4] = fmaf(t.v[4], w1.x, acc[i * 8 + 4]); acc[i * 8 + 5] = fmaf(t.v[5], w1.y, acc[i * 8 + 5]);
          acc[i * 8 + 6] = fmaf(t.v[6], w1.z, acc[i * 8 + 6]); acc[i * 8 + 7] = fmaf(t.v[7], w1.w, acc[i * 8 + 7]);
        }
      }
      float mean, rstd;
      ln24_stats<G>(acc, C, a.eps, mean, rstd);
      const LaneVec g = lane_order_read<G>(wl + 9 * C, gl), b = lane_order_read<G>(wl + 10 * C, gl);
#pragma unroll
      for (int j = 0; j < 24; ++j) acc[j] = (acc[j] - mean) * rstd * g.v[j] + b.v[j];
      store24<T, G>(out + static_cast<long>(row) * C, gl, acc);
    }
    return;
  }
  // ---- pooled key / value branch: depthwise k x k stride k + LayerNorm, one workgroup per pooled token (attention.py:49-76,88-95)
  float* pre = sh16;                             // [2][C] the block's LayerNorm affine (PRELN), lane order
  float* part = sh16 + 2 * C;                    // [2][4][C] per-wavefront sums
  if constexpr (PRELN) {
    lane_order_fill<G>(pre, a.pg, C);
    lane_order_fill<G>(pre + C, a.pb, C);
    __syncthreads();
  }
  const T* __restrict__ xk = static_cast<const T*>(a.xk);
  const T* __restrict__ xv = static_cast<const T*>(a.xv);
  const int tok = blockIdx.x - a.nq;
  const int k = a.k, gh = a.gh, gw = a.gw, H = a.H, W = a.W;
  const int n = tok / (gh * gw), cell = tok - n * gh * gw;
  const int gy = cell / gw, gx = cell - gy * gw;
  const bool same = (xk == xv) && (!PRELN || a.ln_k != 0);
  float ak[24], av[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) { ak[j] = 0.f; av[j] = 0.f; }
  const long img = static_cast<long>(n) * H * W;
#pragma unroll 1
  for (int pos = gr; pos < k * k; pos += TPB) {
    const int dy = pos / k, dx = pos - dy * k;
    const long off = (img + static_cast<long>(gy * k + dy) * W + (gx * k + dx)) * C;
    float tk[24], tv[24];
    load24<T, G>(xk + off, gl, tk);
    if (!same) load24<T, G>(xv + off, gl, tv);
    if constexpr (PRELN) {
      // the block's first LayerNorm on the loaded token, rounded through the storage type as the stand-alone launch stores it
      float mean, rstd;
      if (a.ln_k) {
        ln24_stats<G>(tk, C, a.peps, mean, rstd);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float4 g4 = ld4(pre + ((i * 2 + hf) * G + gl) * 4), b4 = ld4(pre + C + ((i * 2 + hf) * G + gl) * 4);
            const int j = i * 8 + hf * 4;
            tk[j] = static_cast<float>(static_cast<T>((tk[j] - mean) * rstd * g4.x + b4.x));
            tk[j + 1] = static_cast<float>(static_cast<T>((tk[j + 1] - mean) * rstd * g4.y + b4.y));
            tk[j + 2] = static_cast<float>(static_cast<T>((tk[j + 2] - mean) * rstd * g4.z + b4.z));
            tk[j + 3] = static_cast<float>(static_cast<T>((tk[j + 3] - mean) * rstd * g4.w + b4.w));
          }
      }
      if (!same) {
        ln24_stats<G>(tv, C, a.peps, mean, rstd);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float4 g4 = ld4(pre + ((i * 2 + hf) * G + gl) * 4), b4 = ld4(pre + C + ((i * 2 + hf) * G + gl) * 4);
            const int j = i * 8 + hf * 4;
            tv[j] = static_cast<float>(static_cast<T>((tv[j] - mean) * rstd * g4.x + b4.x));
            tv[j + 1] = static_cast<float>(static_cast<T>((tv[j + 1] - mean) * rstd * g4.y + b4.y));
            tv[j + 2] = static_cast<float>(static_cast<T>((tv[j + 2] - mean) * rstd * g4.z + b4.z));
            tv[j + 3] = static_cast<float>(static_cast<T>((tv[j + 3] - mean) * rstd * g4.w + b4.w));
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int c = (gl + i * G) * 8 + hf * 4, j = i * 8 + hf * 4;
        const float4 p1 = ld4(a.wk + static_cast<long>(pos) * C + c), p2 = ld4(a.wv + static_cast<long>(pos) * C + c);
        ak[j] = fmaf(tk[j], p1.x, ak[j]); ak[j + 1] = fmaf(tk[j + 1], p1.y, ak[j + 1]);
        ak[j + 2] = fmaf(tk[j + 2], p1.z, ak[j + 2]); ak[j + 3] = fmaf(tk[j + 3], p1.w, ak[j + 3]);
        av[j] = fmaf(same ? tk[j] : tv[j], p2.x, av[j]); av[j + 1] = fmaf(same ? tk[j + 1] : tv[j + 1], p2.y, av[j + 1]);
        av[j + 2] = fmaf(same ? tk[j + 2] : tv[j + 2], p2.z, av[j + 2]); av[j + 3] = fmaf(same ? tk[j + 3] : tv[j + 3], p2.w, av[j + 3]);
      }
  }
  // position lanes of a wavefront, then the four wavefronts (fixed order)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 24; ++j) { ak[j] = sum_over_groups<G>(ak[j]); av[j] = sum_over_groups<G>(av[j]); }
  if (lane >= 64 - G) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int c = (gl + i * G) * 8 + hf * 4, j = i * 8 + hf * 4;
        st4(part + (0 * 4 + wave) * C + c, make_float4(ak[j], ak[j + 1], ak[j + 2], ak[j + 3]));
        st4(part + (1 * 4 + wave) * C + c, make_float4(av[j], av[j + 1], av[j + 2], av[j + 3]));
      }
  }
  __syncthreads();
  if (gr < 2) {                                  // lane group 0 finishes K, lane group 1 V
    float v[24];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int c = (gl + i * G) * 8 + hf * 4, j = i * 8 + hf * 4;
        float4 t = ld4(part + (gr * 4 + 0) * C + c);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float4 u = ld4(part + (gr * 4 + w) * C + c);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
      }
    float mean, rstd;
    ln24_stats<G>(v, C, a.eps, mean, rstd);
    const float* gm = gr == 0 ? a.gk : a.gv;
    const float* bt = gr == 0 ? a.bk : a.bv;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int c = (gl + i * G) * 8 + hf * 4, j = i * 8 + hf * 4;
        const float4 g4 = ld4(gm + c), b4 = ld4(bt + c);
        v[j] = (v[j] - mean) * rstd * g4.x + b4.x; v[j + 1] = (v[j + 1] - mean) * rstd * g4.y + b4.y;
        v[j + 2] = (v[j + 2] - mean) * rstd * g4.z + b4.z; v[j + 3] = (v[j + 3] - mean) * rstd * g4.w + b4.w;
      }
    store24<T, G>((gr == 0 ? static_cast<T*>(a.ok) : static_cast<T*>(a.ov)) + static_cast<long>(tok) * C, gl, v);
  }
}

// Dispatch on C: lane-group width G = min(64, pow2 >= C/4), NV = ceil(C/4/G).
#define DS_ROW_DISPATCH(C, CALL)                                  \
  do {                                                            \
    const int c4 = (C) / 4;                                       \
    if (c4 <= 8) { CALL(8, 1); }                                  \
    else if (c4 <= 16) { CALL(16, 1); }                           \
    else if (c4 <= 32) { CALL(32, 1); }                           \
    else if (c4 <= 64) { CALL(64, 1); }                           \
    else if (c4 <= 128) { CALL(64, 2); }                          \
    else if (c4 <= 192) { CALL(64, 3); }                          \
    else if (c4 <= 256) { CALL(64, 4); }                          \
    else { set_error("channel count %d > 1024 unsupported", (C)); return DIFFSAL_E_SHAPE; } \
  } while (0)

}  // namespace diffsal

using namespace diffsal;

extern "C" size_t diffsal_groupnorm_ws_bytes(int B, int groups) {
  return static_cast<size_t>(B) * GN_CHUNKS_MAX * groups * 2 * sizeof(double);
}

// GroupNorm (+ swish) of fp32 maps in ONE launch when an (image, group) slab fits the LDS: a 1024-thread workgroup per
// (image, group) parks its HW x (C / groups) floats in LDS while it sums them (mean), reads them back for the centred second moment
// (the two-pass form, as torch) and once more to normalise: x is read once and out written once, no statistics round trip
// through memory and no second launch (K3: six GroupNorms per step were twelve launches of ~8 us, 0.8 TB/s).  A pixel's group
// slice is VEC * U consecutive floats (3, 6, 12 or 24 at 32 groups and C = 96 .. 768); item i = (pixel i / U, piece i % U).
// R/models/saliency_decoder/sal_unet.py:36-44 (Normalize + nonlinearity).
// T: storage type of x / out (fp32, or bf16 / f16: the slab in LDS and all arithmetic stay fp32, one rounding on the way out)
template <typename T, int VEC, int U>
__global__ __launch_bounds__(1024) void gn_slab_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ out, int HW, int C,
                                                       int groups, float eps, int swish) {
  constexpr int CPG = VEC * U;
  extern __shared__ __attribute__((aligned(16))) float gn_slab[];
  __shared__ double red[16];
  __shared__ float gb[2 * CPG];
  const int tid = threadIdx.x, g = blockIdx.x, n = blockIdx.y;
  const int items = HW * U;
  const T* xs = x + static_cast<long>(n) * HW * C + g * CPG;
  T* os = out + static_cast<long>(n) * HW * C + g * CPG;
  if (tid < CPG) { gb[tid] = gamma[g * CPG + tid]; gb[CPG + tid] = beta[g * CPG + tid]; }
  struct __attribute__((packed, aligned(VEC == 4 ? 16 : 4))) Piece { float v[VEC]; };   // C / groups = 12, 24: 16-byte aligned pieces
  struct __attribute__((packed, aligned(VEC == 4 ? 4 * sizeof(T) : sizeof(T)))) PieceT { T v[VEC]; };   // the same piece in memory
  auto block_sum = [&](float v) -> double {        // fp32 inside a wavefront, fp64 across the 16 wavefronts, fixed order
    v = group_sum<64>(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = static_cast<double>(v);
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
  };
  float s = 0.f;
#pragma unroll 4
  for (int i = tid; i < items; i += 1024) {
    const int pix = i / U, u = i - pix * U;
    const PieceT pt = *reinterpret_cast<const PieceT*>(xs + static_cast<long>(pix) * C + u * VEC);
    Piece p;
#pragma unroll
    for (int e = 0; e < VEC; ++e) p.v[e] = static_cast<float>(pt.v[e]);
    *reinterpret_cast<Piece*>(gn_slab + i * VEC) = p;
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += p.v[e];
  }
  const double cnt = static_cast<double>(HW) * CPG;
  const float mean = static_cast<float>(block_sum(s) / cnt);
  float q = 0.f;
  for (int i = tid; i < items; i += 1024) {
    const Piece p = *reinterpret_cast<const Piece*>(gn_slab + i * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { const float d = p.v[e] - mean; q = fmaf(d, d, q); }
  }
  const float rstd = static_cast<float>(1.0 / sqrt(block_sum(q) / cnt + static_cast<double>(eps)));
#pragma unroll 4
  for (int i = tid; i < items; i += 1024) {
    const int pix = i / U, u = i - pix * U;
    Piece p = *reinterpret_cast<const Piece*>(gn_slab + i * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float y = (p.v[e] - mean) * rstd * gb[u * VEC + e] + gb[CPG + u * VEC + e];
      p.v[e] = swish ? swishf(y) : y;
    }
    PieceT po;
#pragma unroll
    for (int e = 0; e < VEC; ++e) po.v[e] = static_cast<T>(p.v[e]);
    *reinterpret_cast<PieceT*>(os + static_cast<long>(pix) * C + u * VEC) = po;
  }
}

// 1 if launched, 0 if the shape does not fit (the caller runs the statistics + normalisation launches)
template <typename T>
static int try_gn_slab(const T* x, const float* gamma, const float* beta, T* out, int B, int HW, int C, int groups, float eps,
                       int swish, hipStream_t s) {
  if (tune(TUNE_GN_CHUNKS) > 0 || tune(TUNE_GN_APPLY_WGS) > 0 || tune(TUNE_NO_GN_SLAB) == 1) return 0;   // the two-launch path is being tuned / forced
  const int cpg = C / groups;
  const size_t lds = static_cast<size_t>(HW) * cpg * sizeof(float);
  // fewer than 128 workgroups (one per CU, streaming its slab at a single CU's ~30 GB/s) lose to the two launches, which spread an
  // image over up to 512 workgroups: measured equal at B = 4 (six GroupNorms of a step: 0.109 ms either way, 6 launches instead
  // of 12), slower below
  // ... and more than ~1000 slabs lose to them as well (16-bit storage, 64 clips: 21.55 against 20.95 ms per step, same box): past
  // that the two launches' whole-row coalesced passes win over 2048 workgroups reading 6-48-byte pieces
  if (lds > 150 * 1024 || static_cast<long>(B) * groups < 128 || static_cast<long>(B) * groups > 1024) return 0;
#define GN_SLAB(VEC, U)                                                                                          \
  do {                                                                                                           \
    DS_RAISE_DYNAMIC_LDS((gn_slab_kernel<T, VEC, U>), 152 * 1024);                                               \
    hipLaunchKernelGGL((gn_slab_kernel<T, VEC, U>), dim3(groups, B), dim3(1024), lds, s, x, gamma, beta, out, HW, C, groups, eps, swish); \
  } while (0)
  if (cpg == 3) GN_SLAB(3, 1);
  else if (cpg == 6) GN_SLAB(3, 2);
  else if (cpg == 12) GN_SLAB(4, 3);
  else if (cpg == 24) GN_SLAB(4, 6);
  else return 0;
#undef GN_SLAB
  return check_launch("groupnorm_swish(slab)") == DIFFSAL_OK ? 1 : DIFFSAL_E_LAUNCH;
}

template <typename T>
static int groupnorm_swish_t(const T* x, const float* gamma, const float* beta, T* out, int B, int HW, int C, int groups,
                             float eps, void* ws, hipStream_t s, int swish = 1) {
  {
    const int r = try_gn_slab<T>(x, gamma, beta, out, B, HW, C, groups, eps, swish, s);
    if (r != 0) return r < 0 ? r : DIFFSAL_OK;
  }
  const int chunks = gn_chunks();
  hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(chunks, B), dim3(256), 2 * C * sizeof(double), s, x,
                     static_cast<double*>(ws), HW, C, groups);
  int rc = check_launch("groupnorm_swish(stats)");
  if (rc) return rc;
  const long total4 = static_cast<long>(HW) * (C / 4);
  int gx = static_cast<int>((total4 + 255) / 256);
  // ~512 workgroups in all (each repeats the per-image prologue: more of them cost more than they hide; tools/bench_gn.py)
  int cap = 512 / B;
  cap = cap < 32 ? 32 : cap;
  if (tune(TUNE_GN_APPLY_WGS) > 0) cap = tune(TUNE_GN_APPLY_WGS);
  gx = gx > cap ? cap : gx;
  hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(gx, B), dim3(256), (2 * C + 2 * groups) * sizeof(float), s, x,
                     static_cast<const double*>(ws), gamma, beta, out, HW, C, groups, eps, swish, chunks);
  return check_launch("groupnorm_swish(apply)");
}

extern "C" int diffsal_groupnorm_swish(const void* x, const float* gamma, const float* beta, void* out, int B,
                                       int HW, int C, int groups, float eps, void* ws, size_t ws_bytes, int dtype,
                                       diffsal_stream_t stream) {
  DS_REQUIRE(x && gamma && beta && out && ws, DIFFSAL_E_ARG, "groupnorm_swish: null argument");
  DS_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && C <= 1024, DIFFSAL_E_SHAPE,
             "groupnorm_swish: bad shape B=%d HW=%d C=%d groups=%d", B, HW, C, groups);
  DS_REQUIRE(ws_bytes >= diffsal_groupnorm_ws_bytes(B, groups), DIFFSAL_E_ARG, "groupnorm_swish: workspace too small");
  DS_REQUIRE(aligned16(x) && aligned16(out) && aligned16(ws), DIFFSAL_E_ALIGN, "groupnorm_swish: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(T) return groupnorm_swish_t<T>(static_cast<const T*>(x), gamma, beta, static_cast<T*>(out), B, HW, C, groups, eps, ws, s)
  DS_DTYPE_DISPATCH(dtype, "groupnorm_swish", CALL);
#undef CALL
  return DIFFSAL_OK;
}

// GroupNorm statistics -> affine form ab [B][2][C] (scale row, shift row): same chunk sums and fp64 finish as gn_apply_kernel's
// prologue, one workgroup per image.
namespace diffsal {
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ ws, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ ab, int HW, int C,
                                                          int groups, float eps, int GN_CHUNKS) {
  // grid (B, ceil(groups / 4)): a wavefront per group, lane k takes chunks k, k + 64, ...: one round trip to memory
  const int b = blockIdx.x, g = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= groups) return;
  const int cpg = C / groups;
  double s = 0, q = 0;
  for (int k = lane; k < GN_CHUNKS; k += 64) {
    const double* o = ws + ((static_cast<long>(b) * GN_CHUNKS + k) * groups + g) * 2;
    s += o[0]; q += o[1];
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  const double n = static_cast<double>(HW) * cpg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  var = var < 0 ? 0 : var;
  const float mf = static_cast<float>(mean), rf = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
    const float sc = rf * gamma[c];
    ab[static_cast<long>(b) * 2 * C + c] = sc;
    ab[static_cast<long>(b) * 2 * C + C + c] = beta[c] - mf * sc;
  }
}
}  // namespace diffsal

extern "C" int diffsal_gn_affine(const void* x, const float* gamma, const float* beta, float* ab, int B, int HW, int C, int groups,
                                 float eps, void* ws, size_t ws_bytes, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(x && gamma && beta && ab && ws, DIFFSAL_E_ARG, "gn_affine: null argument");
  DS_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && C <= 1024, DIFFSAL_E_SHAPE,
             "gn_affine: bad shape B=%d HW=%d C=%d groups=%d", B, HW, C, groups);
  DS_REQUIRE(ws_bytes >= diffsal_groupnorm_ws_bytes(B, groups), DIFFSAL_E_ARG, "gn_affine: workspace too small");
  DS_REQUIRE(aligned16(x) && aligned16(ab) && aligned16(ws), DIFFSAL_E_ALIGN, "gn_affine: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int chunks = gn_chunks();
#define CALL(T)                                                                                                                  \
  hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(chunks, B), dim3(256), 2 * C * sizeof(double), s, static_cast<const T*>(x),       \
                     static_cast<double*>(ws), HW, C, groups)
  DS_DTYPE_DISPATCH(dtype, "gn_affine", CALL);
#undef CALL
  int rc = check_launch("gn_affine(stats)");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B, (groups + 3) / 4), dim3(256), 0, s, static_cast<const double*>(ws), gamma, beta,
                     ab, HW, C, groups, eps, chunks);
  return check_launch("gn_affine(finish)");
}

// GroupNorm with the activation optional: act = 0 plain affine GroupNorm (AttnBlock.norm of the legacy UNet,
// R/models/diffusion_decoder/diffusion.py:145,173), act = 1 the K3 form above.
extern "C" int diffsal_groupnorm(const void* x, const float* gamma, const float* beta, void* out, int B, int HW, int C,
                                 int groups, float eps, int act, void* ws, size_t ws_bytes, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(x && gamma && beta && out && ws, DIFFSAL_E_ARG, "groupnorm: null argument");
  DS_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && C <= 1024 && (act == 0 || act == 1),
             DIFFSAL_E_SHAPE, "groupnorm: bad shape B=%d HW=%d C=%d groups=%d act=%d", B, HW, C, groups, act);
  DS_REQUIRE(ws_bytes >= diffsal_groupnorm_ws_bytes(B, groups), DIFFSAL_E_ARG, "groupnorm: workspace too small");
  DS_REQUIRE(aligned16(x) && aligned16(out) && aligned16(ws), DIFFSAL_E_ALIGN, "groupnorm: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(T) return groupnorm_swish_t<T>(static_cast<const T*>(x), gamma, beta, static_cast<T*>(out), B, HW, C, groups, eps, ws, s, act)
  DS_DTYPE_DISPATCH(dtype, "groupnorm", CALL);
#undef CALL
  return DIFFSAL_OK;
}

static int row_grid(long rows, int rows_per_block) {
  long g = (rows + rows_per_block - 1) / rows_per_block;
  return static_cast<int>(g > 8192 ? 8192 : g);
}

template <typename T>
static int layernorm_t(const T* x, const float* gamma, const float* beta, T* out, int M, int C, float eps, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    const int G = C / 24;
    if (tune(TUNE_NO_STREAM16) != 1 && C % 24 == 0 && (G == 4 || G == 8 || G == 16 || G == 32)) {
      int grid = row_grid(M, 256 / G);
      grid = grid > 4096 ? 4096 : grid;            // each workgroup keeps gamma / beta in registers: a few rows per lane group
#define CALL16(GV) hipLaunchKernelGGL((layernorm16_kernel<T, GV>), dim3(grid), dim3(256), 0, s, x, gamma, beta, out, M, C, eps)
      if (G == 4) CALL16(4); else if (G == 8) CALL16(8); else if (G == 16) CALL16(16); else CALL16(32);
#undef CALL16
      return check_launch("layernorm(16-bit)");
    }
  }
#define CALL(G, NV) \
  hipLaunchKernelGGL((layernorm_kernel<G, NV, T>), dim3(row_grid(M, 256 / G)), dim3(256), 0, s, x, gamma, beta, out, M, C, eps)
  DS_ROW_DISPATCH(C, CALL);
#undef CALL
  return check_launch("layernorm");
}

extern "C" int diffsal_layernorm_multi(const float* const* x, const float* const* gamma, const float* const* beta, float* const* out,
                                       const int* M, int n, int C, const float* eps, diffsal_stream_t stream) {
  DS_REQUIRE(x && gamma && beta && out && M && eps && n >= 1 && n <= 3, DIFFSAL_E_ARG, "layernorm_multi: null argument or n=%d not in 1..3", n);
  DS_REQUIRE(C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "layernorm_multi: bad width C=%d", C);
  LnMulti a{};
  int most = 0;
  for (int t = 0; t < n; ++t) {
    DS_REQUIRE(x[t] && gamma[t] && beta[t] && out[t] && M[t] > 0, DIFFSAL_E_ARG, "layernorm_multi: null tensor %d", t);
    DS_REQUIRE(aligned16(x[t]) && aligned16(out[t]) && aligned16(gamma[t]) && aligned16(beta[t]), DIFFSAL_E_ALIGN,
               "layernorm_multi: misaligned pointer");
    a.x[t] = x[t]; a.gamma[t] = gamma[t]; a.beta[t] = beta[t]; a.out[t] = out[t]; a.M[t] = M[t]; a.eps[t] = eps[t];
    most = M[t] > most ? M[t] : most;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(G, NV) hipLaunchKernelGGL((layernorm_multi_kernel<G, NV>), dim3(row_grid(most, 256 / G), n), dim3(256), 0, s, a, C)
  DS_ROW_DISPATCH(C, CALL);
#undef CALL
  return check_launch("layernorm_multi");
}

extern "C" int diffsal_layernorm(const void* x, const float* gamma, const float* beta, void* out, int M, int C,
                                 float eps, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(x && gamma && beta && out, DIFFSAL_E_ARG, "layernorm: null argument");
  DS_REQUIRE(M > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "layernorm: bad shape M=%d C=%d", M, C);
  DS_REQUIRE(aligned16(x) && aligned16(out) && aligned16(gamma) && aligned16(beta), DIFFSAL_E_ALIGN,
             "layernorm: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALLT(T) return layernorm_t<T>(static_cast<const T*>(x), gamma, beta, static_cast<T*>(out), M, C, eps, s)
  DS_DTYPE_DISPATCH(dtype, "layernorm", CALLT);
#undef CALLT
  return DIFFSAL_OK;
}

template <typename T>
static int dwconv3_ln_t(const T* x, const float* w9, const float* gamma, const float* beta, T* out, int N, int H, int W,
                        int C, float eps, hipStream_t s) {
  const long M = static_cast<long>(N) * H * W;
  if (C <= 256 && W >= 16) {   // one float4 per lane: sliding-window strips along the row
    constexpr int SEG = 8;
    const long n_strips = static_cast<long>(N) * H * ((W + SEG - 1) / SEG);
#define CALLS(G) \
  hipLaunchKernelGGL((dwconv3_ln_strip_kernel<G, SEG, T>), dim3(row_grid(n_strips, 256 / G)), dim3(256), 0, s, x, w9, gamma, beta, out, N, H, W, C, eps)
    if (C <= 32) { CALLS(8); } else if (C <= 64) { CALLS(16); } else if (C <= 128) { CALLS(32); } else { CALLS(64); }
#undef CALLS
    return check_launch("dwconv3_ln(strip)");
  }
#define CALL(G, NV)                                                                                                 \
  hipLaunchKernelGGL((dwconv3_ln_kernel<G, NV, T>), dim3(row_grid(M, 256 / G)), dim3(256), 0, s, x, w9, gamma, beta, \
                     out, N, H, W, C, eps)
  DS_ROW_DISPATCH(C, CALL);
#undef CALL
  return check_launch("dwconv3_ln");
}

extern "C" int diffsal_dwconv3_ln(const void* x, const float* w9, const float* gamma, const float* beta, void* out,
                                  int N, int H, int W, int C, float eps, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(x && w9 && gamma && beta && out, DIFFSAL_E_ARG, "dwconv3_ln: null argument");
  DS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "dwconv3_ln: bad shape");
  DS_REQUIRE(aligned16(x) && aligned16(out) && aligned16(w9) && aligned16(gamma) && aligned16(beta), DIFFSAL_E_ALIGN,
             "dwconv3_ln: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALLT(T) return dwconv3_ln_t<T>(static_cast<const T*>(x), w9, gamma, beta, static_cast<T*>(out), N, H, W, C, eps, s)
  DS_DTYPE_DISPATCH(dtype, "dwconv3_ln", CALLT);
#undef CALLT
  return DIFFSAL_OK;
}

template <typename T>
static int dwpool_ln_kv_t(const T* xk, const T* xv, const float* wk, const float* wv, const float* gk, const float* bk,
                          const float* gv, const float* bv, T* out_k, T* out_v, int N, int H, int W, int C, int k,
                          float eps, hipStream_t s) {
  const int gh = (H - k) / k + 1, gw = (W - k) / k + 1;
#define CALL(G, NV)                                                                                                \
  hipLaunchKernelGGL((dwpool_ln_kv_kernel<G, NV, T>), dim3(N * gh * gw), dim3(256), 2 * (256 / G) * C * sizeof(float), s, \
                     xk, xv, wk, wv, gk, bk, gv, bv, out_k, out_v, H, W, C, k, gh, gw, eps)
  DS_ROW_DISPATCH(C, CALL);
#undef CALL
  return check_launch("dwpool_ln_kv");
}

extern "C" int diffsal_dwpool_ln_kv(const void* xk, const void* xv, const float* wk, const float* wv,
                                    const float* gk, const float* bk, const float* gv, const float* bv, void* out_k,
                                    void* out_v, int N, int H, int W, int C, int k, float eps, int dtype,
                                    diffsal_stream_t stream) {
  DS_REQUIRE(xk && xv && wk && wv && gk && bk && gv && bv && out_k && out_v, DIFFSAL_E_ARG, "dwpool_ln_kv: null argument");
  DS_REQUIRE(N > 0 && C > 0 && C % 4 == 0 && k > 0 && H >= k && W >= k, DIFFSAL_E_SHAPE,
             "dwpool_ln_kv: bad shape H=%d W=%d C=%d k=%d", H, W, C, k);
  DS_REQUIRE(aligned16(xk) && aligned16(xv) && aligned16(wk) && aligned16(wv) && aligned16(out_k) && aligned16(out_v),
             DIFFSAL_E_ALIGN, "dwpool_ln_kv: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALLT(T)                                                                                                  \
  return dwpool_ln_kv_t<T>(static_cast<const T*>(xk), static_cast<const T*>(xv), wk, wv, gk, bk, gv, bv,          \
                           static_cast<T*>(out_k), static_cast<T*>(out_v), N, H, W, C, k, eps, s)
  DS_DTYPE_DISPATCH(dtype, "dwpool_ln_kv", CALLT);
#undef CALLT
  return DIFFSAL_OK;
}

template <typename T>
static int qkv_prep_t(QkvPrepArgs a, hipStream_t s) {
  const int nkv = a.N * a.gh * a.gw;
  if constexpr (sizeof(T) == 2) {
    // 16-bit storage, the decoder's widths: the 16-byte form (qkv_prep16_kernel); its query branch has no folded LayerNorm (that
    // combination -- SalUNet.fold_norm1 -- is measured slower and off: it keeps the form above)
    const int G = a.C / 24;
    const long rows = static_cast<long>(a.N) * a.H * a.W;
    if (tune(TUNE_NO_STREAM16) != 1 && a.C % 24 == 0 && (G == 4 || G == 8 || G == 16 || G == 32) && !a.pwk && !(a.oq && a.pg) && rows < (1L << 31) &&
        aligned16(a.pg ? a.pg : a.gk)) {
      a.nq = a.oq ? row_grid(rows, 256 / G) : 0;
      if (a.nq > 2048) a.nq = 2048;              // each workgroup stages the filter in LDS: a few passes per workgroup
      a.nq = (a.nq + 7) / 8 * 8;                 // XCD-aware walk of the query branch
      const size_t lds = static_cast<size_t>(a.C) * sizeof(float) * (a.oq ? 11 : 10);
#define CALL16(GV)                                                                                                       \
  do {                                                                                                                   \
    if (a.pg) hipLaunchKernelGGL((qkv_prep16_kernel<T, GV, true>), dim3(a.nq + nkv), dim3(256), lds, s, a);                 \
    else hipLaunchKernelGGL((qkv_prep16_kernel<T, GV, false>), dim3(a.nq + nkv), dim3(256), lds, s, a);                     \
  } while (0)
      if (G == 4) CALL16(4); else if (G == 8) CALL16(8); else if (G == 16) CALL16(16); else CALL16(32);
#undef CALL16
      return check_launch("qkv_prep(16-bit)");
    }
  }
  const bool strip = a.C <= 256 && a.W >= 16;
  const size_t lds_of = 2 * a.C * sizeof(float);     // x (256 / G) position lanes
#define CALL(G, NV)                                                                                                  \
  do {                                                                                                               \
    if (strip && NV == 1) {                                                                                          \
      a.nq = a.oq ? row_grid(static_cast<long>(a.N) * a.H * ((a.W + 7) / 8), 256 / G) : 0;                             \
      if (a.pg) hipLaunchKernelGGL((qkv_prep_kernel<true, G, 1, T, true>), dim3(a.nq + nkv), dim3(256), (256 / G) * lds_of, s, a); \
      else hipLaunchKernelGGL((qkv_prep_kernel<true, G, 1, T, false>), dim3(a.nq + nkv), dim3(256), (256 / G) * lds_of, s, a);     \
    } else {                                                                                                         \
      a.nq = a.oq ? row_grid(static_cast<long>(a.N) * a.H * a.W, 256 / G) : 0;                                        \
      if (a.pg) hipLaunchKernelGGL((qkv_prep_kernel<false, G, NV, T, true>), dim3(a.nq + nkv), dim3(256), (256 / G) * lds_of, s, a); \
      else hipLaunchKernelGGL((qkv_prep_kernel<false, G, NV, T, false>), dim3(a.nq + nkv), dim3(256), (256 / G) * lds_of, s, a);     \
    }                                                                                                                \
  } while (0)
  if (!a.oq && a.pg) {
    // pooled branch alone (the query branch lives in block_front): one iteration of a workgroup is a load -> LayerNorm ->
    // accumulate chain, so what matters is bytes in flight.  Three pieces per lane (C / 12 lanes per position) put 4x more
    // positions into one iteration than the one-piece mapping of the row kernels.
    const int c4 = a.C / 4;
#define CALLKV(G)                                                                                                   \
  do {                                                                                                              \
    a.nq = 0;                                                                                                       \
    if (a.pwk)                                                                                                      \
      hipLaunchKernelGGL((qkv_prep_kernel<false, G, 3, T, true, true>), dim3(nkv), dim3(256),                       \
                         (256 / G) * lds_of + 2 * a.C * sizeof(T), s, a);                                           \
    else                                                                                                            \
      hipLaunchKernelGGL((qkv_prep_kernel<false, G, 3, T, true, false>), dim3(nkv), dim3(256), (256 / G) * lds_of, s, a); \
    return check_launch("qkv_prep(kv)");                                                                            \
  } while (0)
    if (c4 == 24) CALLKV(8);
    if (c4 == 48) CALLKV(16);
    if (c4 == 96) CALLKV(32);
    if (c4 == 192) CALLKV(64);
#undef CALLKV
  }
  DS_ROW_DISPATCH(a.C, CALL);
#undef CALL
  return check_launch("qkv_prep");
}

extern "C" int diffsal_qkv_prep(const void* xq, const float* w9, const float* gq, const float* bq, void* out_q, const void* xk,
                                const void* xv, const float* wk, const float* wv, const float* gk, const float* bk,
                                const float* gv, const float* bv, void* out_k, void* out_v, int N, int H, int W, int C, int k,
                                float eps, const float* pre_gamma, const float* pre_beta, float pre_eps, int pre_ln_k, int dtype,
                                diffsal_stream_t stream) {
  DS_REQUIRE((pre_gamma == nullptr) == (pre_beta == nullptr) && (!pre_gamma || (aligned16(pre_gamma) && aligned16(pre_beta))),
             DIFFSAL_E_ARG, "qkv_prep: pre-LayerNorm gamma and beta go together (16-byte aligned)");
  // out_q == nullptr: pooled key / value branch only (the query branch lives in diffsal_block_front)
  DS_REQUIRE((!out_q || (xq && w9 && gq && bq)) && xk && xv && wk && wv && gk && bk && gv && bv && out_k && out_v, DIFFSAL_E_ARG,
             "qkv_prep: null argument");
  DS_REQUIRE(N > 0 && C > 0 && C % 4 == 0 && k > 0 && H >= k && W >= k, DIFFSAL_E_SHAPE,
             "qkv_prep: bad shape H=%d W=%d C=%d k=%d", H, W, C, k);
  DS_REQUIRE((!out_q || (aligned16(xq) && aligned16(out_q) && aligned16(w9) && aligned16(gq) && aligned16(bq))) && aligned16(xk) && aligned16(xv) &&
                 aligned16(wk) && aligned16(wv) && aligned16(out_k) && aligned16(out_v),
             DIFFSAL_E_ALIGN, "qkv_prep: misaligned pointer");
  QkvPrepArgs a{xq, w9, gq, bq, out_q, xk, xv, wk, wv, gk, bk, gv, bv, out_k, out_v, N, H, W, C, k, (H - k) / k + 1, (W - k) / k + 1, 0, eps,
                pre_gamma, pre_beta, pre_eps, pre_ln_k, nullptr, nullptr, nullptr, nullptr};
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALLT(T) return qkv_prep_t<T>(a, s)
  DS_DTYPE_DISPATCH(dtype, "qkv_prep", CALLT);
#undef CALLT
  return DIFFSAL_OK;
}

extern "C" int diffsal_kv_prep_proj(const void* xk, const void* xv, const float* wk, const float* wv, const float* gk,
                                    const float* bk, const float* gv, const float* bv, const void* proj_wk, const float* proj_bk,
                                    const void* proj_wv, const float* proj_bv, void* out_k, void* out_v, int N, int H, int W, int C,
                                    int k, float eps, const float* pre_gamma, const float* pre_beta, float pre_eps, int pre_ln_k,
                                    int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(xk && xv && wk && wv && gk && bk && gv && bv && proj_wk && proj_bk && proj_wv && proj_bv && out_k && out_v && pre_gamma &&
                 pre_beta, DIFFSAL_E_ARG, "kv_prep_proj: null argument");
  DS_REQUIRE(N > 0 && (C == 96 || C == 192) && k > 0 && H >= k && W >= k, DIFFSAL_E_SHAPE,
             "kv_prep_proj: built for C = 96 / 192 (every workgroup reads both [C, C] weights); got H=%d W=%d C=%d k=%d", H, W, C, k);
  DS_REQUIRE(aligned16(xk) && aligned16(xv) && aligned16(wk) && aligned16(wv) && aligned16(out_k) && aligned16(out_v) &&
                 aligned16(proj_wk) && aligned16(proj_wv) && aligned16(pre_gamma) && aligned16(pre_beta),
             DIFFSAL_E_ALIGN, "kv_prep_proj: misaligned pointer");
  QkvPrepArgs a{nullptr, nullptr, nullptr, nullptr, nullptr, xk, xv, wk, wv, gk, bk, gv, bv, out_k, out_v, N, H, W, C, k, (H - k) / k + 1,
                (W - k) / k + 1, 0, eps, pre_gamma, pre_beta, pre_eps, pre_ln_k, proj_wk, proj_bk, proj_wv, proj_bv};
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALLT(T) return qkv_prep_t<T>(a, s)
  DS_DTYPE_DISPATCH(dtype, "kv_prep_proj", CALLT);
#undef CALLT
  return DIFFSAL_OK;
}
