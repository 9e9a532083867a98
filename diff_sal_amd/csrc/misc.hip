// Small / HBM-bound kernels of the SalUNet step: timestep-embedding MLP, conv_in, frame packing,
// bilinear resizes, audio fusion, the pooled-KV attention core, the sigmoid head and the sampler axpy.
#include "common.h"

namespace diffsal {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------------
// K1: sinusoidal embedding + dense0 + swish + dense1, one workgroup per batch element.
// R/models/saliency_decoder/sal_unet.py:15-33, :304-307.  A wavefront computes one output row at a
// time: coalesced weight-row read + 64-lane butterfly.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void temb_kernel(const void* __restrict__ t, int t_is_f32, int ch,
                                                   const float* __restrict__ freq, const float* __restrict__ w0,
                                                   const float* __restrict__ b0, const float* __restrict__ w1,
                                                   const float* __restrict__ b1, float* __restrict__ out) {
  extern __shared__ float sh[];  // emb[ch] | hidden[4ch]
  float* emb = sh;
  float* hid = sh + ch;
  const int b = blockIdx.x;
  const int half = ch / 2, tc = 4 * ch;
  const float tv = t_is_f32 ? static_cast<const float*>(t)[b]
                            : static_cast<float>(static_cast<const long long*>(t)[b]);
  for (int j = threadIdx.x; j < half; j += 256) {
    const float a = tv * freq[j];
    emb[j] = sinf(a);
    emb[half + j] = cosf(a);
  }
  if ((ch & 1) && threadIdx.x == 0) emb[ch - 1] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < tc; r += 4) {
    float s = 0.f;
    for (int k = lane; k < ch; k += 64) s = fmaf(w0[r * ch + k], emb[k], s);
    s = group_sum<64>(s);
    if (lane == 0) hid[r] = swishf(s + b0[r]);
  }
  __syncthreads();
  for (int r = wave; r < tc; r += 4) {
    float s = 0.f;
    for (int k = lane; k < tc; k += 64) s = fmaf(w1[static_cast<long>(r) * tc + k], hid[k], s);
    s = group_sum<64>(s);
    if (lane == 0) out[static_cast<long>(b) * tc + r] = s + b1[r];
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

// ------------------------------------------------------------------------------------------------
// K2: conv_in 1 -> C, 3x3, pad 1; NCHW (C=1) in, NHWC out.  R/.../sal_unet.py:240,292
// thread = (pixel, 4 channels).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int B,
                                                      int H, int W, int C, int skip_mod) {
  const int c4n = C >> 2;
  const long total = static_cast<long>(B) * H * W * c4n;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    const long pix = i / c4n;
    const int xw = static_cast<int>(pix % W);
    const int yh = static_cast<int>((pix / W) % H);
    if (skip_mod > 0 && ((yh % skip_mod) == skip_mod - 1 || (xw % skip_mod) == skip_mod - 1)) continue;
    const float* img = x + (pix / (static_cast<long>(H) * W)) * H * W;
    float o[4] = {bias[c], bias[c + 1], bias[c + 2], bias[c + 3]};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = yh + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = xw + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const float v = img[static_cast<long>(iy) * W + ix];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(v, w[(c + j) * 9 + ky * 3 + kx], o[j]);
      }
    }
    st4(out + pix * C + c, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// K6: NCTHW -> frames-of-NHWC transpose through a 64x65 LDS tile, plus the noise map as last frame.
// vis [B][C][Q], Q = Tv*hw  ->  out [B][Tout*hw][C] (first Q rows); noise [B][hw][C] -> rows Tv*hw...
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_frames_kernel(const float* __restrict__ vis, const float* __restrict__ noise,
                                                          float* __restrict__ out, int C, int Q, int hw, int Tout,
                                                          int tiles_q, int tiles_c, int n_transpose_blocks) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y;
  const long out_b = static_cast<long>(b) * Tout * hw * C;
  if (static_cast<int>(blockIdx.x) < n_transpose_blocks) {
    const int tq = blockIdx.x % tiles_q, tcx = blockIdx.x / tiles_q;
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
      if (q < Q && c < C) out[out_b + static_cast<long>(q) * C + c] = tile[tx][r];
    }
  } else if (noise) {
    const int nb = gridDim.x - n_transpose_blocks;
    const long n4 = static_cast<long>(hw) * C / 4;
    const float* src = noise + static_cast<long>(b) * hw * C;
    float* dst = out + out_b + static_cast<long>(Q) * C;
    for (long i = static_cast<long>(blockIdx.x - n_transpose_blocks) * 256 + threadIdx.x; i < n4;
         i += static_cast<long>(nb) * 256)
      st4(dst + i * 4, ld4(src + i * 4));
  }
}

// ------------------------------------------------------------------------------------------------
// bilinear resize (align_corners=False) on NHWC.  R/.../common_block.py:197; sal_unet.py:325-327,482-484
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int h,
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
    const float* b = in + static_cast<long>(n) * h * w * C + c;
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
      const float v00 = b[(static_cast<long>(y0) * w + x0) * C], v01 = b[(static_cast<long>(y0) * w + x1) * C];
      const float v10 = b[(static_cast<long>(y1) * w + x0) * C], v11 = b[(static_cast<long>(y1) * w + x1) * C];
      out[o] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
  }
}

struct ResizeSumArgs {
  const float* in[4];
  int h[4], w[4];
  float sy[4], sx[4];
  int n_in;
};

// out = ((in0^ + in1^) + in2^) + in3^, x^ = bilinear resize of x to (H, W): one write of the big map.
__global__ __launch_bounds__(256) void resize_sum_kernel(ResizeSumArgs a, float* __restrict__ out, int N, int H, int W,
                                                         int C) {
  const int cv = C >> 2;
  const long total = static_cast<long>(N) * H * W * cv;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % cv) * 4;
    long pix = i / cv;
    const int X = static_cast<int>(pix % W);
    pix /= W;
    const int Y = static_cast<int>(pix % H);
    const int n = static_cast<int>(pix / H);
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
        const float* b = a.in[s] + static_cast<long>(n) * h * w * C + c;
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
__global__ __launch_bounds__(256) void audio_fuse_kernel(const float* __restrict__ a_small, const float* __restrict__ x,
                                                         float* __restrict__ out, int T, int H, int W, int C, int h, int w,
                                                         int up) {
  extern __shared__ float sh[];  // s[32][W+1]
  const int WP = W + 1;
  const int cslabs = C / 32;
  int bid = blockIdx.x;
  const int cs = bid % cslabs; bid /= cslabs;
  const int y = bid % H;
  const int b = bid / H;
  const int cl = threadIdx.x & 31;
  const int xl = threadIdx.x >> 5;  // 0..7
  const int c = cs * 32 + cl;
  const int ys = y / up;
  // pass 1: m[c][x]
  for (int xx = xl; xx < W; xx += 8) {
    const int xs = xx / up;
    float s = 0.f;
    for (int t = 0; t < T; ++t) {
      const float av = a_small[((static_cast<long>(b) * T + t) * h * w + ys * w + xs) * C + c];
      const float xv = x[(((static_cast<long>(b) * T + t) * H + y) * W + xx) * C + c];
      s = fmaf(av, xv, s);
    }
    sh[cl * WP + xx] = s / static_cast<float>(T);
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
  const int total = 32 * T * W;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int xx = i % W;
    const int t = (i / W) % T;
    const int cc = i / (W * T);
    const int cg = cs * 32 + cc;
    const float av = a_small[((static_cast<long>(b) * T + t) * h * w + ys * w + xx / up) * C + cg];
    out[(((static_cast<long>(b) * C + cg) * T + t) * H + y) * W + xx] = av * sh[cc * WP + xx];
  }
}

// ------------------------------------------------------------------------------------------------
// K11: attention core with a short pooled K/V (Lk <= 32).  R/.../attention.py:97-108
// One workgroup = one image n and a run of queries; K and V rows of that image live in LDS.
// One wavefront per query: lanes stride the C channels; per (head, key) partial dot products are
// reduced with 64-lane butterflies; softmax over Lk in registers (every lane holds all scores).
// ------------------------------------------------------------------------------------------------
template <int LK_MAX>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ o, int Lq,
                                                        int Lk, int C, int heads, float scale, int q_per_block) {
  extern __shared__ float sh[];  // K[Lk][C] | V[Lk][C]
  float* Ks = sh;
  float* Vs = sh + Lk * C;
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < Lk * C / 4; i += 256) {
    st4(Ks + i * 4, ld4(k + static_cast<long>(n) * Lk * C + i * 4));
    st4(Vs + i * 4, ld4(v + static_cast<long>(n) * Lk * C + i * 4));
  }
  __syncthreads();
  const int d = C / heads;
  const int q_begin = blockIdx.x * q_per_block;
  const int q_end = min(Lq, q_begin + q_per_block);
  for (int l = q_begin + wave; l < q_end; l += 4) {
    const float* qr = q + (static_cast<long>(n) * Lq + l) * C;
    float* orow = o + (static_cast<long>(n) * Lq + l) * C;
    for (int hd = 0; hd < heads; ++hd) {
      const int cb = hd * d;
      float sc[LK_MAX];
#pragma unroll
      for (int t = 0; t < LK_MAX; ++t) sc[t] = 0.f;
      for (int cc = lane; cc < d; cc += 64) {
        const float qv = qr[cb + cc];
#pragma unroll
        for (int t = 0; t < LK_MAX; ++t)
          if (t < Lk) sc[t] = fmaf(qv, Ks[t * C + cb + cc], sc[t]);
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int t = 0; t < LK_MAX; ++t)
        if (t < Lk) { sc[t] = group_sum<64>(sc[t]) * scale; mx = fmaxf(mx, sc[t]); }
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < LK_MAX; ++t)
        if (t < Lk) { sc[t] = expf(sc[t] - mx); sum += sc[t]; }
      const float inv = 1.0f / sum;
      for (int cc = lane; cc < d; cc += 64) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < LK_MAX; ++t)
          if (t < Lk) acc = fmaf(sc[t] * inv, Vs[t * C + cb + cc], acc);
        orow[cb + cc] = acc;
      }
    }
  }
}

// K14 tail: per-pixel dot with w[C] + sigmoid; G lanes per pixel.
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, const float* __restrict__ w,
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

__global__ __launch_bounds__(256) void axpbypcz_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ z, float a, float b, float c,
                                                       float* __restrict__ out, size_t n) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) {
    float r = a * x[i];
    if (y) r += b * y[i];
    if (z) r += c * z[i];
    out[i] = r;
  }
}

static int ew_grid(long total_threads) {
  long g = (total_threads + 255) / 256;
  return static_cast<int>(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_version(void) { return 1; }
extern "C" const char* diffsal_last_error(void) { return g_err; }

extern "C" int diffsal_temb_mlp(const void* t, int t_is_f32, int B, int ch, const float* freq, const float* w0,
                                const float* b0, const float* w1, const float* b1, float* temb_out,
                                diffsal_stream_t stream) {
  DS_REQUIRE(t && freq && w0 && b0 && w1 && b1 && temb_out, DIFFSAL_E_ARG, "temb_mlp: null argument");
  DS_REQUIRE(B > 0 && ch >= 4 && ch <= 1024, DIFFSAL_E_SHAPE, "temb_mlp: bad shape B=%d ch=%d", B, ch);
  hipLaunchKernelGGL(temb_kernel, dim3(B), dim3(256), 5 * ch * sizeof(float), static_cast<hipStream_t>(stream), t,
                     t_is_f32, ch, freq, w0, b0, w1, b1, temb_out);
  return check_launch("temb_mlp");
}

extern "C" int diffsal_dense_small(const float* in, int B, int K, int swish_in, const float* w, const float* bias,
                                   int N, float* out, diffsal_stream_t stream) {
  DS_REQUIRE(in && w && out, DIFFSAL_E_ARG, "dense_small: null argument");
  DS_REQUIRE(B > 0 && K > 0 && N > 0 && static_cast<long>(B) * K * 4 <= 64 * 1024, DIFFSAL_E_SHAPE,
             "dense_small: bad shape B=%d K=%d N=%d", B, K, N);
  hipLaunchKernelGGL(dense_small_kernel, dim3((N + 3) / 4), dim3(256), static_cast<size_t>(B) * K * sizeof(float),
                     static_cast<hipStream_t>(stream), in, B, K, swish_in, w, bias, N, out);
  return check_launch("dense_small");
}

extern "C" int diffsal_conv_in(const float* x, const float* w, const float* bias, float* out, int B, int H, int W,
                               int C, int skip_mod, diffsal_stream_t stream) {
  DS_REQUIRE(x && w && bias && out, DIFFSAL_E_ARG, "conv_in: null argument");
  DS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "conv_in: bad shape");
  DS_REQUIRE(aligned16(out), DIFFSAL_E_ALIGN, "conv_in: misaligned output");
  const long total = static_cast<long>(B) * H * W * (C / 4);
  hipLaunchKernelGGL(conv_in_kernel, dim3(ew_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, bias,
                     out, B, H, W, C, skip_mod);
  return check_launch("conv_in");
}

extern "C" int diffsal_pack_frames(const float* vis, const float* noise, float* out, int B, int C, int Tv, int Tout,
                                   int hw, diffsal_stream_t stream) {
  DS_REQUIRE(vis && out, DIFFSAL_E_ARG, "pack_frames: null argument");
  DS_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && Tv > 0 && hw > 0 && Tout >= Tv + (noise ? 1 : 0), DIFFSAL_E_SHAPE,
             "pack_frames: bad shape C=%d Tv=%d Tout=%d hw=%d", C, Tv, Tout, hw);
  DS_REQUIRE(aligned16(out) && (!noise || aligned16(noise)), DIFFSAL_E_ALIGN, "pack_frames: misaligned pointer");
  const int Q = Tv * hw;
  const int tiles_q = (Q + 63) / 64, tiles_c = (C + 63) / 64;
  const int ntb = tiles_q * tiles_c;
  const int copy_blocks = noise ? ew_grid(static_cast<long>(hw) * C / 4) : 0;
  hipLaunchKernelGGL(pack_frames_kernel, dim3(ntb + copy_blocks, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     vis, noise, out, C, Q, hw, Tout, tiles_q, tiles_c, ntb);
  return check_launch("pack_frames");
}

extern "C" int diffsal_resize_bilinear(const float* in, float* out, int N, int h, int w, int H, int W, int C,
                                       diffsal_stream_t stream) {
  DS_REQUIRE(in && out, DIFFSAL_E_ARG, "resize_bilinear: null argument");
  DS_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, DIFFSAL_E_SHAPE, "resize_bilinear: bad shape");
  const float sy = static_cast<float>(h) / static_cast<float>(H), sx = static_cast<float>(w) / static_cast<float>(W);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (C % 4 == 0 && aligned16(in) && aligned16(out)) {
    const long total = static_cast<long>(N) * H * W * (C / 4);
    hipLaunchKernelGGL((resize_kernel<4>), dim3(ew_grid(total)), dim3(256), 0, s, in, out, N, h, w, H, W, C, sy, sx);
  } else {
    const long total = static_cast<long>(N) * H * W * C;
    hipLaunchKernelGGL((resize_kernel<1>), dim3(ew_grid(total)), dim3(256), 0, s, in, out, N, h, w, H, W, C, sy, sx);
  }
  return check_launch("resize_bilinear");
}

extern "C" int diffsal_resize_sum(const float* const* ins, const int* hs, const int* ws, int n_in, float* out, int N,
                                  int H, int W, int C, diffsal_stream_t stream) {
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
  const long total = static_cast<long>(N) * H * W * (C / 4);
  hipLaunchKernelGGL(resize_sum_kernel, dim3(ew_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), a, out,
                     N, H, W, C);
  return check_launch("resize_sum");
}

extern "C" int diffsal_audio_fuse(const float* a_small, const float* x, float* out, int B, int T, int H, int W, int C,
                                  int h, int w, diffsal_stream_t stream) {
  DS_REQUIRE(a_small && x && out, DIFFSAL_E_ARG, "audio_fuse: null argument");
  DS_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && h > 0 && w > 0, DIFFSAL_E_SHAPE,
             "audio_fuse: bad shape C=%d", C);
  int up = 1;
  if (h != H && w != W) {  // quirk Q3: upsample only when BOTH differ; factor H // h
    up = H / h;
    DS_REQUIRE(up >= 1 && h * up == H && w * up == W, DIFFSAL_E_SHAPE,
               "audio_fuse: audio map %dx%d does not upsample to %dx%d by an integer factor", h, w, H, W);
  } else {
    DS_REQUIRE(h == H && w == W, DIFFSAL_E_SHAPE, "audio_fuse: audio map %dx%d incompatible with %dx%d", h, w, H, W);
  }
  const size_t lds = static_cast<size_t>(32) * (W + 1) * sizeof(float);
  hipLaunchKernelGGL(audio_fuse_kernel, dim3(B * H * (C / 32)), dim3(256), lds, static_cast<hipStream_t>(stream),
                     a_small, x, out, T, H, W, C, h, w, up);
  return check_launch("audio_fuse");
}

extern "C" int diffsal_attention(const float* q, const float* k, const float* v, float* o, int N, int Lq, int Lk,
                                 int C, int heads, float scale, diffsal_stream_t stream) {
  DS_REQUIRE(q && k && v && o, DIFFSAL_E_ARG, "attention: null argument");
  DS_REQUIRE(N > 0 && Lq > 0 && Lk > 0 && Lk <= 32 && heads > 0 && C % heads == 0 && C % 4 == 0, DIFFSAL_E_SHAPE,
             "attention: bad shape Lq=%d Lk=%d C=%d heads=%d", Lq, Lk, C, heads);
  const size_t lds = static_cast<size_t>(2) * Lk * C * sizeof(float);
  DS_REQUIRE(lds <= 160 * 1024, DIFFSAL_E_SHAPE, "attention: K/V tile (%zu B) exceeds LDS", lds);
  DS_REQUIRE(aligned16(k) && aligned16(v), DIFFSAL_E_ALIGN, "attention: misaligned K/V");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // enough workgroups to fill the chip, but >= 16 queries each so the K/V staging amortises
  int qpb = (Lq * N + 1023) / 1024;
  qpb = qpb < 16 ? 16 : qpb;
  qpb = (qpb + 3) & ~3;
  const dim3 grid((Lq + qpb - 1) / qpb, N);
  if (lds > 64 * 1024) {  // opt in to > 64 KiB of dynamic LDS (host-side attribute, not a stream op)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<18>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (Lk <= 18)
    hipLaunchKernelGGL((attention_kernel<18>), grid, dim3(256), lds, s, q, k, v, o, Lq, Lk, C, heads, scale, qpb);
  else
    hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(256), lds, s, q, k, v, o, Lq, Lk, C, heads, scale, qpb);
  return check_launch("attention");
}

extern "C" int diffsal_head_sigmoid(const float* x, const float* w, const float* bias, float* out, int NHW, int C,
                                    diffsal_stream_t stream) {
  DS_REQUIRE(x && w && bias && out, DIFFSAL_E_ARG, "head_sigmoid: null argument");
  DS_REQUIRE(NHW > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "head_sigmoid: bad shape");
  DS_REQUIRE(aligned16(x) && aligned16(w), DIFFSAL_E_ALIGN, "head_sigmoid: misaligned pointer");
  long g = (static_cast<long>(NHW) + 31) / 32;
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(head_kernel, dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, bias,
                     out, static_cast<long>(NHW), C);
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
