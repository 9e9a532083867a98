// "Convolve at the low resolution, interpolate the taps": a 3x3 convolution applied to a bilinearly up-sampled map (or to a
// sum of up-sampled maps) without ever forming the up-sampled map.
//
//   conv3x3_d( sum_i up_i(z_i) )[p] = sum_i sum_tap up_i( W_tap z_i )[p + d * delta_tap]
//
// because a 1x1 channel mixing (W_tap) commutes with bilinear interpolation (whose weights are per-pixel scalars) and a 3x3
// convolution is nine shifted 1x1 mixings; taps that fall outside the image contribute zero (the convolution's zero padding
// applies to the up-sampled image).  The channel mixings run as ONE GEMM per source at the SOURCE resolution
// (Y_i = z_i [M_i, Cin] x Wcat [9 Cout, Cin]^T, diffsal_conv_igemm): f^2 times fewer rows than the convolution at the target
// resolution -- 4x fewer FLOPs for UpEmbed's first convolution (bilinear x2 then 3x3 dilation 2: R/.../common_block.py:196-216),
// 3x fewer for mt_proj on the 4-scale sum (x2 .. x16: R/.../sal_unet.py:480-489, :407) -- and this kernel gathers:
//
//   out[n, Y, X, c] = act( scale[c] * ( bias[c] + sum_i sum_tap [p' inside] bilerp_i(Y_i[..., tap * C + c]; p') ) + shift[c] ),
//                     p' = (Y, X) + dil * (ky - 1, kx - 1)
//
// One wavefront owns a 4x4 output patch of TWO images (lanes 0-31 image 2k, lanes 32-63 image 2k+1: same coordinates, same
// interpolation weights, so every weight stays a wave-uniform scalar) and a 128-channel slab, 4 channels per lane.  Per tap and
// source the patch needs at most 4 (factor <= 2) or 3 (factor >= 4) source rows / columns, consumed a row at a time as in
// resize_sum_kernel (misc.hip).  Exact up to summation order; fp32 arithmetic on fp32 / bf16 / fp16 storage.
#include "common.h"

namespace diffsal {

struct TapSumArgs {
  const void* in[4];   // Y_i: [N, h_i, w_i, 9 * C] (tap-major channel blocks), storage type of the launch
  int h[4], w[4];
  float sy[4], sx[4];
  int n_in;
  const float* bias;   // per output channel, any of the three may be null
  const float* scale;
  const float* shift;
  int act;
  int N, H, W, C, dil;
  // head form (C <= 128): instead of storing the [N,H,W,C] map, head_out[n,Y,X] = sigmoid(head_b + sum_c head_w[c] out[n,Y,X,c])
  const float* head_w;
  const float* head_b;
  float* head_out;
};

__device__ __forceinline__ float uni_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// acc[d][e] += bilinear sample of `src` (row pitch P channels, this lane's channel offset already applied) at the target
// positions (Ys + d, Xs + e), d, e = 0..3; positions outside [0,H) x [0,W) contribute nothing.
template <typename T>
__device__ __forceinline__ void tap_accumulate(const T* __restrict__ img, int nr, int h, int w, long P, int Ys, int Xs, int H,
                                               int W, float sy, float sx, float4 (&acc)[4][4]) {
  int y0d[4], y1d[4], x0d[4], x1d[4];
  float lyd[4], lxd[4];
  int ry0 = 1 << 30, rx0 = 1 << 30;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int Yp = Ys + d, Xp = Xs + d;
    const bool vy = Yp >= 0 && Yp < H, vx = Xp >= 0 && Xp < W;
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_coord(vy ? Yp : 0, sy, h, y0, y1, ly);
    bilin_coord(vx ? Xp : 0, sx, w, x0, x1, lx);
    y0d[d] = uni_i(vy ? y0 : -(1 << 20)); y1d[d] = uni_i(vy ? y1 : -(1 << 20)); lyd[d] = uni_f(ly);
    x0d[d] = uni_i(vx ? x0 : -(1 << 20)); x1d[d] = uni_i(vx ? x1 : -(1 << 20)); lxd[d] = uni_f(lx);
    if (vy && y0 < ry0) ry0 = y0;
    if (vx && x0 < rx0) rx0 = x0;
  }
  ry0 = uni_i(ry0);
  rx0 = uni_i(rx0);
  if (ry0 == (1 << 30) || rx0 == (1 << 30)) return;   // the whole 4x4 block of positions lies outside the image
  float wx[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      wx[e][k] = (rx0 + k == x0d[e] ? 1.f - lxd[e] : 0.f) + (rx0 + k == x1d[e] ? lxd[e] : 0.f);
#pragma unroll 1
  for (int i = 0; i < nr; ++i) {
    const T* rp = img + static_cast<long>(min(ry0 + i, h - 1)) * w * P;
    float4 row[4];
    row[0] = ld4(rp + static_cast<long>(min(rx0, w - 1)) * P);
    row[1] = ld4(rp + static_cast<long>(min(rx0 + 1, w - 1)) * P);
    row[2] = ld4(rp + static_cast<long>(min(rx0 + 2, w - 1)) * P);
    row[3] = nr > 3 ? ld4(rp + static_cast<long>(min(rx0 + 3, w - 1)) * P) : make_float4(0, 0, 0, 0);
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

template <typename T, bool HEAD = false>
__global__ __launch_bounds__(256) void tapsum_kernel(TapSumArgs a, T* __restrict__ out, int w_patches, int slabs, long n_items) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), and neighbouring patches read
  // the same source rows for every tap.  Workgroup b therefore takes the b/8-th item block of the contiguous run that belongs to
  // XCD b % 8 (bijective for any grid size): an XCD's L2 sees whole bands of one image instead of every eighth patch of all of
  // them.  PMC before: 2.6x the once-through bytes fetched from HBM, at which point the gather was HBM-bound.
  const unsigned nb = gridDim.x, xq = nb >> 3, xr = nb & 7u, xcd = blockIdx.x & 7u;
  const unsigned vb = xcd * xq + (xcd < xr ? xcd : xr) + (blockIdx.x >> 3);
  const long item = static_cast<long>(vb) * 4 + (threadIdx.x >> 6);
  if (item >= n_items) return;
  const int slab = static_cast<int>(item % slabs);
  long t = item / slabs;
  const int h_patches = (a.H + 3) >> 2;
  const int px = static_cast<int>(t % w_patches); t /= w_patches;
  const int py = static_cast<int>(t % h_patches);
  const int pair = static_cast<int>(t / h_patches);
  const int Y0 = uni_i(py * 4), X0 = uni_i(px * 4);
  const int n = pair * 2 + half, c = slab * 128 + l32 * 4;
  const bool live = n < a.N && c < a.C;
  const int nc = n < a.N ? n : a.N - 1, cc = c < a.C ? c : a.C - 4;   // dead lanes walk valid memory and store nothing
  const long P = 9L * a.C;
  float4 acc[4][4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[d][e] = make_float4(0, 0, 0, 0);
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int Ys = Y0 + a.dil * (ky - 1), Xs = X0 + a.dil * (kx - 1);
#pragma unroll 1
    for (int s = 0; s < a.n_in; ++s) {
      const int hs = a.h[s], ws = a.w[s];
      const int f = a.H / hs;
      const T* img = static_cast<const T*>(a.in[s]) + static_cast<long>(nc) * hs * ws * P + static_cast<long>(tap) * a.C + cc;
      tap_accumulate<T>(img, f <= 2 ? 4 : 3, hs, ws, P, Ys, Xs, a.H, a.W, a.sy[s], a.sx[s], acc);
    }
  }
  if (!HEAD && !live) return;
  float4 bi = make_float4(0, 0, 0, 0), sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
  if (a.bias) bi = ld4(a.bias + cc);
  if (a.scale) sc = ld4(a.scale + cc);
  if (a.shift) sh = ld4(a.shift + cc);
  float4 hw = make_float4(0, 0, 0, 0);
  if constexpr (HEAD) { if (c < a.C) hw = ld4(a.head_w + c); }     // lanes beyond C contribute nothing to the pixel's dot product
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float4 v = acc[d][e];
      v.x = (v.x + bi.x) * sc.x + sh.x; v.y = (v.y + bi.y) * sc.y + sh.y;
      v.z = (v.z + bi.z) * sc.z + sh.z; v.w = (v.w + bi.w) * sc.w + sh.w;
      if (a.act == DIFFSAL_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if constexpr (HEAD) {
        // MLPHead (R/models/saliency_decoder/common_block.py:111-122): 1x1 convolution to one channel + sigmoid, over the 32
        // lanes of this image's half-wave; the C-channel map itself is never stored
        float t = fmaf(v.x, hw.x, fmaf(v.y, hw.y, fmaf(v.z, hw.z, v.w * hw.w)));
        t = group_sum<32>(t);
        if (l32 == 0 && n < a.N && Y0 + d < a.H && X0 + e < a.W)
          a.head_out[(static_cast<long>(n) * a.H + Y0 + d) * a.W + X0 + e] = sigmoidf_(t + a.head_b[0]);
      } else {
        if (Y0 + d < a.H && X0 + e < a.W)       // ragged last patch row / column when H or W is not a multiple of 4
          st4(out + ((static_cast<long>(n) * a.H + Y0 + d) * a.W + X0 + e) * a.C + c, v);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Adjoint of tapsum with respect to one source's tap products (training): dY[n, iy, ix, tap, :] =
//   sum_{Y, X} wy(Y -> iy) wx(X -> ix) dU[n, Y - dil (ky-1), X - dil (kx-1), :]   (terms outside the image drop out),
// gather form, deterministic.  Separable like the plain resize adjoint (backward.hip): a row pass makes the three
// ky-shifted row reductions R[ky] [N][h][W*C], a column pass the nine (ky, kx) outputs.  blockIdx.y = ky (rows) / tap (columns).
// ------------------------------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(256) void tapsum_bwd_axis_kernel(const float* __restrict__ in, float* __restrict__ out, long outer,
                                                              int L, int l, int inner, float scale, int dil, long slab) {
  const int j = blockIdx.y;
  const int shift = dil * ((PASS == 0 ? j : j % 3) - 1);
  const float* src = in + (PASS == 0 ? 0L : static_cast<long>(j / 3) * slab);
  float* dst = out + (PASS == 0 ? static_cast<long>(j) * slab : static_cast<long>(j) * inner);
  const long out_pitch = PASS == 0 ? inner : 9L * inner;
  const int cv = inner / 4;
  const long total = outer * l * cv;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % cv) * 4;
    long r = i / cv;
    const int il = static_cast<int>(r % l);
    const long o = r / l;
    const int lo = max(0, static_cast<int>(floorf((il - 0.5f) / scale - 0.5f)) - 1);
    const int hi = min(L - 1, static_cast<int>(ceilf((il + 1.5f) / scale - 0.5f)) + 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int Y = lo; Y <= hi; ++Y) {
      int y0, y1; float ly;
      bilin_coord(Y, scale, l, y0, y1, ly);
      const float wy = (y0 == il ? 1.f - ly : 0.f) + (y1 == il ? ly : 0.f);
      const int Ys = Y - shift;
      if (wy == 0.f || Ys < 0 || Ys >= L) continue;
      const float4 v = ld4(src + (o * L + Ys) * inner + c);
      acc.x = fmaf(wy, v.x, acc.x); acc.y = fmaf(wy, v.y, acc.y);
      acc.z = fmaf(wy, v.z, acc.z); acc.w = fmaf(wy, v.w, acc.w);
    }
    st4(dst + (o * l + il) * out_pitch + c, acc);
  }
}

}  // namespace diffsal

using namespace diffsal;

static int tapsum_impl(const void* const* srcs, const int* hs, const int* ws, int n_src, void* out, int N, int H, int W, int C,
                       int dil, const float* bias, const float* scale, const float* shift, int act, int dtype,
                       const float* head_w, const float* head_b, float* head_out, diffsal_stream_t stream) {
  DS_REQUIRE(srcs && hs && ws && (out || head_out), DIFFSAL_E_ARG, "tapsum: null argument");
  DS_REQUIRE(!head_out || (head_w && head_b && C <= 128 && aligned16(head_w)), DIFFSAL_E_ARG,
             "tapsum: the head form needs its weight (16-byte aligned), its bias and C <= 128");
  DS_REQUIRE(n_src >= 1 && n_src <= 4 && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 &&
                 (dil == 1 || dil == 2) && (act == DIFFSAL_ACT_NONE || act == DIFFSAL_ACT_RELU),
             DIFFSAL_E_SHAPE, "tapsum: bad shape n_src=%d N=%d H=%d W=%d C=%d dil=%d act=%d", n_src, N, H, W, C, dil, act);
  DS_REQUIRE((scale == nullptr) == (shift == nullptr), DIFFSAL_E_ARG, "tapsum: scale and shift go together");
  TapSumArgs a;
  a.n_in = n_src;
  for (int i = 0; i < 4; ++i) {
    a.in[i] = i < n_src ? srcs[i] : nullptr;
    a.h[i] = i < n_src ? hs[i] : 1;
    a.w[i] = i < n_src ? ws[i] : 1;
    a.sy[i] = static_cast<float>(a.h[i]) / static_cast<float>(H);
    a.sx[i] = static_cast<float>(a.w[i]) / static_cast<float>(W);
    if (i < n_src) {
      DS_REQUIRE(srcs[i] && aligned16(srcs[i]) && hs[i] > 0 && ws[i] > 0, DIFFSAL_E_ARG, "tapsum: bad source %d", i);
      const int f = H / hs[i];
      DS_REQUIRE(f >= 1 && (f & (f - 1)) == 0 && hs[i] * f == H && ws[i] * f == W && hs[i] >= 2 && ws[i] >= 2, DIFFSAL_E_SHAPE,
                 "tapsum: source %d (%dx%d) must be a power-of-two factor smaller than %dx%d", i, hs[i], ws[i], H, W);
    }
  }
  a.bias = bias; a.scale = scale; a.shift = shift; a.act = act;
  a.N = N; a.H = H; a.W = W; a.C = C; a.dil = dil;
  a.head_w = head_w; a.head_b = head_b; a.head_out = head_out;
  DS_REQUIRE((!out || aligned16(out)) && (!bias || aligned16(bias)) && (!scale || (aligned16(scale) && aligned16(shift))), DIFFSAL_E_ALIGN,
             "tapsum: misaligned pointer");
  const int slabs = (C + 127) / 128;
  const long n_items = static_cast<long>((N + 1) / 2) * ((H + 3) / 4) * ((W + 3) / 4) * slabs;
  DS_REQUIRE((n_items + 3) / 4 < (1L << 31), DIFFSAL_E_SHAPE, "tapsum: output too large");
#define CALL(T)                                                                                                          \
  do {                                                                                                                   \
    if (head_out)                                                                                                        \
      hipLaunchKernelGGL((tapsum_kernel<T, true>), dim3(static_cast<unsigned>((n_items + 3) / 4)), dim3(256), 0,          \
                         static_cast<hipStream_t>(stream), a, static_cast<T*>(nullptr), (W + 3) / 4, slabs, n_items);    \
    else                                                                                                                 \
      hipLaunchKernelGGL((tapsum_kernel<T, false>), dim3(static_cast<unsigned>((n_items + 3) / 4)), dim3(256), 0,         \
                         static_cast<hipStream_t>(stream), a, static_cast<T*>(out), (W + 3) / 4, slabs, n_items);        \
  } while (0)
  DS_DTYPE_DISPATCH(dtype, "tapsum", CALL);
#undef CALL
  return check_launch("tapsum");
}

extern "C" int diffsal_tapsum(const void* const* srcs, const int* hs, const int* ws, int n_src, void* out, int N, int H, int W,
                              int C, int dil, const float* bias, const float* scale, const float* shift, int act, int dtype,
                              diffsal_stream_t stream) {
  DS_REQUIRE(out, DIFFSAL_E_ARG, "tapsum: null output");
  return tapsum_impl(srcs, hs, ws, n_src, out, N, H, W, C, dil, bias, scale, shift, act, dtype, nullptr, nullptr, nullptr, stream);
}

extern "C" int diffsal_tapsum_head(const void* const* srcs, const int* hs, const int* ws, int n_src, int N, int H, int W, int C,
                                   int dil, const float* bias, const float* scale, const float* shift, int act,
                                   const float* head_w, const float* head_b, float* head_out, int dtype,
                                   diffsal_stream_t stream) {
  DS_REQUIRE(head_out, DIFFSAL_E_ARG, "tapsum_head: null output");
  return tapsum_impl(srcs, hs, ws, n_src, nullptr, N, H, W, C, dil, bias, scale, shift, act, dtype, head_w, head_b, head_out, stream);
}

extern "C" long diffsal_tapsum_bwd_ws_bytes(int N, int W, int C, int h) {
  return 3L * N * h * W * C * static_cast<long>(sizeof(float));
}

extern "C" int diffsal_tapsum_bwd(const float* du, float* dy, float* ws, int N, int H, int W, int C, int h, int w, int dil,
                                  diffsal_stream_t stream) {
  DS_REQUIRE(du && dy && ws, DIFFSAL_E_ARG, "tapsum_bwd: null argument");
  DS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && h >= 2 && w >= 2 && h <= H && w <= W && (dil == 1 || dil == 2),
             DIFFSAL_E_SHAPE, "tapsum_bwd: bad shape N=%d H=%d W=%d C=%d h=%d w=%d dil=%d", N, H, W, C, h, w, dil);
  DS_REQUIRE(aligned16(du) && aligned16(dy) && aligned16(ws), DIFFSAL_E_ALIGN, "tapsum_bwd: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long slab = static_cast<long>(N) * h * W * C;
  const long rows = static_cast<long>(N) * h * (static_cast<long>(W) * C / 4);
  const long cols = static_cast<long>(N) * h * w * (C / 4);
  const unsigned gr = static_cast<unsigned>(std::min<long>((rows + 255) / 256, 1L << 20));
  const unsigned gc = static_cast<unsigned>(std::min<long>((cols + 255) / 256, 1L << 20));
  hipLaunchKernelGGL((tapsum_bwd_axis_kernel<0>), dim3(gr, 3), dim3(256), 0, s, du, ws, static_cast<long>(N), H, h, W * C,
                     static_cast<float>(h) / static_cast<float>(H), dil, slab);
  hipLaunchKernelGGL((tapsum_bwd_axis_kernel<1>), dim3(gc, 9), dim3(256), 0, s, ws, dy, static_cast<long>(N) * h, W, w, C,
                     static_cast<float>(w) / static_cast<float>(W), dil, slab);
  return check_launch("tapsum_bwd");
}
