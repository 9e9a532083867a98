// conv3x3_dilation2(bilinear_up2(z)) from a 3x3 convolution at the SOURCE resolution (UpEmbed's first convolution,
// R/models/saliency_decoder/common_block.py:196-206: nn.Upsample(x2, bilinear, align_corners=False) -> Conv2d(3x3, padding 2,
// dilation 2, no bias) -> BatchNorm2d -> ReLU).
//
// A shift by 2 output pixels is a shift by 1 source pixel, so away from the border the convolution commutes with the
// interpolation.  Exactly (per axis, n source pixels, z zero outside [0, n)):
//
//   I0 f (2m) = 0.75 f[m] + 0.25 f[m-1],  I0 f (2m+1) = 0.75 f[m] + 0.25 f[m+1]        interpolation WITHOUT clamping
//   up2(z) = I0 z + D z on [0, 2n), 0 outside;   D z = 0.25 z[0] (delta_0 - delta_-1) + 0.25 z[n-1] (delta_2n-1 - delta_2n)
//
//   conv(up2 z)(p) = I0_y I0_x c (p)  +  sum_taps w_tap [ D_y I0_x + I0_y D_x + D_y D_x ] z (p + 2 tap)
//
// with c = conv3x3(z) (dilation 1, zero padding) evaluated on the grid EXTENDED by one pixel (y in [-1, h], x in [-1, w]) -- the
// F(4x4) Winograd path computes it (csrc/wino4.hip, extended-grid form) with 4x fewer products than the nine tap mixings of
// csrc/tapsum.hip.  The correction terms are non-zero only for outputs p in {0, 1, 2} or {2n-3, 2n-2, 2n-1} along an axis and
// involve z only through the nine tap products T = W_tap z of the BORDER rows and columns of z (a product with ~10 % of the rows).
// This kernel: the interpolation of c, the corrections on the border ring, BatchNorm affine + activation.  Exact up to summation
// order (tests: 1e-5 against F.interpolate + F.conv2d).
#include "common.h"

namespace diffsal {

// T: storage type of c, the tap products and the output (fp32, or bf16 / f16 storage with fp32 arithmetic in between)
template <typename T>
struct UpCommuteArgs {
  const T* c;          // [N][h + 2][w + 2][C]: conv3x3(z) on the extended grid
  const T* tb;     // [N][2 w + 2 h - 4][9][C]: tap products of the border pixels: top row, bottom row, then the left and the
                       // right column WITHOUT their corner pixels (rows 1 .. h - 2)
  const float* scale;  // BatchNorm affine (may be null)
  const float* shift;
  T* out;              // [N][2 h][2 w][C]
  int N, h, w, C, act;
};

// (coefficient, 0 = first / 1 = last border line) of D at high-resolution index q along an axis of n source pixels; 0 if none
__device__ __forceinline__ float upc_dcoef(int q, int n, int& which) {
  which = q >= n ? 1 : 0;          // q in {2n-1, 2n} -> last line
  if (q == 0 || q == 2 * n - 1) return 0.25f;
  if (q == -1 || q == 2 * n) return -0.25f;
  return 0.f;
}

// the two source indices and weights of I0 at high-resolution index q (zero-extended signal of n entries: the caller masks)
__device__ __forceinline__ void upc_i0(int q, int& m0, int& m1) {
  const int m = q >> 1;            // floor (q may be -1)
  m0 = m;
  m1 = (q & 1) ? m + 1 : m - 1;
}

template <typename T>
__device__ __forceinline__ void upc_finish(const UpCommuteArgs<T>& p, float (&v)[4], const float4& sc, const float4& sh, T* dst) {
  v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
  if (p.act == DIFFSAL_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  st4(dst, make_float4(v[0], v[1], v[2], v[3]));
}

// Interior: item = (source pixel (y, x), channel quad): the 3 x 3 neighbourhood of c around it gives the 2 x 2 output pixels
// (2y + sy, 2x + sx) (nine 16-byte loads for four outputs); outputs on the border ring are left to the ring kernel.  One
// workgroup row = one source row of one image; 32-bit index arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void up2_conv_commute_kernel(UpCommuteArgs<T> p) {
  const int c4n = p.C >> 2;
  const int H2 = 2 * p.h, W2 = 2 * p.w;
  const int row_items = p.w * c4n;
  const long crow = static_cast<long>(p.w + 2) * p.C;
  // rows (image, source row) in a grid-stride loop: N * h may exceed the 65535 workgroups of grid dimension y
  for (long row = blockIdx.y; row < static_cast<long>(p.N) * p.h; row += gridDim.y) {
  const int n = static_cast<int>(row / p.h), y = static_cast<int>(row - static_cast<long>(n) * p.h);
  const T* cimg = p.c + static_cast<long>(n) * (p.h + 2) * crow;
  T* oimg = p.out + static_cast<long>(n) * H2 * W2 * p.C;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < row_items; it += gridDim.x * 256) {
    const int x = it / c4n, co = (it - x * c4n) * 4;
    float4 cc[3][3];                 // c on rows y - 1 .. y + 1, columns x - 1 .. x + 1 (extended grid: + 1)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) cc[a][b] = ld4(cimg + (y + a) * crow + static_cast<long>(x + b) * p.C + co);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) { sc = ld4(p.scale + co); sh = ld4(p.shift + co); }
#pragma unroll
    for (int sy = 0; sy < 2; ++sy)
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        const int py = 2 * y + sy, px = 2 * x + sx;
        if (py < 3 || py >= H2 - 3 || px < 3 || px >= W2 - 3) continue;
        // I0: the near sample (weight 0.75) is the centre, the far one (0.25) the neighbour on the side of the parity
        const int ya = sy ? 2 : 0, xa = sx ? 2 : 0;
        const float4 a00 = cc[1][1], a01 = cc[1][xa], a10 = cc[ya][1], a11 = cc[ya][xa];
        float v[4];
        v[0] = 0.75f * (0.75f * a00.x + 0.25f * a01.x) + 0.25f * (0.75f * a10.x + 0.25f * a11.x);
        v[1] = 0.75f * (0.75f * a00.y + 0.25f * a01.y) + 0.25f * (0.75f * a10.y + 0.25f * a11.y);
        v[2] = 0.75f * (0.75f * a00.z + 0.25f * a01.z) + 0.25f * (0.75f * a10.z + 0.25f * a11.z);
        v[3] = 0.75f * (0.75f * a00.w + 0.25f * a01.w) + 0.25f * (0.75f * a10.w + 0.25f * a11.w);
        upc_finish(p, v, sc, sh, oimg + (static_cast<long>(py) * W2 + px) * p.C + co);
      }
  }
  }
}

// The interior kernel on 16-bit storage with 16-byte accesses: item = (source pixel, channel OCTET) -- nine 16-byte loads, four
// 16-byte stores (the form above moves 8 bytes per lane there: 1.5-1.8 TB/s at 64 clips).  Same arithmetic per element.
template <typename T>
__global__ __launch_bounds__(256) void up2_conv_commute16_kernel(UpCommuteArgs<T> p, int gx, int rows_per_xcd) {
  const int c8n = p.C >> 3;
  const int H2 = 2 * p.h, W2 = 2 * p.w;
  const int row_items = p.w * c8n;
  const long crow = static_cast<long>(p.w + 2) * p.C;
  // 1-D grid, XCD-aware: workgroup b = 8 slot + xcd takes piece slot % gx of source row xcd * rows_per_xcd + slot / gx -- an XCD owns a
  // contiguous run of (image, source row) rows, so the three rows of c an output row reads are fetched into ONE L2
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bx = slot % gx;
  {
    const long row = static_cast<long>(xcd) * rows_per_xcd + slot / gx;
    if (slot / gx >= rows_per_xcd || row >= static_cast<long>(p.N) * p.h) return;
    const int n = static_cast<int>(row / p.h), y = static_cast<int>(row - static_cast<long>(n) * p.h);
    const T* cimg = p.c + static_cast<long>(n) * (p.h + 2) * crow;
    T* oimg = p.out + static_cast<long>(n) * H2 * W2 * p.C;
    for (int it = bx * 256 + threadIdx.x; it < row_items; it += gx * 256) {
      const int x = it / c8n, co = (it - x * c8n) * 8;
      f8v cc[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) cc[a][b] = ld8(cimg + (y + a) * crow + static_cast<long>(x + b) * p.C + co);
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
      if (p.scale) {
        const float4 s0 = ld4(p.scale + co), s1 = ld4(p.scale + co + 4), h0 = ld4(p.shift + co), h1 = ld4(p.shift + co + 4);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
      }
#pragma unroll
      for (int sy = 0; sy < 2; ++sy)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const int py = 2 * y + sy, px = 2 * x + sx;
          if (py < 3 || py >= H2 - 3 || px < 3 || px >= W2 - 3) continue;
          const int ya = sy ? 2 : 0, xa = sx ? 2 : 0;
          f8v v;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = 0.75f * (0.75f * cc[1][1].v[e] + 0.25f * cc[1][xa].v[e]) + 0.25f * (0.75f * cc[ya][1].v[e] + 0.25f * cc[ya][xa].v[e]);
            t = t * sc[e] + sh[e];
            v.v[e] = p.act == DIFFSAL_ACT_RELU ? fmaxf(t, 0.f) : t;
          }
          st8(oimg + (static_cast<long>(py) * W2 + px) * p.C + co, v);
        }
    }
  }
}

// Border ring: item = (ring pixel, channel quad); ring pixel r of an image: the six full rows first (0, 1, 2, H2-3, H2-2, H2-1),
// then six columns of each remaining row.
template <typename T>
__global__ __launch_bounds__(256) void up2_conv_commute_ring_kernel(UpCommuteArgs<T> p, int gx) {
  const int c4n = p.C >> 2;
  const int H2 = 2 * p.h, W2 = 2 * p.w;
  const int nb = 2 * p.w + 2 * p.h - 4;
  const int full = H2 < 6 ? H2 : 6;                              // rows that are ring over their whole width
  const int colw = W2 < 6 ? W2 : 6;
  const int per_img = full * W2 + (H2 - full) * colw;
  const long crow = static_cast<long>(p.w + 2) * p.C;
  // 1-D grid, XCD-aware (round 6): workgroup b = 8 slot + xcd takes piece slot % gx of image (slot / gx) * 8 + xcd: the gx workgroups of an
  // image run on ONE XCD, whose L2 holds that image's tap products once
  const bool walk_ = gx > 0;              // gx < 0: image = blockIdx.x / |gx| in dispatch order (DIFFSAL_NO_XCD_ORDER=1)
  gx = gx > 0 ? gx : -gx;
  const int xcd_ = blockIdx.x & 7, slot_ = blockIdx.x >> 3;
  const int bx_ = walk_ ? slot_ % gx : static_cast<int>(blockIdx.x) % gx;
  {
  const int n = walk_ ? (slot_ / gx) * 8 + xcd_ : static_cast<int>(blockIdx.x) / gx;
  if (n >= p.N) return;
  const T* cb = p.c + static_cast<long>(n) * (p.h + 2) * crow;
  const T* tbn = p.tb + static_cast<long>(n) * nb * 9 * p.C;
  T* oimg = p.out + static_cast<long>(n) * H2 * W2 * p.C;
  for (int it = bx_ * 256 + threadIdx.x; it < per_img * c4n; it += gx * 256) {
    const int r = it / c4n, co = (it - r * c4n) * 4;
    int py, px;
    if (r < full * W2) {
      const int k = r / W2;
      px = r - k * W2;
      py = H2 < 6 ? k : (k < 3 ? k : H2 - 6 + k);
    } else {
      const int q = r - full * W2, k = q / colw, j = q - k * colw;
      py = 3 + k;
      px = W2 < 6 ? j : (j < 3 ? j : W2 - 6 + j);
    }
    int y0, y1, x0, x1;
    upc_i0(py, y0, y1);
    upc_i0(px, x0, x1);
    const float4 a00 = ld4(cb + (y0 + 1) * crow + static_cast<long>(x0 + 1) * p.C + co);
    const float4 a01 = ld4(cb + (y0 + 1) * crow + static_cast<long>(x1 + 1) * p.C + co);
    const float4 a10 = ld4(cb + (y1 + 1) * crow + static_cast<long>(x0 + 1) * p.C + co);
    const float4 a11 = ld4(cb + (y1 + 1) * crow + static_cast<long>(x1 + 1) * p.C + co);
    float v[4];
    v[0] = 0.75f * (0.75f * a00.x + 0.25f * a01.x) + 0.25f * (0.75f * a10.x + 0.25f * a11.x);
    v[1] = 0.75f * (0.75f * a00.y + 0.25f * a01.y) + 0.25f * (0.75f * a10.y + 0.25f * a11.y);
    v[2] = 0.75f * (0.75f * a00.z + 0.25f * a01.z) + 0.25f * (0.75f * a10.z + 0.25f * a11.z);
    v[3] = 0.75f * (0.75f * a00.w + 0.25f * a01.w) + 0.25f * (0.75f * a10.w + 0.25f * a11.w);
    auto tap_at = [&](int line_idx, int tap) { return ld4(tbn + (static_cast<long>(line_idx) * 9 + tap) * p.C + co); };
    // pixel (row m, border column wx) in the list: the corners live in the row lists
    auto col_line = [&](int wx, int m) {
      const int xc = wx ? p.w - 1 : 0;
      return m == 0 ? xc : (m == p.h - 1 ? p.w + xc : 2 * p.w + wx * (p.h - 2) + (m - 1));
    };
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int qy = py + 2 * (ky - 1);
      int wy;
      const float cy = upc_dcoef(qy, p.h, wy);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int qx = px + 2 * (kx - 1);
        int wx;
        const float cx = upc_dcoef(qx, p.w, wx);
        const int tap = ky * 3 + kx;
        if (cy != 0.f) {          // D_y I0_x: border ROW wy (top / bottom), interpolated along x at qx
          int m0, m1;
          upc_i0(qx, m0, m1);
          if (m0 >= 0 && m0 < p.w) {
            const float4 t = tap_at(wy * p.w + m0, tap);
            const float f = cy * 0.75f;
            v[0] += f * t.x; v[1] += f * t.y; v[2] += f * t.z; v[3] += f * t.w;
          }
          if (m1 >= 0 && m1 < p.w) {
            const float4 t = tap_at(wy * p.w + m1, tap);
            const float f = cy * 0.25f;
            v[0] += f * t.x; v[1] += f * t.y; v[2] += f * t.z; v[3] += f * t.w;
          }
        }
        if (cx != 0.f) {          // I0_y D_x: border COLUMN wx (left / right), interpolated along y at qy
          int m0, m1;
          upc_i0(qy, m0, m1);
          if (m0 >= 0 && m0 < p.h) {
            const float4 t = tap_at(col_line(wx, m0), tap);
            const float f = cx * 0.75f;
            v[0] += f * t.x; v[1] += f * t.y; v[2] += f * t.z; v[3] += f * t.w;
          }
          if (m1 >= 0 && m1 < p.h) {
            const float4 t = tap_at(col_line(wx, m1), tap);
            const float f = cx * 0.25f;
            v[0] += f * t.x; v[1] += f * t.y; v[2] += f * t.z; v[3] += f * t.w;
          }
        }
        if (cy != 0.f && cx != 0.f) {   // D_y D_x: the corner pixel (row wy, column wx), taken from the row list
          const float4 t = tap_at(wy * p.w + (wx ? p.w - 1 : 0), tap);
          const float f = cy * cx;
          v[0] += f * t.x; v[1] += f * t.y; v[2] += f * t.z; v[3] += f * t.w;
        }
      }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) { sc = ld4(p.scale + co); sh = ld4(p.shift + co); }
    upc_finish(p, v, sc, sh, oimg + (static_cast<long>(py) * W2 + px) * p.C + co);
  }
  }
}

// The border pixels of z [N][h][w][C] in the order diffsal_up2_conv_commute reads their tap products: top row, bottom row, then the
// left and the right column without their corner pixels -> zb [N][2 w + 2 h - 4][C].  16-byte pieces, one per lane.
template <typename T>
__global__ __launch_bounds__(256) void border_gather_kernel(const T* __restrict__ z, T* __restrict__ zb, int N, int h, int w, int C) {
  constexpr int EPP = 16 / sizeof(T);                 // elements per piece
  const int ppx = C / EPP, nb = 2 * w + 2 * h - 4;
  const long total = static_cast<long>(N) * nb * ppx;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int q = static_cast<int>(i % ppx);
    const long r = i / ppx;
    const int b = static_cast<int>(r % nb), n = static_cast<int>(r / nb);
    int y, x;
    if (b < w) { y = 0; x = b; }
    else if (b < 2 * w) { y = h - 1; x = b - w; }
    else if (b < 2 * w + h - 2) { y = b - 2 * w + 1; x = 0; }
    else { y = b - 2 * w - (h - 2) + 1; x = w - 1; }
    const float4 v = *reinterpret_cast<const float4*>(z + ((static_cast<long>(n) * h + y) * w + x) * C + q * EPP);
    *reinterpret_cast<float4*>(zb + r * C + q * EPP) = v;
  }
}

// The ring kernel on 16-bit storage: ONE ring pixel per workgroup (blockIdx.x), so the pixel decode, the tap loop and every
// "does this tap touch the border" test are uniform -- scalar instructions and scalar branches instead of per-lane arithmetic in
// front of an 8-byte access (the form above: 170-215 us per stage at 64 clips, most of the K12-tap class); a thread is (image,
// channel octet): 16-byte accesses, 256 / (C / 8) images per workgroup.  Same terms in the same order per element.
template <typename T>
__global__ __launch_bounds__(256) void up2_conv_commute_ring16_kernel(UpCommuteArgs<T> p) {
  const int c8n = p.C >> 3;
  const int ipw = 256 / c8n;                                     // images per workgroup
  const int H2 = 2 * p.h, W2 = 2 * p.w;
  const int nb = 2 * p.w + 2 * p.h - 4;
  const int full = H2 < 6 ? H2 : 6;
  const int colw = W2 < 6 ? W2 : 6;
  const long crow = static_cast<long>(p.w + 2) * p.C;
  // 1-D grid, XCD-aware: workgroups are dealt round-robin over the eight XCDs, so workgroup b = 8 slot + xcd takes ring pixel
  // slot % per_img of image group (slot / per_img) * 8 + xcd -- ALL ring pixels of an image group run on one XCD, whose L2 then holds
  // that group's tap products and c once (with the pixels spread over the XCDs every L2 fetched every image's 235 KB again)
  const int per_img_ = full * W2 + (H2 - full) * colw;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int r = slot % per_img_;
  const int img_group = (slot / per_img_) * 8 + xcd;
  int py, px;
  if (r < full * W2) {
    const int k = r / W2;
    px = r - k * W2;
    py = H2 < 6 ? k : (k < 3 ? k : H2 - 6 + k);
  } else {
    const int q = r - full * W2, k = q / colw, j = q - k * colw;
    py = 3 + k;
    px = W2 < 6 ? j : (j < 3 ? j : W2 - 6 + j);
  }
  const int sub = threadIdx.x / c8n, co = (threadIdx.x - sub * c8n) * 8;
  if (sub >= ipw) return;
  int y0, y1, x0, x1;
  upc_i0(py, y0, y1);
  upc_i0(px, x0, x1);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
  if (p.scale) {
    const float4 s0 = ld4(p.scale + co), s1 = ld4(p.scale + co + 4), h0 = ld4(p.shift + co), h1 = ld4(p.shift + co + 4);
    sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
  }
  {
    const int n = img_group * ipw + sub;
    if (n >= p.N) return;
    const T* cb = p.c + static_cast<long>(n) * (p.h + 2) * crow;
    const T* tbn = p.tb + static_cast<long>(n) * nb * 9 * p.C;
    const f8v a00 = ld8(cb + (y0 + 1) * crow + static_cast<long>(x0 + 1) * p.C + co);
    const f8v a01 = ld8(cb + (y0 + 1) * crow + static_cast<long>(x1 + 1) * p.C + co);
    const f8v a10 = ld8(cb + (y1 + 1) * crow + static_cast<long>(x0 + 1) * p.C + co);
    const f8v a11 = ld8(cb + (y1 + 1) * crow + static_cast<long>(x1 + 1) * p.C + co);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.75f * (0.75f * a00.v[e] + 0.25f * a01.v[e]) + 0.25f * (0.75f * a10.v[e] + 0.25f * a11.v[e]);
    auto add_tap = [&](int line_idx, int tap, float f) {
      const f8v t = ld8(tbn + (static_cast<long>(line_idx) * 9 + tap) * p.C + co);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += f * t.v[e];
    };
    auto col_line = [&](int wx, int m) {
      const int xc = wx ? p.w - 1 : 0;
      return m == 0 ? xc : (m == p.h - 1 ? p.w + xc : 2 * p.w + wx * (p.h - 2) + (m - 1));
    };
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int qy = py + 2 * (ky - 1);
      int wy;
      const float cy = upc_dcoef(qy, p.h, wy);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int qx = px + 2 * (kx - 1);
        int wx;
        const float cx = upc_dcoef(qx, p.w, wx);
        const int tap = ky * 3 + kx;
        if (cy != 0.f) {
          int m0, m1;
          upc_i0(qx, m0, m1);
          if (m0 >= 0 && m0 < p.w) add_tap(wy * p.w + m0, tap, cy * 0.75f);
          if (m1 >= 0 && m1 < p.w) add_tap(wy * p.w + m1, tap, cy * 0.25f);
        }
        if (cx != 0.f) {
          int m0, m1;
          upc_i0(qy, m0, m1);
          if (m0 >= 0 && m0 < p.h) add_tap(col_line(wx, m0), tap, cx * 0.75f);
          if (m1 >= 0 && m1 < p.h) add_tap(col_line(wx, m1), tap, cx * 0.25f);
        }
        if (cy != 0.f && cx != 0.f) add_tap(wy * p.w + (wx ? p.w - 1 : 0), tap, cy * cx);
      }
    }
    f8v o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = v[e] * sc[e] + sh[e];
      o.v[e] = p.act == DIFFSAL_ACT_RELU ? fmaxf(t, 0.f) : t;
    }
    st8(p.out + (static_cast<long>(n) * H2 * W2 + static_cast<long>(py) * W2 + px) * p.C + co, o);
  }
}

}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_border_gather(const void* z, void* zb, int N, int h, int w, int C, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(z && zb, DIFFSAL_E_ARG, "border_gather: null argument");
  // a piece is 16 bytes: four fp32 or eight 16-bit channels
  DS_REQUIRE(N > 0 && h >= 2 && w >= 2 && C > 0 && C % (dtype == DIFFSAL_F32 ? 4 : 8) == 0, DIFFSAL_E_SHAPE,
             "border_gather: N=%d h=%d w=%d C=%d", N, h, w, C);
  DS_REQUIRE(aligned16(z) && aligned16(zb), DIFFSAL_E_ALIGN, "border_gather: misaligned pointer");
  const long pieces = static_cast<long>(N) * (2 * w + 2 * h - 4) * (C / (dtype == DIFFSAL_F32 ? 4 : 8));
  long g = (pieces + 255) / 256;
  g = g > 4096 ? 4096 : g;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(T) hipLaunchKernelGGL(border_gather_kernel<T>, dim3(static_cast<unsigned>(g)), dim3(256), 0, s, static_cast<const T*>(z), static_cast<T*>(zb), N, h, w, C)
  DS_DTYPE_DISPATCH(dtype, "border_gather", CALL);
#undef CALL
  return check_launch("border_gather");
}

namespace {
template <typename T>
int up2_commute_t(const void* c_ext, const void* tap_border, const float* scale, const float* shift, void* out, int N, int h, int w, int C,
                  int act, hipStream_t s, bool ring_only) {
  UpCommuteArgs<T> a{static_cast<const T*>(c_ext), static_cast<const T*>(tap_border), scale, shift, static_cast<T*>(out), N, h, w, C, act};
  if (!ring_only) {
    const int row_items = w * (C / 4);
    int gx = (row_items + 255) / 256;
    gx = gx > 64 ? 64 : gx;
    const long rows = static_cast<long>(N) * h;
    if constexpr (sizeof(T) == 2) {
      if (C % 8 == 0 && tune(TUNE_NO_STREAM16) != 1) {
        int g8 = (w * (C / 8) + 255) / 256;
        g8 = g8 > 64 ? 64 : g8;
        const long rpx = (rows + 7) / 8;                          // rows per XCD
        hipLaunchKernelGGL(up2_conv_commute16_kernel<T>, dim3(static_cast<unsigned>(8 * rpx * g8)), dim3(256), 0, s, a, g8, static_cast<int>(rpx));
      } else {
        hipLaunchKernelGGL(up2_conv_commute_kernel<T>, dim3(gx, static_cast<unsigned>(rows < 65535 ? rows : 65535)), dim3(256), 0, s, a);
      }
    } else {
      hipLaunchKernelGGL(up2_conv_commute_kernel<T>, dim3(gx, static_cast<unsigned>(rows < 65535 ? rows : 65535)), dim3(256), 0, s, a);
    }
    const int rc = check_launch("up2_conv_commute(interior)");
    if (rc) return rc;
  }
  const int H2 = 2 * h, W2 = 2 * w, full = H2 < 6 ? H2 : 6, colw = W2 < 6 ? W2 : 6;
  const long ring_items = (static_cast<long>(full) * W2 + static_cast<long>(H2 - full) * colw) * (C / 4);
  int gr = static_cast<int>((ring_items + 255) / 256);
  gr = gr > 1024 ? 1024 : gr;
  if constexpr (sizeof(T) == 2) {
    if (C % 8 == 0 && C / 8 <= 256 && tune(TUNE_NO_STREAM16) != 1) {
      const int per_img = full * W2 + (H2 - full) * colw, ipw = 256 / (C / 8);
      const long groups8 = ((N + ipw - 1) / ipw + 7) / 8 * 8;          // image groups, a multiple of the XCD count
      if (groups8 * per_img < (1L << 31))
        hipLaunchKernelGGL(up2_conv_commute_ring16_kernel<T>, dim3(static_cast<unsigned>(groups8 * per_img)), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL(up2_conv_commute_ring_kernel<T>, dim3(static_cast<unsigned>(static_cast<long>(gr) * ((N + 7) / 8 * 8))), dim3(256), 0, s, a, tune(TUNE_NO_XCD_ORDER) == 1 ? -gr : gr);
      return check_launch("up2_conv_commute(ring, 16-bit)");
    }
  }
  hipLaunchKernelGGL(up2_conv_commute_ring_kernel<T>, dim3(static_cast<unsigned>(static_cast<long>(gr) * ((N + 7) / 8 * 8))), dim3(256), 0, s, a, tune(TUNE_NO_XCD_ORDER) == 1 ? -gr : gr);
  return check_launch("up2_conv_commute");
}
}  // namespace

static int up2_commute_impl(const void* c_ext, const void* tap_border, const float* scale, const float* shift, void* out,
                            int N, int h, int w, int C, int act, int dtype, diffsal_stream_t stream, bool ring_only) {
  DS_REQUIRE(c_ext && tap_border && out, DIFFSAL_E_ARG, "up2_conv_commute: null argument");
  DS_REQUIRE(N > 0 && h >= 2 && w >= 2 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "up2_conv_commute: N=%d h=%d w=%d C=%d", N, h, w, C);
  DS_REQUIRE((scale == nullptr) == (shift == nullptr), DIFFSAL_E_ARG, "up2_conv_commute: scale and shift come together");
  DS_REQUIRE(act == DIFFSAL_ACT_NONE || act == DIFFSAL_ACT_RELU, DIFFSAL_E_ARG, "up2_conv_commute: act=%d", act);
  DS_REQUIRE(aligned16(c_ext) && aligned16(tap_border) && aligned16(out) && (!scale || (aligned16(scale) && aligned16(shift))),
             DIFFSAL_E_ALIGN, "up2_conv_commute: misaligned pointer");
  DS_REQUIRE(static_cast<long>(w) * (C / 4) < (1L << 30) && static_cast<long>(N) * h < (1L << 31), DIFFSAL_E_SHAPE,
             "up2_conv_commute: row too long");
  const long ring_bound = (6L * 2 * w + 2L * h * 6) * (C / 4);
  DS_REQUIRE(ring_bound < (1L << 30), DIFFSAL_E_SHAPE, "up2_conv_commute: ring too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(T) return up2_commute_t<T>(c_ext, tap_border, scale, shift, out, N, h, w, C, act, s, ring_only)
  DS_DTYPE_DISPATCH(dtype, "up2_conv_commute", CALL);
#undef CALL
  return DIFFSAL_OK;
}

extern "C" int diffsal_up2_conv_commute(const void* c_ext, const void* tap_border, const float* scale, const float* shift, void* out,
                                        int N, int h, int w, int C, int act, int dtype, diffsal_stream_t stream) {
  return up2_commute_impl(c_ext, tap_border, scale, shift, out, N, h, w, C, act, dtype, stream, false);
}

/* Only the 3-pixel border ring of `out` is written (the pixels whose value needs the tap products): for a consumer that forms the
 * interior itself from c_ext -- diffsal_conv_wino4_ex with ext.up2_c (the interior never exists in memory). */
extern "C" int diffsal_up2_conv_commute_ring(const void* c_ext, const void* tap_border, const float* scale, const float* shift, void* out,
                                             int N, int h, int w, int C, int act, int dtype, diffsal_stream_t stream) {
  return up2_commute_impl(c_ext, tap_border, scale, shift, out, N, h, w, C, act, dtype, stream, true);
}
