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
#include <type_traits>

#include "common.h"

namespace diffsal {

struct TapSumArgs {
  const void* in[4];   // Y_i: [N, h_i, w_i, 9 * C] (tap-major channel blocks), storage type of the launch
  int h[4], w[4];
  float sy[4], sx[4];
  int lf[4];           // log2 of the factor H / h (head kernels: every factor is a power of two)
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
// mt_proj form (R/models/saliency_decoder/sal_unet.py:480-489, :407: dilation 1, every source at a power-of-two factor, MLPHead
// folded): the same gather, reorganised so that it is no longer bound by vector-instruction issue.
//
// A 4 x 4 output patch whose origin is a multiple of 4, shifted by the nine taps, touches the 6 x 6 target positions
// (Y0 - 1 + j, X0 - 1 + i), j, i = 0..5; per source and axis they read NL = 4 (factor 2) or 3 (factor >= 4) consecutive source
// lines.  tapsum_kernel recomputes coordinates and two 4 x 4 weight tables for each of the 36 (tap, source) pairs and then
// interpolates densely: ~35 K instructions per wavefront, vector-issue bound at 134 us.  Here, per source:
//   * the six positions of an axis are set up ONCE (weights of bilin_coord: align_corners = False, index clamped; a position
//     outside the image has weight 0 -- the convolution's zero padding);
//   * the sum runs over (ky, source line r):  G[e] = sum_kx sum_c wx[e + kx][c] Y[ky, kx][r][c]  (the horizontal pass of the
//     three taps of a kernel row TOGETHER), then acc[d][e] += wy[d + ky][r] G[e]: the vertical pass is paid once per kernel row,
//     not once per tap -- 360-480 FMAs per channel, patch and source where the per-tap separable form needs 864-1152;
//   * factor 2 / 4: the column pair of position i is the compile-time pattern i / 2, i / 3 (two-tap horizontal pass);
//   * (ky, r) are RUNTIME loops (the vertical weights come from a 24-entry table in LDS, written once per source), the loads
//     of the next (ky, r) are issued before the current one is summed: ~5 K instructions per wavefront at ~150 registers.
//     (A fully unrolled per-tap version was built first: hipcc gave it 300-400 registers whatever the ring depth, one
//     wavefront per SIMD, and it was no faster than tapsum_kernel -- profiles/NOTES.md.)
// CPL channels per lane: 3 when C = 96 (all 32 lanes of an image's half-wave busy), else 4.
// ------------------------------------------------------------------------------------------------
template <int CPL> struct TapVec { float v[CPL]; };

// uniform base (scalar registers) + this lane's 32-bit byte offset: addressed by the hardware as saddr + voffset
template <int CPL>
__device__ __forceinline__ TapVec<CPL> tap_ld(const char* ubase, unsigned lane_byte) {
  TapVec<CPL> r;
  const char* p = ubase + static_cast<unsigned long>(lane_byte);
  if constexpr (CPL == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
  else {
    struct __attribute__((packed, aligned(4))) Piece { float v[CPL]; };      // one 12-byte load
    const Piece t = *reinterpret_cast<const Piece*>(p);
#pragma unroll
    for (int e = 0; e < CPL; ++e) r.v[e] = t.v[e];
  }
  return r;
}

// position p of an axis of L target / n source samples (factor f = L / n): the pair's lower line BEFORE clamping and the weights
// (a on it, b on the next); a = b = 0 outside [0, L)
__device__ __forceinline__ void tap_pos(int p, int L, int n, float scale, int& lower, float& a, float& b) {
  const int f = L / n;
  const bool valid = p >= 0 && p < L;
  int i0, i1;
  float l1;
  bilin_coord(valid ? p : 0, scale, n, i0, i1, l1);
  const int num = 2 * p + 1 - f;                    // floor((p + 0.5) / f - 0.5), -1 where bilin_coord clamps the coordinate at 0
  lower = num >= 0 ? num / (2 * f) : -1;
  a = 1.f - l1;
  b = l1;
  if (lower < 0) { a = 0.f; b = 1.f; }              // both lines of the pair clamp to line 0: its whole weight
  if (!valid) { a = 0.f; b = 0.f; }
}

// PS: the patch is PS rows x 4 columns of output pixels (PS = 4 or 8; origin a multiple of (PS, 4)): PS + 2 positions and
// NLY = PS / F + 2 source lines (3 from factor 8 on) on the y axis, 6 positions and NL = 4 / F + 2 columns on the x axis.  The tall
// patch reads (8 / F + 2)(4 / F + 2) source pixels per source where two 4 x 4 patches read 2 (4 / F + 2)^2: 24 against 32 at factor 2,
// 12 against 18 at factor 4, 9 against 18 beyond -- the kernel's requests to the L2 were 8x the tap products' bytes (12.8 GB at 64
// clips: ~10 TB/s, what the L2s deliver), now 5x.  (An 8 x 8 patch -- 3.3x -- needs 192 accumulators and two sets of 54 staging
// registers: hipcc spills 217-306 of them.)  Same sums in the same order: the extra lines of a position carry weight 0.
// SX: the two 32-lane halves of the wavefront hold two horizontally adjacent patches of ONE image instead of the same patch of two
// images: the x-axis weights and column offsets become per-lane values, and the halves' column windows overlap (columns 2, 3 of the left
// half are columns 0, 1 of the right one at factor 2) in loads issued back to back by one wavefront -- one fetch from the L2.
template <int CPL, int F, int PS, bool SX>
__device__ __forceinline__ void tap_source_rows(const char* __restrict__ src, unsigned lane_byte, int h, int w, int P, int C, int Y0,
                                                int X0, int H, int W, float sy, float sx, float* __restrict__ wtab,
                                                float (&acc)[PS][4][CPL]) {
  constexpr int NPY = PS + 2;                                  // positions on the y axis
  constexpr int NL = F == 2 ? 4 : 3;                           // source columns
  constexpr int NLY = F == 2 ? PS / 2 + 2 : (F == 4 ? PS / 4 + 2 : 3);      // source lines
  constexpr int NLP = PS == 4 ? 4 : 6;                         // pitch of the vertical weight table
  const int lane = threadIdx.x & 63;
  // ---- x axis: six positions, weights in registers (every lane computes the same values)
  float wxa[6], wxb[6];
  int xr[6], first_x = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int lower;
    float a, b;
    tap_pos(X0 - 1 + i, W, w, sx, lower, a, b);
    wxa[i] = SX ? a : uni_f(a);                     // !SX: wave-uniform, scalar registers
    wxb[i] = SX ? b : uni_f(b);
    if (i == 0) first_x = lower;
    xr[i] = SX ? lower - first_x : uni_i(lower - first_x);     // F = 2, 4: (2 i - 1 + F) / (2 F) by construction; F = 0: 0 or 1
  }
  if constexpr (!SX) first_x = uni_i(first_x);
  unsigned coff[NL];               // !SX: uniform column offset (added to the scalar base); SX: this lane's byte offset, column included
#pragma unroll
  for (int c = 0; c < NL; ++c) {
    const int col = min(max(first_x + c, 0), w - 1);
    coff[c] = SX ? lane_byte + static_cast<unsigned>(col * P * 4) : static_cast<unsigned>(uni_i(col) * P * 4);
  }
  // ---- y axis: dense weights wtab[j * 4 + r] of line r (0 .. NL-1) for position j, one entry per lane
  int first_y;
  {
    float a0, b0;
    tap_pos(Y0 - 1, H, h, sy, first_y, a0, b0);
    first_y = uni_i(first_y);
    if (lane < NPY * NLP) {
      const int j = lane / NLP, r = lane - j * NLP;
      int lower;
      float a, b;
      tap_pos(Y0 - 1 + j, H, h, sy, lower, a, b);
      wtab[lane] = (lower - first_y == r ? a : 0.f) + (lower - first_y + 1 == r ? b : 0.f);
    }
  }
  // ---- (ky, r) items; item it = ky * NL + r.  V[kx][c]: the NL vectors of source line r in tap (ky, kx).  The loads of item
  // it + 1 are issued before item it is summed (two register sets, two items per trip so that the sets swap roles without copies).
  // Measured: a ring of three sets at sub-step (kx) granularity -- a third of the registers, two wavefronts per SIMD instead of
  // one -- is SLOWER (99 against 87 us): what the gather needs is loads in flight, and a whole item ahead is 9-12 of them.
  constexpr int NIT = 3 * NLY;
  auto fetch = [&](TapVec<CPL> (&V)[3][NL], int it) __attribute__((always_inline)) {
    const int ky = it / NLY, r = it - ky * NLY;
    const int row = min(max(first_y + r, 0), h - 1);
    const int roff = row * w * P * 4 + ky * 3 * C * 4;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int c = 0; c < NL; ++c) {    // the offset IS uniform; inside the runtime loop hipcc no longer proves it: say so
        if constexpr (SX) V[kx][c] = tap_ld<CPL>(src + static_cast<unsigned>(uni_i(roff + kx * C * 4)), coff[c]);
        else V[kx][c] = tap_ld<CPL>(src + static_cast<unsigned>(uni_i(roff + static_cast<int>(coff[c]) + kx * C * 4)), lane_byte);
      }
  };
  auto sum = [&](const TapVec<CPL> (&V)[3][NL], int it) __attribute__((always_inline)) {
    const int ky = it / NLY, r = it - ky * NLY;
    float wv[PS];
#pragma unroll
    for (int d = 0; d < PS; ++d) wv[d] = wtab[(d + ky) * NLP + r];
    float G[4][CPL];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < CPL; ++q) G[e][q] = 0.f;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = e + kx;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          if constexpr (F != 0) {
            constexpr int F2 = F != 0 ? 2 * F : 1;
            const int col = (2 * i - 1 + F) / F2;           // the pair's lower column, relative to the first position's
            G[e][q] = fmaf(wxb[i], V[kx][col + 1].v[q], fmaf(wxa[i], V[kx][col].v[q], G[e][q]));
          } else {          // factor >= 8: the pair of position i starts at column xr[i] in {0, 1}: dense three-column weights
            const float w0 = xr[i] == 0 ? wxa[i] : 0.f, w1 = xr[i] == 0 ? wxb[i] : wxa[i], w2 = xr[i] == 0 ? 0.f : wxb[i];
            G[e][q] = fmaf(w2, V[kx][2].v[q], fmaf(w1, V[kx][1].v[q], fmaf(w0, V[kx][0].v[q], G[e][q])));
          }
        }
      }
#pragma unroll
    for (int d = 0; d < PS; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[d][e][q] = fmaf(wv[d], G[e][q], acc[d][e][q]);
  };
  // (Deeper rings were tried with the items unrolled so that the set indices are static: hipcc then allocates 256-512 registers
  // and spills, scheduling barriers or not.  Two sets in a runtime loop is what it compiles well.)
  if constexpr (CPL <= 2) {
    // light form (2 channels per lane): THREE register sets, the loads of items it + 1 and it + 2 in flight while item it is
    // summed; NIT = 9 or 12 is a multiple of 3, so three items per trip keep the sets' roles fixed
    TapVec<CPL> Va[3][NL], Vb[3][NL], Vc[3][NL];
    fetch(Va, 0);
    fetch(Vb, 1);
#pragma unroll 1
    for (int it = 0; it < NIT - 3; it += 3) {
      fetch(Vc, it + 2);
      sum(Va, it);
      fetch(Va, it + 3);
      sum(Vb, it + 1);
      fetch(Vb, it + 4);
      sum(Vc, it + 2);
    }
    fetch(Vc, NIT - 1);
    sum(Va, NIT - 3);
    sum(Vb, NIT - 2);
    sum(Vc, NIT - 1);
  } else {
    TapVec<CPL> Va[3][NL], Vb[3][NL];
    fetch(Va, 0);
    constexpr int PAIRS = (NIT - 1) / 2;             // trips with both fetches unconditional (no phi copies of the register sets)
#pragma unroll 1
    for (int it = 0; it < 2 * PAIRS; it += 2) {
      fetch(Vb, it + 1);
      sum(Va, it);
      fetch(Va, it + 2);
      sum(Vb, it + 1);
    }
    if constexpr (NIT % 2 == 1) {
      sum(Va, NIT - 1);
    } else {
      fetch(Vb, NIT - 1);
      sum(Va, NIT - 2);
      sum(Vb, NIT - 1);
    }
  }
}

// IMGS images per wavefront (64 / IMGS lanes each, CPL channels per lane): 2 x 32 lanes x 3 channels for C = 96 keeps every lane
// busy; 1 x 64 lanes x 2 channels (48 live lanes at C = 96) makes twice as many, lighter wavefronts -- more of them resident per
// SIMD, more loads in flight (the launch is latency-bound: ~2.6 wavefronts per SIMD in the 2-image form)
template <int CPL, int IMGS, int PS, bool SX = false>
__global__ __launch_bounds__(256, IMGS == 1 ? 2 : 1) void tapsum_head_rows_kernel(TapSumArgs a, int w_patches, long n_items, int row_map) {
  static_assert(!SX || IMGS == 2, "SX: the two halves of a wavefront are two patches of one image");
  constexpr int LPI = 64 / IMGS;
  __shared__ float wtab_all[4][64];
  const int lane = threadIdx.x & 63, sub = lane / LPI, li = lane % LPI;
  const unsigned nb = gridDim.x, xq = nb >> 3, xr = nb & 7u, xcd = blockIdx.x & 7u;       // XCD-aware order, as tapsum_kernel
  const unsigned vb = xcd * xq + (xcd < xr ? xcd : xr) + (blockIdx.x >> 3);
  const long item = static_cast<long>(vb) * 4 + (threadIdx.x >> 6);
  if (item >= n_items) return;
  float* wtab = wtab_all[threadIdx.x >> 6];
  const int h_patches = a.H / PS;
  int px, py, grp;
  if (!row_map && ((w_patches | h_patches) & 1) == 0) {
    // the four wavefronts of a workgroup take a 2 x 2 block of patches (not four in a row): their source windows overlap in rows
    // AND columns, and they run on one CU -- one L1
    const long blk = item >> 2;
    const int wv = static_cast<int>(item & 3), wb = w_patches >> 1, hb = h_patches >> 1;
    px = static_cast<int>(blk % wb) * 2 + (wv & 1);
    py = static_cast<int>((blk / wb) % hb) * 2 + (wv >> 1);
    grp = static_cast<int>(blk / (static_cast<long>(wb) * hb));
  } else {
    long t = item;
    px = static_cast<int>(t % w_patches); t /= w_patches;
    py = static_cast<int>(t % h_patches);
    grp = static_cast<int>(t / h_patches);
  }
  const int Y0 = uni_i(py * PS), X0 = SX ? (px * 2 + sub) * 4 : uni_i(px * 4);     // SX: w_patches counts 8-pixel patch pairs
  const int n = SX ? grp : grp * IMGS + sub, c = li * CPL;
  const int nc = n < a.N ? n : a.N - 1, cc = c < a.C ? c : a.C - CPL;   // dead lanes walk valid memory and contribute nothing
  const int P = 9 * a.C;
  float acc[PS][4][CPL];
#pragma unroll
  for (int d = 0; d < PS; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < CPL; ++q) acc[d][e][q] = 0.f;
#pragma unroll 1
  for (int s = 0; s < a.n_in; ++s) {
    const int hs = a.h[s], ws = a.w[s];
    const int f = a.H / hs;
    const char* src = static_cast<const char*>(a.in[s]);
    const unsigned lane_byte = static_cast<unsigned>((static_cast<long>(nc) * hs * ws * P + cc) * 4);     // < 4 GiB: checked by the host
    if (f == 2) tap_source_rows<CPL, 2, PS, SX>(src, lane_byte, hs, ws, P, a.C, Y0, X0, a.H, a.W, a.sy[s], a.sx[s], wtab, acc);
    else if (f == 4) tap_source_rows<CPL, 4, PS, SX>(src, lane_byte, hs, ws, P, a.C, Y0, X0, a.H, a.W, a.sy[s], a.sx[s], wtab, acc);
    else tap_source_rows<CPL, 0, PS, SX>(src, lane_byte, hs, ws, P, a.C, Y0, X0, a.H, a.W, a.sy[s], a.sx[s], wtab, acc);
  }
  float bi[CPL], sc[CPL], sh[CPL], hw[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    bi[q] = a.bias ? a.bias[cc + q] : 0.f;
    sc[q] = a.scale ? a.scale[cc + q] : 1.f;
    sh[q] = a.scale ? a.shift[cc + q] : 0.f;
    hw[q] = c < a.C ? a.head_w[c + q] : 0.f;          // lanes beyond C contribute nothing to the pixel's dot product
  }
  float hb = a.head_b[0];                             // read once and pinned (see tapsum_head_lds_kernel)
  asm volatile("" : "+v"(hb));
#pragma unroll
  for (int d = 0; d < PS; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t2 = 0.f;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        float v = (acc[d][e][q] + bi[q]) * sc[q] + sh[q];
        if (a.act == DIFFSAL_ACT_RELU) v = fmaxf(v, 0.f);
        t2 = fmaf(v, hw[q], t2);
      }
      t2 = group_sum<LPI>(t2);
      if (li == 0 && n < a.N) a.head_out[(static_cast<long>(n) * a.H + Y0 + d) * a.W + X0 + e] = sigmoidf_(t2 + hb);
    }
}

// ------------------------------------------------------------------------------------------------
// mt_proj form, source lines staged in LDS by the whole workgroup.
//
// tapsum_head_rows_kernel is one wavefront per SIMD (400 registers) that waits a full memory latency for every (ky, line) item, and
// its patches re-read their neighbours' source pixels from the L2: at 64 clips 5x the tap products' bytes at the ~10 TB/s the L2s
// deliver -- both limits at once.  Here a workgroup (4 wavefronts, 2 x 2, a wavefront = two adjacent 8 x 4 patches as above) owns a
// 16 x 16 block of one image and walks the flattened (source, ky, line) items of the BLOCK's window -- 16 / f + 2 lines of 16 / f + 2
// source pixels x 3 taps x C fp32 (11.5 KB at factor 2) -- through a six-slot LDS ring filled by LDS-DMA four items ahead (three
// 1 KiB instructions per wavefront and item, the 1152 contiguous bytes of a pixel's three kx taps as 72 pieces; lanes past the window
// re-read its last column): every source pixel of the window crosses the L2 -> CU path once per block (1.9x the tap products' bytes
// instead of 5x), the latency is covered by the ring instead of by registers, and at 170 registers two workgroups share a CU.  A
// wavefront multiplies the lines of an item that its own 8 rows touch (6 of 10 at factor 2) and skips the others -- they carried
// weight 0 in the row-streamed kernel: same sums, same order, same bits.
// ------------------------------------------------------------------------------------------------
constexpr int kTlSlot = 12288, kTlDepth = 6;
typedef int tl_i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* tl_lds_ptr_t;

__device__ __forceinline__ void tl_dma(unsigned lds_addr, unsigned voff, tl_i32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}

// one (ky, line) item of this wavefront: slot = the staged line (pixel-major: [column][3 kx][C] fp32), rl = the line's index among the
// wavefront's own lines, col0 = this lane half's first column in the block's window, ncb = the window's columns
template <int CPL, int F>
__device__ __forceinline__ void tl_item(const float* __restrict__ slot, int ky, int rl, int col0, int ncb, int C, int li,
                                        const float* __restrict__ wtab, const float (&wxa)[6], const float (&wxb)[6], const int (&xr)[6],
                                        float (&acc)[8][4][CPL]) {
  constexpr int NL = F == 2 ? 4 : 3;
  float wv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) wv[d] = wtab[(d + ky) * 6 + rl];
  float G[4][CPL];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int q = 0; q < CPL; ++q) G[e][q] = 0.f;
  int cidx[NL];
#pragma unroll
  for (int c = 0; c < NL; ++c) cidx[c] = min(col0 + c, ncb - 1) * 3 * C + li * CPL;   // a column past the window carries weight 0
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    float V[NL][CPL];
#pragma unroll
    for (int c = 0; c < NL; ++c)
#pragma unroll
      for (int q = 0; q < CPL; ++q) V[c][q] = slot[cidx[c] + kx * C + q];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = e + kx;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        if constexpr (F != 0) {
          constexpr int F2 = F != 0 ? 2 * F : 1;
          const int col = (2 * i - 1 + F) / F2;
          G[e][q] = fmaf(wxb[i], V[col + 1][q], fmaf(wxa[i], V[col][q], G[e][q]));
        } else {
          const float w0 = xr[i] == 0 ? wxa[i] : 0.f, w1 = xr[i] == 0 ? wxb[i] : wxa[i], w2 = xr[i] == 0 ? 0.f : wxb[i];
          G[e][q] = fmaf(w2, V[2][q], fmaf(w1, V[1][q], fmaf(w0, V[0][q], G[e][q])));
        }
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < CPL; ++q) acc[d][e][q] = fmaf(wv[d], G[e][q], acc[d][e][q]);
}

// tap_pos for a power-of-two factor 2^lf (no integer division)
__device__ __forceinline__ void tap_pos2(int p, int L, int n, float scale, int lf, int& lower, float& a, float& b) {
  const bool valid = p >= 0 && p < L;
  int i0, i1;
  float l1;
  bilin_coord(valid ? p : 0, scale, n, i0, i1, l1);
  const int num = 2 * p + 1 - (1 << lf);
  lower = num >= 0 ? num >> (lf + 1) : -1;
  a = 1.f - l1;
  b = l1;
  if (lower < 0) { a = 0.f; b = 1.f; }
  if (!valid) { a = 0.f; b = 0.f; }
}

// the DMA side's position: which item of which source goes out next (the ring runs on across sources: while the last items of a source
// are multiplied, the first items of the next one are already on their way)
struct TlIssue {
  unsigned voff[3];     // this lane's three pieces of a line: byte offset of (window column, piece)
  tl_i32x4 rsrc;        // the source's image n
  int s, bfy, hs, ws, nlb, iky, ir, slot;
  bool done;            // the source's last item has gone out
};

template <int C>
__device__ __forceinline__ void tl_issue_setup(const TapSumArgs& a, TlIssue& st, int s, int n, int BY, int BX) {
  constexpr int P = 9 * C, PPC = 3 * C / 4;
  const int tid = threadIdx.x;
  st.s = s;
  st.hs = a.h[s]; st.ws = a.w[s];
  st.nlb = max(16 >> a.lf[s], 1) + 2;                          // lines and columns of the block's window
  int lo; float aa, bb;
  tap_pos2(BY - 1, a.H, st.hs, a.sy[s], a.lf[s], lo, aa, bb);
  st.bfy = uni_i(lo);
  tap_pos2(BX - 1, a.W, st.ws, a.sx[s], a.lf[s], lo, aa, bb);
  const int bfx = uni_i(lo);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int pc = q * 256 + tid;
    const int ci = pc / PPC, j = pc - ci * PPC;
    const int col = min(max(bfx + min(ci, st.nlb - 1), 0), st.ws - 1);
    st.voff[q] = static_cast<unsigned>(col * P * 4 + j * 16);
  }
  const unsigned long base = reinterpret_cast<unsigned long>(a.in[s]) + static_cast<unsigned long>(n) * st.hs * st.ws * P * 4ul;
  st.rsrc = tl_i32x4{static_cast<int>(base), static_cast<int>(base >> 32) & 0xFFFF, st.hs * st.ws * P * 4, 0x00020000};
  st.iky = 0; st.ir = 0; st.done = false;
}

// one item (three DMA instructions per wavefront) into the ring's next slot; past a source's last item: that item again (into a free slot)
template <int C>
__device__ __forceinline__ void tl_issue_one(TlIssue& st, unsigned lds0, int wave) {
  constexpr int P = 9 * C;
  const int line = min(max(st.bfy + st.ir, 0), st.hs - 1);
  const unsigned soff = static_cast<unsigned>(line * st.ws * P * 4 + st.iky * 3 * C * 4);
#pragma unroll
  for (int q = 0; q < 3; ++q) tl_dma(lds0 + st.slot * kTlSlot + (q * 4 + wave) * 1024, st.voff[q], st.rsrc, soff);
  st.slot = st.slot + 1 == kTlDepth ? 0 : st.slot + 1;
  const bool wrap = st.ir + 1 == st.nlb, last = wrap && st.iky == 2;
  st.done = st.done || last;
  st.ir = last ? st.ir : (wrap ? 0 : st.ir + 1);
  st.iky = wrap && !last ? st.iky + 1 : st.iky;
}

// the items of one source (F: its factor class -- column pattern, lines per wavefront): wait, barrier, issue five items ahead, multiply
template <int CPL, int F, int C>
__device__ __forceinline__ void tl_source(const TapSumArgs& a, int s, int n, int BY, int BX, int Y0, int X0, unsigned lds0,
                                          const unsigned char* __restrict__ smem, float* __restrict__ wtab, TlIssue& is, int& sl,
                                          float (&acc)[8][4][CPL]) {
  constexpr int NLYW = F == 2 ? 6 : (F == 4 ? 4 : 3);
  const int lane = threadIdx.x & 63, wave = uni_i(threadIdx.x >> 6), li = lane & 31;
  const int hs = a.h[s], ws = a.w[s], lf = a.lf[s];
  const int nlb = max(16 >> lf, 1) + 2;
  const int n_it = 3 * nlb;
  // ---- window origin, this lane half's x weights, this wavefront's y weights
  float wxa[6], wxb[6];
  int xr[6], first_x = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int lower; float aa, bb;
    tap_pos2(X0 - 1 + i, a.W, ws, a.sx[s], lf, lower, aa, bb);
    wxa[i] = aa; wxb[i] = bb;
    if (i == 0) first_x = lower;
    xr[i] = lower - first_x;
  }
  int lo, first_yw; float aa, bb;
  tap_pos2(BX - 1, a.W, ws, a.sx[s], lf, lo, aa, bb);
  const int col0 = first_x - uni_i(lo);
  tap_pos2(Y0 - 1, a.H, hs, a.sy[s], lf, first_yw, aa, bb);
  first_yw = uni_i(first_yw);
  tap_pos2(BY - 1, a.H, hs, a.sy[s], lf, lo, aa, bb);
  const int lw = first_yw - uni_i(lo);
  if (lane < 60) {                                 // (the table is this wavefront's own: its reads of the previous source's are behind it)
    const int j = lane / 6, r = lane - j * 6;
    int lower;
    tap_pos2(Y0 - 1 + j, a.H, hs, a.sy[s], lf, lower, aa, bb);
    wtab[lane] = (lower - first_yw == r ? aa : 0.f) + (lower - first_yw + 1 == r ? bb : 0.f);
  }
  int cky = 0, cr = 0;
#pragma unroll 1
  for (int it = 0; it < n_it; ++it) {
    // this wavefront's pieces of item `it` have landed (three per item: those of the four items after it may be in flight) and its LDS
    // reads of the item before have returned; after the barrier everybody's have, and that item's slot takes the item five ahead
    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    tl_issue_one<C>(is, lds0, wave);
    if (is.done && is.s + 1 < a.n_in) tl_issue_setup<C>(a, is, is.s + 1, n, BY, BX);     // (no vector memory operation in there)
    const int rl = cr - lw;
    if (rl >= 0 && rl < NLYW)
      tl_item<CPL, F>(reinterpret_cast<const float*>(smem + sl * kTlSlot), cky, rl, col0, nlb, C, li, wtab, wxa, wxb, xr, acc);
    sl = sl + 1 == kTlDepth ? 0 : sl + 1;
    if (++cr == nlb) { cr = 0; ++cky; }
  }
}

template <int CPL, int C>
__global__ __launch_bounds__(256, 2) void tapsum_head_lds_kernel(TapSumArgs a, int bw, int bh) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
  float* wtab = reinterpret_cast<float*>(tl_smem + kTlDepth * kTlSlot) + (threadIdx.x >> 6) * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni_i(tid >> 6), sub = lane >> 5, li = lane & 31;
  const unsigned nb = gridDim.x, xq = nb >> 3, xrm = nb & 7u, xcd = blockIdx.x & 7u;       // XCD-aware order: an image's blocks on one XCD
  const unsigned vb = xcd * xq + (xcd < xrm ? xcd : xrm) + (blockIdx.x >> 3);
  const int bx = static_cast<int>(vb % bw), by = static_cast<int>((vb / bw) % bh), n = static_cast<int>(vb / (static_cast<unsigned>(bw) * bh));
  const int BY = by * 16, BX = bx * 16;
  const int Y0 = BY + (wave >> 1) * 8;                         // wave-uniform
  const int X0 = BX + ((wave & 1) * 2 + sub) * 4;              // per lane half
  const int c0 = li * CPL;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((tl_lds_ptr_t)tl_smem));
  float acc[8][4][CPL];
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < CPL; ++q) acc[d][e][q] = 0.f;
  TlIssue is;
  is.slot = 0;
  tl_issue_setup<C>(a, is, 0, n, BY, BX);
#pragma unroll 1
  for (int d = 0; d < kTlDepth - 1; ++d) {
    tl_issue_one<C>(is, lds0, wave);
    if (is.done && is.s + 1 < a.n_in) tl_issue_setup<C>(a, is, is.s + 1, n, BY, BX);
  }
  // the sources come coarsest first (host-checked): factor >= 8, then 4, then 2 -- three loops in a row, one instantiation each (as the
  // arms of one branch inside a single loop they cost hipcc 354 registers instead of 254: the 96 accumulators meet in a three-way join)
  int s = 0, sl = 0;
#pragma unroll 1
  for (; s < a.n_in && a.lf[s] >= 3; ++s) tl_source<CPL, 0, C>(a, s, n, BY, BX, Y0, X0, lds0, tl_smem, wtab, is, sl, acc);
#pragma unroll 1
  for (; s < a.n_in && a.lf[s] == 2; ++s) tl_source<CPL, 4, C>(a, s, n, BY, BX, Y0, X0, lds0, tl_smem, wtab, is, sl, acc);
#pragma unroll 1
  for (; s < a.n_in && a.lf[s] == 1; ++s) tl_source<CPL, 2, C>(a, s, n, BY, BX, Y0, X0, lds0, tl_smem, wtab, is, sl, acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the repeated last items

  const int cc = c0 < C ? c0 : C - CPL;
  float bi[CPL], sc[CPL], sh[CPL], hw[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    bi[q] = a.bias ? a.bias[cc + q] : 0.f;
    sc[q] = a.scale ? a.scale[cc + q] : 1.f;
    sh[q] = a.scale ? a.shift[cc + q] : 0.f;
    hw[q] = c0 < C ? a.head_w[c0 + q] : 0.f;
  }
  // the head's bias, read ONCE and pinned: left to itself hipcc re-reads it from memory before every pixel's sigmoid (the stores in between
  // might alias it) -- 32 exposed memory latencies per wavefront
  float hb = a.head_b[0];
  asm volatile("" : "+v"(hb));
  float* __restrict__ hout = a.head_out + (static_cast<long>(n) * a.H + Y0) * a.W + X0;
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t2 = 0.f;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        float v = (acc[d][e][q] + bi[q]) * sc[q] + sh[q];
        if (a.act == DIFFSAL_ACT_RELU) v = fmaxf(v, 0.f);
        t2 = fmaf(v, hw[q], t2);
      }
      t2 = group_sum<32>(t2);
      if (li == 0) hout[d * a.W + e] = sigmoidf_(t2 + hb);
      __builtin_amdgcn_sched_barrier(0);       // one pixel at a time: interleaved, the 32 reductions need ~100 registers more than the loops
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
    a.lf[i] = 0;
    while ((a.h[i] << (a.lf[i] + 1)) <= H) ++a.lf[i];
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
  // mt_proj form: head folded, dilation 1, fp32, H and W multiples of 4, every source at a factor 2 .. 64
  bool rows = head_out && dil == 1 && dtype == DIFFSAL_F32 && tune(TUNE_NO_TAPSUM_ROWS) != 1 && H % 4 == 0 && W % 4 == 0;
  for (int i = 0; i < n_src && rows; ++i)
    rows = H / hs[i] >= 2 && H / hs[i] <= 64 && static_cast<long>(N) * hs[i] * ws[i] * 9 * C * 4 < (1L << 32) - (1L << 20);
  if (rows) {
    // DIFFSAL_TAPSUM_ROWS_FORM unset: the LDS-staged kernel where it applies (C = 96: three channels per lane; H and W multiples of 16;
    // a source image below 2 GiB; sources coarsest first); 1 .. 4: the row-streamed kernel's lane mappings
    {
      bool lds_ok = tune(TUNE_TAPSUM_ROWS_FORM) < 0 && H % 16 == 0 && W % 16 == 0 && C == 96;     // 10 columns x 72 pieces <= 768 per item
      for (int i = 0; i < n_src && lds_ok; ++i)      // coarsest first: the kernel walks the factor classes >= 8, 4, 2 in that order
        lds_ok = static_cast<long>(hs[i]) * ws[i] * 9 * C * 4 < (1L << 31) && (i == 0 || hs[i] >= hs[i - 1]);
      if (lds_ok) {
        const long blocks = static_cast<long>(N) * (H / 16) * (W / 16);
        DS_REQUIRE(blocks < (1L << 31), DIFFSAL_E_SHAPE, "tapsum: output too large");
        const size_t lds = kTlDepth * kTlSlot + 4 * 64 * sizeof(float);
        DS_RAISE_DYNAMIC_LDS((tapsum_head_lds_kernel<3, 96>), 160 * 1024);
        hipLaunchKernelGGL((tapsum_head_lds_kernel<3, 96>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, static_cast<hipStream_t>(stream), a, W / 16, H / 16);
        return check_launch("tapsum(head, LDS-staged)");
      }
    }
    // lane mapping (DIFFSAL_TAPSUM_ROWS_FORM): unset -- an 8 x 4 patch per 32-lane half, 3 or 4 channels per lane, the halves of a
    // wavefront two adjacent patches of one image (H, W multiples of 8), else = 3: the same patch of two images (H a multiple of 8), else
    // = 1: 4 x 4 patches of two images; = 2: one image per wavefront (2 channels per lane, C <= 128, 4 x 4: twice the wavefronts at 177
    // instead of 272 registers: measured equal at 4 clips, slower at 64)
    const int form = tune(TUNE_TAPSUM_ROWS_FORM);
    const bool tall = form != 1 && form != 2 && H % 8 == 0;
    const bool sx = tall && form != 3 && W % 8 == 0;
    const int row_map = form == 4 ? 1 : 0;
    const int imgs = form == 2 || sx ? 1 : 2;
    const int ps = tall ? 8 : 4, pw = sx ? 8 : 4;
    const long n_items = static_cast<long>((N + imgs - 1) / imgs) * (H / ps) * (W / pw);
    const dim3 grid(static_cast<unsigned>((n_items + 3) / 4));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool three = C % 3 == 0 && C / 3 <= 32 && C / 3 > 24;
#define TS_HEAD(...) hipLaunchKernelGGL((tapsum_head_rows_kernel<__VA_ARGS__>), grid, dim3(256), 0, st, a, W / pw, n_items, row_map)
    if (form == 2 && C % 2 == 0) TS_HEAD(2, 1, 4);
    else if (sx && three) TS_HEAD(3, 2, 8, true);
    else if (sx) TS_HEAD(4, 2, 8, true);
    else if (tall && three) TS_HEAD(3, 2, 8);
    else if (tall) TS_HEAD(4, 2, 8);
    else if (three) TS_HEAD(3, 2, 4);
    else TS_HEAD(4, 2, 4);
#undef TS_HEAD
    return check_launch("tapsum(head)");
  }
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
