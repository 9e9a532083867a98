// 3x3 stride-1 "same" convolutions (dilation d = padding, d in {1, 2}) of the fp32 denoiser as Winograd F(4x4, 3x3): 36
// multiplications per 4x4 output tile, input channel and output channel instead of 144 -- the matrix pipe does 4x fewer MFMAs
// than the direct convolution (F(2x2, 3x3) of csrc/wino.hip: 2.25x fewer).  Same layers as that file: ResnetBlock.conv1 / conv2
// (R/models/saliency_decoder/sal_unet.py:104-142) and UpEmbed's second convolution (common_block.py:196-216).
//
//   Y = A^T [ sum_ci (G g G^T) o (B^T x B) ] A        per tile; o = element-wise over the 36 positions xi = (i, j)
//
// interpolation points 0, +-1, +-2, infinity (Lavin & Gray 2015, the standard F(4x4, 3x3) matrices).  The transforms multiply by
// up to 8 and cancel: results differ from the direct fp32 convolution by ~1e-5 of the output maximum (F(2x2): ~1e-6), two
// orders inside the 1e-3 parity bar; DIFFSAL_NO_WINOGRAD4=1 keeps F(2x2) / the direct kernel.
//
// Unlike wino.hip this path is NOT fused: a 36-position accumulator set for a 64 x 64 tile block is 590 KB, more than the 512 KB
// of vector registers of a CU, and a block that fits (32 x 64) needs 12 TB/s of operand traffic to keep the matrix pipe busy.
// The positions are 36 independent plain products [tiles, Cin] x [Cout, Cin]^T instead -- exactly what gemm_dma_kernel is built
// for (96 x 96 tiles, LDS-DMA ring, 0.7-0.85 of the fp32 matrix peak) -- run as ONE batched launch between two streaming kernels:
//
//  1. wino4_input_kernel    V[xi][tile][ci] = B^T x B        2.25x the input bytes (F(2x2): 4x)
//  2. gemm_dma_kernel       M[xi][tile][co] = V[xi] U[xi]^T   batch of 36, K = Cin; U = G g G^T [36][Cout][Cin] from the host
//  3. wino4_output_kernel   Y = A^T M A + the usual epilogue (bias, BN affine, per-image vector, activation, residual)
//
// A dilated convolution is d*d undilated ones on the polyphase sub-grids: a tile is (image, py, px, ty, tx) and covers outputs
// (py + d(4 ty + a), px + d(4 tx + b)), a, b in 0..3; its inputs are rows py + d(4 ty + i - 1), i in 0..5.
#include "common.h"

namespace diffsal {

int gemm_dma_batched(const float* a, const float* w, float* out, long M, int K, int N, int batch, long a_bs, long w_bs, long o_bs,
                     hipStream_t s, const float* side_a, const float* side_w, float* side_out, long side_rows);

struct Wino4Geom {
  int N, H, W, Cin, Cout, d;
  int TY, TX;          // tiles per sub-grid
  int n_tiles;         // N * d*d * TY * TX
  // extended-grid form (d == 1, padding 2): outputs on (H + 2) x (W + 2), output (oy, ox) = the convolution centred on input
  // (oy - 1, ox - 1) -- the convolution evaluated one pixel beyond the map on every side (diffsal_up2_conv_commute needs it)
  int e, HO, WO;
  int xcd_walk;        // input transforms: XCD-aware workgroup order (DIFFSAL_NO_XCD_ORDER=1: dispatch order)
};

__device__ __forceinline__ void wino4_tile_coords(const Wino4Geom& g, int t, int& n, int& y0, int& x0) {
  const int per_class = g.TY * g.TX, per_img = g.d * g.d * per_class;
  n = t / per_img;
  const int r = t - n * per_img;
  const int cls = r / per_class, rr = r - cls * per_class;
  const int py = cls / g.d, px = cls - py * g.d;
  const int ty = rr / g.TX, tx = rr - ty * g.TX;
  y0 = py + g.d * 4 * ty;      // first output row of the tile; rows y0 + d a, a = 0..3; input rows y0 + d (i - 1), i = 0..5
  x0 = px + g.d * 4 * tx;
}

// The transform kernels work on VT = float4 / float2 / float channels per lane: small layers (a few thousand tile x channel-quad
// items) take the narrower types so that the launch still has enough lanes to hide its 36 dependent-free loads per lane.
__device__ __forceinline__ float4 w4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 w4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
// a + s b
__device__ __forceinline__ float4 w4_fma(float s, float4 b, float4 a) {
  return make_float4(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z), fmaf(s, b.w, a.w));
}
__device__ __forceinline__ float2 w4_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 w4_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 w4_fma(float s, float2 b, float2 a) { return make_float2(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y)); }
__device__ __forceinline__ float w4_add(float a, float b) { return a + b; }
__device__ __forceinline__ float w4_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float w4_fma(float s, float b, float a) { return fmaf(s, b, a); }
template <typename VT> __device__ __forceinline__ VT w4_zero();
template <> __device__ __forceinline__ float4 w4_zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float2 w4_zero<float2>() { return make_float2(0.f, 0.f); }
template <> __device__ __forceinline__ float w4_zero<float>() { return 0.f; }
template <typename VT> __device__ __forceinline__ VT w4_ld(const float* p) { return *reinterpret_cast<const VT*>(p); }
template <typename VT> __device__ __forceinline__ void w4_st(float* p, VT v) { *reinterpret_cast<VT*>(p) = v; }
template <typename VT> __device__ __forceinline__ float w4_get(const VT& v, int e);
template <> __device__ __forceinline__ float w4_get<float4>(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
template <> __device__ __forceinline__ float w4_get<float2>(const float2& v, int e) { return e == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float w4_get<float>(const float& v, int) { return v; }

// the interpolation / epilogue expressions of up2_conv_commute_kernel (csrc/upconv.hip), per component
__device__ __forceinline__ float w4_lerp1(float a, float b) { return 0.75f * a + 0.25f * b; }
__device__ __forceinline__ float4 w4_lerp(float4 a, float4 b) { return make_float4(w4_lerp1(a.x, b.x), w4_lerp1(a.y, b.y), w4_lerp1(a.z, b.z), w4_lerp1(a.w, b.w)); }
__device__ __forceinline__ float2 w4_lerp(float2 a, float2 b) { return make_float2(w4_lerp1(a.x, b.x), w4_lerp1(a.y, b.y)); }
__device__ __forceinline__ float w4_lerp(float a, float b) { return w4_lerp1(a, b); }
__device__ __forceinline__ float4 w4_affine(float4 v, float4 s, float4 h) { return make_float4(v.x * s.x + h.x, v.y * s.y + h.y, v.z * s.z + h.z, v.w * s.w + h.w); }
__device__ __forceinline__ float2 w4_affine(float2 v, float2 s, float2 h) { return make_float2(v.x * s.x + h.x, v.y * s.y + h.y); }
__device__ __forceinline__ float w4_affine(float v, float s, float h) { return v * s + h; }
__device__ __forceinline__ float4 w4_relu(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }
__device__ __forceinline__ float2 w4_relu(float2 v) { return make_float2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)); }
__device__ __forceinline__ float w4_relu(float v) { return fmaxf(v, 0.f); }

// B^T applied to six values (rows of B^T: [4 0 -5 0 1 0], [0 -4 -4 1 1 0], [0 4 -4 -1 1 0], [0 -2 -1 2 1 0], [0 2 -1 -2 1 0],
// [0 4 0 -5 0 1])
template <typename VT>
__device__ __forceinline__ void w4_bt(const VT (&d)[6], VT (&t)[6]) {
  t[0] = w4_fma(4.f, d[0], w4_fma(-5.f, d[2], d[4]));
  const VT a = w4_fma(-4.f, d[2], d[4]), b = w4_fma(-4.f, d[1], d[3]);
  t[1] = w4_add(a, b);
  t[2] = w4_sub(a, b);
  const VT c = w4_sub(d[4], d[2]), e = w4_sub(d[3], d[1]);
  t[3] = w4_fma(2.f, e, c);
  t[4] = w4_fma(-2.f, e, c);
  t[5] = w4_fma(4.f, d[1], w4_fma(-5.f, d[3], d[5]));
}

// ------------------------------------------------------------------------------------------------------------------------
// 1. input transform.  item = (tile, channel quad), quads fastest: a wave reads 16-byte pieces of consecutive channels of one
//    pixel (128-byte runs and longer) and writes, per position, consecutive channels of consecutive tiles (fully coalesced).
// ------------------------------------------------------------------------------------------------------------------------
// y = a x + b per channel, then (swish) y sigmoid(y): GroupNorm's affine form + the ResnetBlock's nonlinearity applied as the
// transform loads its input (R/models/saliency_decoder/sal_unet.py:36-44, 125, 133)
__device__ __forceinline__ float w4_gn1(float x, float a, float b, int swish) {
  const float y = fmaf(a, x, b);
  return swish ? swishf_fast(y) : y;
}
__device__ __forceinline__ float4 w4_gn(float4 x, float4 a, float4 b, int sw) {
  return make_float4(w4_gn1(x.x, a.x, b.x, sw), w4_gn1(x.y, a.y, b.y, sw), w4_gn1(x.z, a.z, b.z, sw), w4_gn1(x.w, a.w, b.w, sw));
}
__device__ __forceinline__ float2 w4_gn(float2 x, float2 a, float2 b, int sw) {
  return make_float2(w4_gn1(x.x, a.x, b.x, sw), w4_gn1(x.y, a.y, b.y, sw));
}
__device__ __forceinline__ float w4_gn(float x, float a, float b, int sw) { return w4_gn1(x, a, b, sw); }

// GN: the input is read through a per-(image, channel) affine map ab [N][2][Cin] (scale row, shift row: diffsal_gn_affine) and
// an optional swish -- GroupNorm + nonlinearity without a normalised tensor in memory; the convolution's zero padding applies
// to the NORMALISED map, so taps outside the image stay exactly zero.
// XCD-aware workgroup index (round 6): workgroups are dealt round-robin over the eight XCDs; give XCD x the contiguous run of virtual
// indices that starts at its share, so that the tiles whose 6 x 6 input windows overlap are transformed on ONE XCD and the shared input
// columns / rows come out of its L2 (a bijection of [0, gridDim.x) for any grid size)
__device__ __forceinline__ unsigned w4_xcd_block() {
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xcd = bid & 7u, slot = bid >> 3, q = nwg >> 3, r = nwg & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <typename VT, bool GN>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, float* __restrict__ V, Wino4Geom g,
                                                          const float* __restrict__ ab, int swish) {
  constexpr int VW = sizeof(VT) / 4;
  const int q4n = g.Cin / VW;
  const long items = static_cast<long>(g.n_tiles) * q4n;
  const long pos_stride = static_cast<long>(g.n_tiles) * g.Cin;
  for (long it = static_cast<long>(g.xcd_walk ? w4_xcd_block() : blockIdx.x) * 256 + threadIdx.x; it < items; it += static_cast<long>(gridDim.x) * 256) {
    const int t = static_cast<int>(it / q4n);
    const int q4 = static_cast<int>(it - static_cast<long>(t) * q4n);
    int n, y0, x0;
    wino4_tile_coords(g, t, n, y0, x0);
    y0 -= g.e; x0 -= g.e;
    const float* base = x + static_cast<long>(n) * g.H * g.W * g.Cin + q4 * VW;
    VT ga = w4_zero<VT>(), gb = w4_zero<VT>();
    if constexpr (GN) {
      ga = w4_ld<VT>(ab + static_cast<long>(n) * 2 * g.Cin + q4 * VW);
      gb = w4_ld<VT>(ab + static_cast<long>(n) * 2 * g.Cin + g.Cin + q4 * VW);
    }
    VT tt[6][6];
    // columns first: tt[.][j] = B^T (column j of the 6 x 6 patch)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int xx = x0 + g.d * (j - 1);
      const bool vx = xx >= 0 && xx < g.W;
      VT dd[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int y = y0 + g.d * (i - 1);
        const bool v = vx && y >= 0 && y < g.H;
        VT r = w4_ld<VT>(base + (v ? (static_cast<long>(y) * g.W + xx) : 0) * g.Cin);
        if constexpr (GN) r = w4_gn(r, ga, gb, swish);
        dd[i] = v ? r : w4_zero<VT>();
      }
      VT c[6];
      w4_bt(dd, c);
#pragma unroll
      for (int i = 0; i < 6; ++i) tt[i][j] = c[i];
    }
    float* dst = V + static_cast<long>(t) * g.Cin + q4 * VW;
#pragma unroll
    for (int i = 0; i < 6; ++i) {      // (.) B : rows
      VT o[6];
      w4_bt(tt[i], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) w4_st<VT>(dst + (i * 6 + j) * pos_stride, o[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// 1b. input transform of UpEmbed's SECOND convolution (dilation 2) reading the source-resolution result of the FIRST one:
//     u1 = act(BN(I0_y I0_x c)) is formed from c = conv3x3(z) on the extended source grid [N][h + 2][w + 2][C] as the patch is
//     gathered (the arithmetic of up2_conv_commute_kernel, expression for expression; equal up to FMA contraction), so the interior of u1
//     -- 74 MB at stage 3 -- is neither written nor read back; the 3-pixel border ring, whose values need the tap products, is
//     read from `u1` where diffsal_up2_conv_commute_ring left it.  R/models/saliency_decoder/common_block.py:196-216.
//     Per patch column: the 7 source rows of the two source columns (14 loads), horizontal then vertical interpolation.
// ------------------------------------------------------------------------------------------------------------------------
struct Wino4Up2 {
  const float* c;        // [N][h + 2][w + 2][C]
  const float* u1;       // [N][2h][2w][C]: ring pixels valid
  const float* scale;    // BatchNorm affine of the first convolution (may be null)
  const float* shift;
  int h, w, act;
};

template <typename VT>
__global__ __launch_bounds__(256) void wino4_input_up2_kernel(Wino4Up2 u, float* __restrict__ V, Wino4Geom g) {
  constexpr int VW = sizeof(VT) / 4;
  const int q4n = g.Cin / VW;
  const long items = static_cast<long>(g.n_tiles) * q4n;
  const long pos_stride = static_cast<long>(g.n_tiles) * g.Cin;
  const int H2 = g.H, W2 = g.W;                       // = 2 h, 2 w
  const long crow = static_cast<long>(u.w + 2) * g.Cin;
  for (long it = static_cast<long>(g.xcd_walk ? w4_xcd_block() : blockIdx.x) * 256 + threadIdx.x; it < items; it += static_cast<long>(gridDim.x) * 256) {
    const int t = static_cast<int>(it / q4n);
    const int q4 = static_cast<int>(it - static_cast<long>(t) * q4n);
    int n, y0, x0;
    wino4_tile_coords(g, t, n, y0, x0);
    const float* cimg = u.c + static_cast<long>(n) * (u.h + 2) * crow + q4 * VW;
    const float* uimg = u.u1 + static_cast<long>(n) * H2 * W2 * g.Cin + q4 * VW;
    VT sc, sh;
    if (u.scale) { sc = w4_ld<VT>(u.scale + q4 * VW); sh = w4_ld<VT>(u.shift + q4 * VW); }
    // rows of the patch: y = y0 + 2 (i - 1), one parity; source row m_i = y >> 1, far row m_i + 1 (odd y) or m_i - 1 (even y).
    // The seven source rows involved: r = 0..6 <-> source row mb + r, mb = (first patch row >> 1) - (even parity ? 1 : 0)
    const int yf = y0 - 2, sy = yf & 1;
    const int mb = (yf >> 1) - (sy ? 0 : 1);          // >> on a negative yf: arithmetic shift = floor, as the rows outside are masked anyway
    VT tt[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int xx = x0 + 2 * (j - 1);
      const bool vx = xx >= 0 && xx < W2;
      const int sx = xx & 1, nx = xx >> 1;
      const int xc = nx + 1, xf_ = (sx ? nx + 1 : nx - 1) + 1;      // centre / far column on the extended grid (+ 1)
      const bool ringx = xx < 3 || xx >= W2 - 3;
      VT hrow[7];
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        const int cr = mb + r + 1;                   // row on the extended grid
        const bool ok = vx && cr >= 0 && cr < u.h + 2;
        const float* rp = cimg + (ok ? static_cast<long>(cr) * crow : 0);
        const VT a0 = w4_ld<VT>(rp + (ok ? static_cast<long>(xc) * g.Cin : 0)), a1 = w4_ld<VT>(rp + (ok ? static_cast<long>(xf_) * g.Cin : 0));
        hrow[r] = w4_lerp(a0, a1);                    // 0.75 a0 + 0.25 a1, the expression of up2_conv_commute_kernel
      }
      VT dd[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int y = yf + 2 * i;
        const bool v = vx && y >= 0 && y < H2;
        const bool ring = ringx || y < 3 || y >= H2 - 3;
        // centre row m = y >> 1 <-> r = (y >> 1) - mb; far row: + 1 (odd) / - 1 (even)
        // (selects, not hrow[runtime index]: that would put the array in scratch)
        const VT hc = sy ? hrow[i] : hrow[i + 1], hf = sy ? hrow[i + 1] : hrow[i];
        VT val = w4_lerp(hc, hf);
        if (u.scale) val = w4_affine(val, sc, sh);
        if (u.act == DIFFSAL_ACT_RELU) val = w4_relu(val);
        if (v && ring) val = w4_ld<VT>(uimg + (static_cast<long>(y) * W2 + xx) * g.Cin);
        dd[i] = v ? val : w4_zero<VT>();
      }
      VT cc[6];
      w4_bt(dd, cc);
#pragma unroll
      for (int i = 0; i < 6; ++i) tt[i][j] = cc[i];
    }
    float* dst = V + static_cast<long>(t) * g.Cin + q4 * VW;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      VT o[6];
      w4_bt(tt[i], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) w4_st<VT>(dst + (i * 6 + j) * pos_stride, o[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// 3. output transform + epilogue.  item = (tile, channel quad), quads fastest.
// ------------------------------------------------------------------------------------------------------------------------
struct Wino4Out {
  const float* M;
  float* out;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const float* residual;
  int act, rowvec_ld;
  Wino4Geom g;
  // STATS: per-(image, group) sum and sum of squares of the RESULT (after the epilogue) for the GroupNorm that follows
  // (ResnetBlock.norm2 on conv1's output, sal_unet.py:131-133): every workgroup leaves the sums of its 256 items in
  // stats[blockIdx.x][2 images][groups][2] (fp64; the items of a workgroup touch at most two images), diffsal_gn_affine_wino4
  // adds them up in a fixed order
  double* stats;
  int groups, tiles_per_img;
};

__device__ __forceinline__ float w4_act(float x, int act) {
  if (act == DIFFSAL_ACT_RELU) return fmaxf(x, 0.f);
  if (act == DIFFSAL_ACT_GELU_ERF) return gelu_erf(x);
  if (act == DIFFSAL_ACT_SIGMOID) return sigmoidf_(x);
  return x;
}

// A^T applied to six values (rows of A^T: [1 1 1 1 1 0], [0 1 -1 2 -2 0], [0 1 1 4 4 0], [0 1 -1 8 -8 1])
template <typename VT>
__device__ __forceinline__ void w4_at(const VT (&m)[6], VT (&y)[4]) {
  const VT s12 = w4_add(m[1], m[2]), d12 = w4_sub(m[1], m[2]), s34 = w4_add(m[3], m[4]), d34 = w4_sub(m[3], m[4]);
  y[0] = w4_add(w4_add(m[0], s12), s34);
  y[1] = w4_fma(2.f, d34, d12);
  y[2] = w4_fma(4.f, s34, s12);
  y[3] = w4_add(w4_fma(8.f, d34, d12), m[5]);
}

constexpr int kW4StatGroups = 64;
template <typename VT, bool STATS>
__global__ __launch_bounds__(256) void wino4_output_kernel(Wino4Out p) {
  constexpr int VW = sizeof(VT) / 4;
  const Wino4Geom& g = p.g;
  const int q4n = g.Cout / VW;
  const long items = static_cast<long>(g.n_tiles) * q4n;
  const long pos_stride = static_cast<long>(g.n_tiles) * g.Cout;
  const long HW = static_cast<long>(g.HO) * g.WO;
  // STATS: every thread leaves the per-channel sums of its item in LDS; the workgroup's (image, group) sums are then added
  // up in a FIXED order (tile, channel) -- no atomics, the statistics are bit-reproducible from run to run
  __shared__ float2 st_sh[STATS ? 256 * VW : 1];
  int n_first = 0;
  if constexpr (STATS) {          // one pass per workgroup (grid = ceil(items / 256)): its statistics slot is blockIdx.x
#pragma unroll
    for (int e = 0; e < VW; ++e) st_sh[threadIdx.x * VW + e] = make_float2(0.f, 0.f);    // threads past the last item
    n_first = static_cast<int>((static_cast<long>(blockIdx.x) * 256) / q4n) / p.tiles_per_img;
  }
  for (long it = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; it < items; it += static_cast<long>(gridDim.x) * 256) {
    const int t = static_cast<int>(it / q4n);
    const int co = static_cast<int>(it - static_cast<long>(t) * q4n) * VW;
    int n, y0, x0;
    wino4_tile_coords(g, t, n, y0, x0);
    const float* src = p.M + static_cast<long>(t) * g.Cout + co;
    VT s[4][6];                     // A^T M : columns j of M, rows a of the result
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      VT m[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) m[i] = w4_ld<VT>(src + (i * 6 + j) * pos_stride);
      VT y[4];
      w4_at(m, y);
#pragma unroll
      for (int a = 0; a < 4; ++a) s[a][j] = y[a];
    }
    float bb[VW], ss[VW], hh[VW], rr[VW];
    float ssum[VW], ssq[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      bb[e] = p.bias ? p.bias[co + e] : 0.f;
      ss[e] = p.scale ? p.scale[co + e] : 1.f;
      hh[e] = p.scale ? p.shift[co + e] : 0.f;
      rr[e] = p.rowvec ? p.rowvec[static_cast<long>(n) * p.rowvec_ld + co + e] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int oy = y0 + g.d * a;
      VT y[4];
      w4_at(s[a], y);                   // (.) A
      if (oy >= g.HO) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ox = x0 + g.d * b;
        if (ox >= g.WO) continue;
        const long o = (static_cast<long>(n) * HW + static_cast<long>(oy) * g.WO + ox) * g.Cout + co;
        float v[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          float xv = w4_get<VT>(y[b], e) + bb[e];
          if (p.scale) xv = xv * ss[e] + hh[e];
          xv += rr[e];
          v[e] = w4_act(xv, p.act);
        }
        if (p.residual) {
          const VT rs = w4_ld<VT>(p.residual + o);
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] += w4_get<VT>(rs, e);
        }
        if constexpr (STATS) {
#pragma unroll
          for (int e = 0; e < VW; ++e) { ssum[e] += v[e]; ssq[e] = fmaf(v[e], v[e], ssq[e]); }
        }
        VT ov;
        if constexpr (VW == 4) ov = make_float4(v[0], v[1], v[2], v[3]);
        else if constexpr (VW == 2) ov = make_float2(v[0], v[1]);
        else ov = v[0];
        w4_st<VT>(p.out + o, ov);
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int e = 0; e < VW; ++e) st_sh[threadIdx.x * VW + e] = make_float2(ssum[e], ssq[e]);
    }
  }
  if constexpr (STATS) {
    __syncthreads();
    // thread i = (image slot, group, sum | sum of squares): the workgroup's tiles of that image in order, the group's channels in
    // order.  Item of (tile t, channel c) = t q4n + c / VW, element c % VW; this workgroup holds items [base, base + 256)
    const int cpg = g.Cout / p.groups;
    const long base = static_cast<long>(blockIdx.x) * 256;
    const int t_lo = static_cast<int>(base / q4n);
    const long last = base + 255 < items - 1 ? base + 255 : items - 1;
    const int t_hi = static_cast<int>(last / q4n);
    for (int i = threadIdx.x; i < 2 * p.groups * 2; i += 256) {
      const int img = i / (p.groups * 2), r = i - img * (p.groups * 2);
      const int grp = r >> 1, which = r & 1;
      double acc = 0.0;
      for (int t = t_lo; t <= t_hi; ++t) {
        if (t / p.tiles_per_img != n_first + img) continue;
        for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) {
          const long item = static_cast<long>(t) * q4n + c / VW;
          if (item < base || item > last) continue;
          const float2 v = st_sh[static_cast<int>(item - base) * VW + (c % VW)];
          acc += static_cast<double>(which ? v.y : v.x);
        }
      }
      p.stats[(static_cast<long>(blockIdx.x) * 2 + img) * (p.groups * 2) + r] = acc;
    }
  }
}

// ab [N][2][C] from the workgroup sums wino4_output_kernel<.., true> left: the workgroups whose 256 items touch image n are
// w in [n ipi / 256, ((n + 1) ipi - 1) / 256] (ipi = items per image), slot = n - (first image of w).  One workgroup per image;
// 8 lanes per group take every 8th partial, combined in a fixed order, fp64 mean / variance (E[x^2] - mean^2, as gn_apply_kernel).
__global__ __launch_bounds__(256) void wino4_gn_finalize_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ ab, int C,
                                                                int groups, long ipi, int q4n, int tiles_per_img, double count, float eps) {
  // grid (N, ceil(groups / 4)): a wavefront per group, its 64 lanes take every 64th workgroup sum (one round trip instead of a
  // chain of dependent ones), butterfly in a fixed order
  const int n = blockIdx.x, grp = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (grp >= groups) return;
  const long w_lo = (n * ipi) / 256, w_hi = ((n + 1) * ipi - 1) / 256;
  double s = 0, q = 0;
  for (long w = w_lo + lane; w <= w_hi; w += 64) {
    const int nf = static_cast<int>((w * 256) / q4n) / tiles_per_img;
    const double* o = stats + ((w * 2 + (n - nf)) * groups + grp) * 2;
    s += o[0]; q += o[1];
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  const double mean = s / count;
  double var = q / count - mean * mean;
  var = var < 0 ? 0 : var;
  const float mf = static_cast<float>(mean), rf = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  const int cpg = C / groups;
  for (int c = grp * cpg + lane; c < (grp + 1) * cpg; c += 64) {
    const float sc = rf * gamma[c];
    ab[static_cast<long>(n) * 2 * C + c] = sc;
    ab[static_cast<long>(n) * 2 * C + C + c] = beta[c] - mf * sc;
  }
}

namespace {

bool wino4_shape_ok(const diffsal_conv_desc* d) {
  if (!d) return false;
  const bool same = d->pad_t == d->dil_h && d->pad_l == d->dil_w && d->Ho == d->H && d->Wo == d->W;
  const bool ext = d->dil_h == 1 && d->pad_t == 2 && d->pad_l == 2 && d->Ho == d->H + 2 && d->Wo == d->W + 2;
  return d->KH == 3 && d->KW == 3 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == d->dil_w &&
         (d->dil_h == 1 || d->dil_h == 2) && (same || ext) &&
         d->Cin > 0 && d->Cin % 96 == 0 && d->Cout > 0 && d->Cout % 4 == 0 && d->N > 0 && d->H > 0 && d->W > 0 &&
         d->dtype == DIFFSAL_F32 && d->precision == DIFFSAL_PREC_FP32;
}

Wino4Geom wino4_geom(const diffsal_conv_desc* d) {
  Wino4Geom g{};
  g.N = d->N; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout; g.d = d->dil_h;
  g.e = d->Ho == d->H + 2 ? 1 : 0;
  g.xcd_walk = tune(TUNE_NO_XCD_ORDER) == 1 ? 0 : 1;
  g.HO = d->Ho; g.WO = d->Wo;
  g.TY = ((g.HO + g.d - 1) / g.d + 3) / 4;
  g.TX = ((g.WO + g.d - 1) / g.d + 3) / 4;
  g.n_tiles = g.N * g.d * g.d * g.TY * g.TX;
  return g;
}

constexpr long kW4Lanes = 65536;       // lanes a transform launch should have before it takes wider pieces per lane

size_t wino4_v_bytes(const Wino4Geom& g) { return 36ul * g.n_tiles * g.Cin * sizeof(float); }
size_t wino4_m_bytes(const Wino4Geom& g) { return 36ul * g.n_tiles * g.Cout * sizeof(float); }

}  // namespace
// U = G g G^T of every (output, input) channel pair on the device (training: the weights change every step; inference packs U once on
// the host in fp64, ops.pack_wino4_weight).  w [Cout][Cin][3][3] (the parameter's layout).  dgrad = 0: U [36][Cout][Cin] -- the forward
// convolution; dgrad = 1: U [36][Cin][Cout] of the flipped kernel g'[a][b] = g[2-a][2-b] -- the data gradient dX = conv(dY, w^T flipped).
// The fastest thread index is the output's last dimension (coalesced stores; the nine-float reads are strided either way).
__global__ __launch_bounds__(256) void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int dgrad) {
  const long n = static_cast<long>(Cout) * Cin;
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int inner = dgrad ? Cout : Cin;
  const int o = static_cast<int>(i / inner), c = static_cast<int>(i - static_cast<long>(o) * inner);     // output row / column of U's [.][.]
  const int co = dgrad ? c : o, ci = dgrad ? o : c;
  const float* gp = w + (static_cast<long>(co) * Cin + ci) * 9;
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g[a][b] = dgrad ? gp[(2 - a) * 3 + (2 - b)] : gp[a * 3 + b];
  constexpr float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                             {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  float t[6][3];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int b = 0; b < 3; ++b) t[r][b] = G[r][0] * g[0][b] + G[r][1] * g[1][b] + G[r][2] * g[2][b];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int q = 0; q < 6; ++q) U[(r * 6 + q) * n + i] = t[r][0] * G[q][0] + t[r][1] * G[q][1] + t[r][2] * G[q][2];
}

}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_wino4_weight(const float* w, float* U, int Cout, int Cin, int dgrad, diffsal_stream_t stream) {
  DS_REQUIRE(w && U, DIFFSAL_E_ARG, "wino4_weight: null argument");
  DS_REQUIRE(Cout > 0 && Cin > 0 && static_cast<long>(Cout) * Cin < (1L << 31), DIFFSAL_E_SHAPE, "wino4_weight: bad shape %d x %d", Cout, Cin);
  const long n = static_cast<long>(Cout) * Cin;
  hipLaunchKernelGGL(wino4_weight_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, U,
                     Cout, Cin, dgrad ? 1 : 0);
  return check_launch("wino4_weight");
}

// 1 when diffsal_conv_wino4 accepts the descriptor AND the planner expects it to beat both the F(2x2) path and the direct kernel
extern "C" int diffsal_conv_wino4_supported(const diffsal_conv_desc* d) {
  if (!wino4_shape_ok(d) || tune(TUNE_NO_WINOGRAD) == 1 || tune(TUNE_NO_WINOGRAD4) == 1) return 0;
  if (tune(TUNE_FORCE_WINOGRAD) == 1) return 1;
  const Wino4Geom g = wino4_geom(d);
  // the transformed input and the position products (36 x tiles x (Cin + Cout) floats) are written and read back once: they have
  // to stay in the 256 MB Infinity Cache for the two streaming kernels to cost less than the products they save.  From 24 tiles
  // (one 14 x 24 map): alone such a convolution was slower than F(2x2) (round 4: 0.71-0.83x, hence a bar of 96 tiles), but the
  // fused ResnetBlock rides on this path (GroupNorm in the input transform, shortcut in the batched launch, statistics from the
  // output transform: three launches fewer per block) -- measured per step: B = 1 582 -> 602 steps/s, B = 2 877 -> 888
  return (wino4_v_bytes(g) + wino4_m_bytes(g) <= 400ul * 1000 * 1000 && d->Cout >= 96 && g.n_tiles >= 24) ? 1 : 0;
}

extern "C" size_t diffsal_conv_wino4_ws_bytes(const diffsal_conv_desc* d) {
  if (!wino4_shape_ok(d)) return 0;
  const Wino4Geom g = wino4_geom(d);
  return wino4_v_bytes(g) + wino4_m_bytes(g);
}


namespace {
// channels per lane of the transform kernels for `scalars` = tiles x channels: wide pieces when there are plenty of items, else
// narrower ones (>= ~128 K lanes keep 256 CUs busy)
int wino4_vw(long scalars) { return scalars >= 4 * kW4Lanes ? 4 : (scalars >= 2 * kW4Lanes ? 2 : 1); }

// statistics of the output: a workgroup's 256 items must touch at most two images
bool wino4_stats_ok(const Wino4Geom& g, int groups) {
  if (groups <= 0 || groups > kW4StatGroups || g.Cout % groups != 0 || g.e != 0) return false;
  const int vw = wino4_vw(static_cast<long>(g.n_tiles) * g.Cout);
  const long ipi = static_cast<long>(g.n_tiles / g.N) * (g.Cout / vw);
  return ipi >= 256;
}
long wino4_out_wgs(const Wino4Geom& g) {
  const int vw = wino4_vw(static_cast<long>(g.n_tiles) * g.Cout);
  return (static_cast<long>(g.n_tiles) * (g.Cout / vw) + 255) / 256;
}
}  // namespace

/* bytes of the `out_stats` buffer of diffsal_conv_wino4_ex for this convolution and group count; 0: not available for the shape */
extern "C" size_t diffsal_conv_wino4_stats_bytes(const diffsal_conv_desc* d, int groups) {
  if (!wino4_shape_ok(d)) return 0;
  const Wino4Geom g = wino4_geom(d);
  if (!wino4_stats_ok(g, groups)) return 0;
  return static_cast<size_t>(wino4_out_wgs(g)) * 2 * groups * 2 * sizeof(double);
}

/* 1 if a plain product with `side_rows` rows can share the launch of this convolution's position products */
extern "C" int diffsal_conv_wino4_side_supported(const diffsal_conv_desc* d, long side_rows) {
  if (!wino4_shape_ok(d) || side_rows <= 0) return 0;
  const Wino4Geom g = wino4_geom(d);
  return (side_rows % g.n_tiles == 0 && side_rows / g.n_tiles <= 4096) ? 1 : 0;
}

// U: the transformed weight G g G^T as [36][Cout][Cin] fp32 (ops.pack_wino4_weight).  stages: bit 0 input transform, bit 1 the
// position products, bit 2 output transform + epilogue -- the three launches of the path, callable one by one (same arguments,
// same workspace) so that a profiler can bracket each: diffsal_conv_wino4 is stages = 7.  ext (may be NULL): GroupNorm affine +
// swish applied to the input as it is transformed, a plain product riding in the launch of the position products, statistics
// of the result for the next GroupNorm (include/diffsal.h).
extern "C" int diffsal_conv_wino4_ex(const diffsal_conv_desc* d, const float* x, const float* U, const float* bias, const float* scale,
                                     const float* shift, const float* rowvec, const float* residual, float* out, void* ws,
                                     size_t ws_bytes, const diffsal_wino4_ext* ext, int stages, diffsal_stream_t stream) {
  DS_REQUIRE(d && x && U && out && ws, DIFFSAL_E_ARG, "conv_wino4: null argument");
  DS_REQUIRE(wino4_shape_ok(d), DIFFSAL_E_SHAPE,
             "conv_wino4: fp32 3x3 stride-1 convolutions with padding = dilation in {1, 2} (or dilation 1, padding 2, output (H + 2) x "
             "(W + 2)), Cin %% 96 == 0, Cout %% 4 == 0 only");
  DS_REQUIRE((scale == nullptr) == (shift == nullptr), DIFFSAL_E_ARG, "conv_wino4: scale and shift come together");
  const Wino4Geom g = wino4_geom(d);
  const size_t vb = wino4_v_bytes(g), mb = wino4_m_bytes(g);
  DS_REQUIRE(ws_bytes >= vb + mb, DIFFSAL_E_ARG, "conv_wino4: workspace of %zu bytes, need %zu", ws_bytes, vb + mb);
  DS_REQUIRE(aligned16(x) && aligned16(U) && aligned16(out) && aligned16(ws) && (!bias || aligned16(bias)) &&
                 (!scale || (aligned16(scale) && aligned16(shift))) && (!rowvec || aligned16(rowvec)) &&
                 (!residual || aligned16(residual)) && (!rowvec || d->rowvec_ld % 4 == 0),
             DIFFSAL_E_ALIGN, "conv_wino4: misaligned pointer");
  DS_REQUIRE(static_cast<long>(d->N) * d->H * d->W * d->Cin < (1L << 31) && static_cast<long>(d->N) * d->Ho * d->Wo * d->Cout < (1L << 31),
             DIFFSAL_E_SHAPE, "conv_wino4: tensor too large");
  diffsal_wino4_ext e{};
  if (ext) e = *ext;
  DS_REQUIRE(!e.in_ab || aligned16(e.in_ab), DIFFSAL_E_ALIGN, "conv_wino4: misaligned in_ab");
  DS_REQUIRE(!e.up2_c || (!e.in_ab && g.d == 2 && g.e == 0 && g.H % 2 == 0 && g.W % 2 == 0 && g.H >= 4 && g.W >= 4 && aligned16(e.up2_c) &&
                          (e.up2_scale == nullptr) == (e.up2_shift == nullptr) && (!e.up2_scale || (aligned16(e.up2_scale) && aligned16(e.up2_shift))) &&
                          (e.up2_act == DIFFSAL_ACT_NONE || e.up2_act == DIFFSAL_ACT_RELU)),
             DIFFSAL_E_ARG, "conv_wino4: up2 input needs dilation 2, even H and W >= 4, no in_ab, act NONE / RELU");
  DS_REQUIRE(e.side_rows == 0 || (e.side_a && e.side_w && e.side_out && diffsal_conv_wino4_side_supported(d, e.side_rows)), DIFFSAL_E_ARG,
             "conv_wino4: side product of %lld rows does not fit (rows must be a multiple of the %d tiles)", e.side_rows, g.n_tiles);
  DS_REQUIRE(!e.out_stats || (wino4_stats_ok(g, e.out_groups) && aligned16(e.out_stats)), DIFFSAL_E_ARG,
             "conv_wino4: output statistics not available for this shape (groups=%d)", e.out_groups);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* V = static_cast<float*>(ws);
  float* Mb = reinterpret_cast<float*>(static_cast<char*>(ws) + vb);
  DS_REQUIRE(stages >= 1 && stages <= 7, DIFFSAL_E_ARG, "conv_wino4: stages=%d", stages);
  if (stages & 1) {
    const long scalars = static_cast<long>(g.n_tiles) * g.Cin;
    const int vw = wino4_vw(scalars);
    long gi = (scalars / vw + 255) / 256;
    gi = gi > 16384 ? 16384 : gi;
    const dim3 gd(static_cast<unsigned>(gi));
    const Wino4Up2 up{e.up2_c, x, e.up2_scale, e.up2_shift, g.H / 2, g.W / 2, e.up2_act};
#define W4_IN(VT)                                                                                                   \
  do {                                                                                                              \
    if (e.up2_c) hipLaunchKernelGGL((wino4_input_up2_kernel<VT>), gd, dim3(256), 0, s, up, V, g);                   \
    else if (e.in_ab) hipLaunchKernelGGL((wino4_input_kernel<VT, true>), gd, dim3(256), 0, s, x, V, g, e.in_ab, e.in_swish); \
    else hipLaunchKernelGGL((wino4_input_kernel<VT, false>), gd, dim3(256), 0, s, x, V, g, nullptr, 0);             \
  } while (0)
    if (vw == 4) W4_IN(float4);
    else if (vw == 2) W4_IN(float2);
    else W4_IN(float);
#undef W4_IN
    const int rc = check_launch("conv_wino4(input transform)");
    if (rc) return rc;
  }
  if (stages & 2) {
    const int r = gemm_dma_batched(V, U, Mb, g.n_tiles, g.Cin, g.Cout, 36, static_cast<long>(g.n_tiles) * g.Cin,
                                   static_cast<long>(g.Cout) * g.Cin, static_cast<long>(g.n_tiles) * g.Cout, s, e.side_a, e.side_w,
                                   e.side_out, e.side_rows);
    if (r < 0) return r;
    DS_REQUIRE(r == 1, DIFFSAL_E_SHAPE, "conv_wino4: the position products do not fit the batched GEMM kernel");
    if (stages == 2) return DIFFSAL_OK;          // the kernel's own name stays (diffsal_last_gemm_kernel)
  }
  if (!(stages & 4)) return DIFFSAL_OK;
  Wino4Out a{};
  a.M = Mb; a.out = out; a.bias = bias; a.scale = scale; a.shift = shift; a.rowvec = rowvec; a.residual = residual;
  a.act = d->act; a.rowvec_ld = d->rowvec_ld; a.g = g;
  a.stats = e.out_stats; a.groups = e.out_groups; a.tiles_per_img = g.n_tiles / g.N;
  const long scalars = static_cast<long>(g.n_tiles) * g.Cout;
  const int vw = wino4_vw(scalars);
  long go = (scalars / vw + 255) / 256;
  if (!a.stats) go = go > 16384 ? 16384 : go;    // with statistics: one pass per workgroup, its slot is blockIdx.x
  const dim3 gd(static_cast<unsigned>(go));
#define W4_OUT(VT)                                                                                   \
  do {                                                                                               \
    if (a.stats) hipLaunchKernelGGL((wino4_output_kernel<VT, true>), gd, dim3(256), 0, s, a);        \
    else hipLaunchKernelGGL((wino4_output_kernel<VT, false>), gd, dim3(256), 0, s, a);               \
  } while (0)
  if (vw == 4) W4_OUT(float4);
  else if (vw == 2) W4_OUT(float2);
  else W4_OUT(float);
#undef W4_OUT
  if (stages == 7)
    note_kernel("wino4_input_kernel + 36 x gemm_dma_kernel<float, 3, 3, 3, 2, false, false> + wino4_output_kernel [F(4x4,3x3): 36 x (M=%d K=%d N=%d)]",
                g.n_tiles, g.Cin, g.Cout);
  return check_launch("conv_wino4(output transform)");
}

/* ab [N][2][Cout] (scale row, shift row) of GroupNorm(groups, eps; gamma, beta) over the OUTPUT of the convolution `d`, from the
 * workgroup sums diffsal_conv_wino4_ex left in `stats` (ext.out_stats, same d and groups). */
extern "C" int diffsal_gn_affine_wino4(const diffsal_conv_desc* d, const double* stats, const float* gamma, const float* beta, int groups,
                                       float eps, float* ab, diffsal_stream_t stream) {
  DS_REQUIRE(d && stats && gamma && beta && ab, DIFFSAL_E_ARG, "gn_affine_wino4: null argument");
  DS_REQUIRE(wino4_shape_ok(d), DIFFSAL_E_SHAPE, "gn_affine_wino4: not a convolution of the F(4x4) path");
  const Wino4Geom g = wino4_geom(d);
  DS_REQUIRE(wino4_stats_ok(g, groups), DIFFSAL_E_SHAPE, "gn_affine_wino4: statistics not available for this shape (groups=%d)", groups);
  const int vw = wino4_vw(static_cast<long>(g.n_tiles) * g.Cout);
  const int q4n = g.Cout / vw, tpi = g.n_tiles / g.N;
  const double count = static_cast<double>(g.HO) * g.WO * (g.Cout / groups);
  hipLaunchKernelGGL(wino4_gn_finalize_kernel, dim3(g.N, (groups + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), stats, gamma, beta, ab, g.Cout,
                     groups, static_cast<long>(tpi) * q4n, q4n, tpi, count, eps);
  return check_launch("gn_affine_wino4");
}

extern "C" int diffsal_conv_wino4_stages(const diffsal_conv_desc* d, const float* x, const float* U, const float* bias, const float* scale,
                                         const float* shift, const float* rowvec, const float* residual, float* out, void* ws,
                                         size_t ws_bytes, int stages, diffsal_stream_t stream) {
  return diffsal_conv_wino4_ex(d, x, U, bias, scale, shift, rowvec, residual, out, ws, ws_bytes, nullptr, stages, stream);
}

extern "C" int diffsal_conv_wino4(const diffsal_conv_desc* d, const float* x, const float* U, const float* bias, const float* scale,
                                  const float* shift, const float* rowvec, const float* residual, float* out, void* ws,
                                  size_t ws_bytes, diffsal_stream_t stream) {
  return diffsal_conv_wino4_ex(d, x, U, bias, scale, shift, rowvec, residual, out, ws, ws_bytes, nullptr, 7, stream);
}
