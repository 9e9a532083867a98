// Weight-gradient of the implicit-GEMM convolution / linear layer, and the bias (column-sum) gradient.
//
//   dW[co, k] = sum_m dY[m, co] * A[m, k],   A = im2col view of the layer input (never materialised),
//   k = (ci/32, tap, ci%32): dW comes out directly in the packed layout the forward kernel consumes
//   (ops.pack_conv_weight); PyTorch's autograd un-permutes it back to the parameter layout.
//
// GEMM view: C[co][k] with the reduction over the pixel index m.  Both operands are m-major in memory, so the
// LDS tiles are [m][c] and every MFMA operand is one ds_read_b32 (32 consecutive channels of one m per half
// wave: conflict-free).  The M range is split over gridDim.z; partial tiles go to slabs that a second kernel
// sums in a fixed order (deterministic, no atomics).  Replaces the wgrad half of autograd's conv/linear
// backward for every call site listed in diffsal.h (training step).
#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
  const float* in;   // layer input, NHWC
  const float* dy;   // [M][Cout]
  float* slabs;      // [splits][Cout][K]
  int M, K, Cout;
  int H, W, Cin, Ho, Wo;
  int KW, taps, stride_h, stride_w, pad_t, pad_l, dil_h, dil_w;
  int rows_per_split;
  unsigned in_bytes;
};

constexpr int WG_BCO = 128;   // co tile (2 waves x 2 MFMA tiles)
constexpr int WG_SL = 6;      // K slices of 32 per tile (2 waves x 3 MFMA tiles)
constexpr int WG_BKI = WG_SL * 32;
constexpr int WG_BM = 32;     // pixels per reduction step
constexpr int WG_PA = WG_BCO + 4;
constexpr int WG_PB = WG_BKI + 4;

__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) float dYs[WG_BM * WG_PA];
  __shared__ __attribute__((aligned(16))) float Xs[WG_BM * WG_PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves over (co, k)
  const int co0 = blockIdx.x * WG_BCO;
  const int sl0 = blockIdx.y * WG_SL;       // first K slice of this tile
  const int n_slices = p.K / 32;
  const int m_begin = blockIdx.z * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int HoWo = p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in), 0, static_cast<int>(p.in_bytes), 0x00020000);

  // per-slice tap displacement (wave-uniform)
  int sl_dy[WG_SL], sl_dx[WG_SL];
  unsigned sl_delta[WG_SL];
#pragma unroll
  for (int s = 0; s < WG_SL; ++s) {
    const int kt = min(sl0 + s, n_slices - 1);
    const int chunk = kt / p.taps, tap = kt - chunk * p.taps;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    sl_dy[s] = ky * p.dil_h;
    sl_dx[s] = kx * p.dil_w;
    sl_delta[s] = static_cast<unsigned>((sl_dy[s] * p.W + sl_dx[s]) * p.Cin + chunk * 32) * 4u;
  }

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // loader mapping: dY tile 32 x 128 floats = 1024 float4 (4 per thread); X tile 32 x 192 = 1536 float4 (6 per thread)
  for (int mb = m_begin; mb < m_end; mb += WG_BM) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;        // 0..1023
      const int r = idx >> 5, c4 = (idx & 31) * 4;
      const int m = mb + r, co = co0 + c4;
      float4 v = make_float4(0, 0, 0, 0);
      if (m < m_end && co < p.Cout) v = ld4(p.dy + static_cast<long>(m) * p.Cout + co);  // Cout % 4 == 0
      st4(&dYs[r * WG_PA + c4], v);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int idx = tid + 256 * q;        // 0..1535
      const int r = idx / 48, rem = idx - r * 48;
      const int s = rem >> 3, c4 = (rem & 7) * 4;
      const int m = mb + r;
      unsigned off = 0xFFFFFFFFu;
      if (m < m_end && sl0 + s < n_slices) {
        const int n = m / HoWo, rm = m - n * HoWo;
        const int oy = rm / p.Wo, ox = rm - oy * p.Wo;
        const int iy0 = oy * p.stride_h - p.pad_t, ix0 = ox * p.stride_w - p.pad_l;
        const int iy = iy0 + sl_dy[s], ix = ix0 + sl_dx[s];
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
          off = static_cast<unsigned>(((n * p.H + iy0) * p.W + ix0) * p.Cin + c4) * 4u + sl_delta[s];
      }
      const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
      st4(&Xs[r * WG_PB + s * 32 + c4], v);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < WG_BM / 2; ++ks) {
      const int mrow = ks * 2 + (lane >> 5);
      float a[2], b[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = dYs[mrow * WG_PA + (wm * 2 + i) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = Xs[mrow * WG_PB + (wn * 3 + j) * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D map: col = lane & 31 (k within slice), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (co within tile)
  float* slab = p.slabs + static_cast<long>(blockIdx.z) * p.Cout * p.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int sl = sl0 + wn * 3 + j;
      if (sl >= n_slices) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Cout) slab[static_cast<long>(co) * p.K + sl * 32 + (lane & 31)] = acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                       long n, int splits) {
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long>(gridDim.x) * 256) {
    float s = slabs[i];
    for (int k = 1; k < splits; ++k) s += slabs[static_cast<long>(k) * n + i];
    out[i] = s;
  }
}

// out[g, c] = sum over rows m in segment g (seg_rows rows each) of act'(y[m,c]) * dy[m, c]  (bias / per-image grads)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, double* __restrict__ part, int M,
                                                     int C, int seg_rows, int chunks) {
  // grid: (chunks, segments); block covers a row chunk of one segment; threads stride (row, c4).
  // fp32 per thread over a few rows, fp64 across threads and workgroups (bias gradients cancel heavily).
  extern __shared__ double shd[];  // [C]
  const int seg = blockIdx.y, chunk = blockIdx.x;
  const int c4n = C >> 2;
  const int rpp = 256 / c4n > 0 ? 256 / c4n : 1;
  const int c4 = threadIdx.x % c4n, rs = threadIdx.x / c4n;
  const long r0 = static_cast<long>(seg) * seg_rows;
  const long rb = r0 + static_cast<long>(seg_rows) * chunk / chunks;
  const long re = min(static_cast<long>(M), r0 + static_cast<long>(seg_rows) * (chunk + 1) / chunks);
  float4 s = make_float4(0, 0, 0, 0);
  if (rs < rpp)
    for (long m = rb + rs; m < re; m += rpp) {
      const float4 v = ld4(dy + m * C + c4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  for (int i = threadIdx.x; i < C; i += 256) shd[i] = 0.0;
  __syncthreads();
  if (rs < rpp) {
    atomicAdd(&shd[c4 * 4 + 0], static_cast<double>(s.x)); atomicAdd(&shd[c4 * 4 + 1], static_cast<double>(s.y));
    atomicAdd(&shd[c4 * 4 + 2], static_cast<double>(s.z)); atomicAdd(&shd[c4 * 4 + 3], static_cast<double>(s.w));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) part[(static_cast<long>(seg) * chunks + chunk) * C + i] = shd[i];
}

// out[g][c] (fp32) = sum over chunks of part[g][chunk][c] (fp64)
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, float* __restrict__ out,
                                                           int segs, int chunks, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= segs * C) return;
  const int g = i / C, c = i - g * C;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += part[(static_cast<long>(g) * chunks + k) * C + c];
  out[i] = static_cast<float>(s);
}

}  // namespace diffsal

using namespace diffsal;

extern "C" size_t diffsal_conv_wgrad_ws_bytes(const diffsal_conv_desc* d) {
  if (!d || d->Cin <= 0 || d->Cin % 32) return 0;
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const long K = static_cast<long>(d->KH) * d->KW * d->Cin;
  const long tiles = ((d->Cout + WG_BCO - 1) / WG_BCO) * ((K / 32 + WG_SL - 1) / WG_SL);
  long splits = (1024 + tiles - 1) / tiles;
  const long max_splits = (M + 4 * WG_BM - 1) / (4 * WG_BM);
  splits = splits < 1 ? 1 : (splits > max_splits ? max_splits : splits);
  splits = splits > 256 ? 256 : splits;
  return static_cast<size_t>(splits) * d->Cout * K * sizeof(float);
}

extern "C" int diffsal_conv_wgrad(const diffsal_conv_desc* d, const float* in, const float* dy, float* dw_packed,
                                  void* ws, size_t ws_bytes, diffsal_stream_t stream) {
  DS_REQUIRE(d && in && dy && dw_packed && ws, DIFFSAL_E_ARG, "conv_wgrad: null argument");
  DS_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0 && d->Cout % 4 == 0, DIFFSAL_E_SHAPE,
             "conv_wgrad: Cin=%d must be a multiple of 32 and Cout=%d of 4", d->Cin, d->Cout);
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const long K = static_cast<long>(d->KH) * d->KW * d->Cin;
  const long in_bytes = static_cast<long>(d->N) * d->H * d->W * d->Cin * 4;
  DS_REQUIRE(M > 0 && M < (1L << 31) && in_bytes < (1L << 32) - 16 && d->KH * d->KW <= 32, DIFFSAL_E_SHAPE,
             "conv_wgrad: problem too large");
  const size_t need = diffsal_conv_wgrad_ws_bytes(d);
  DS_REQUIRE(ws_bytes >= need && aligned16(ws) && aligned16(dy) && aligned16(in), DIFFSAL_E_ARG,
             "conv_wgrad: needs %zu bytes of 16-byte aligned workspace", need);
  const int splits = static_cast<int>(need / (static_cast<size_t>(d->Cout) * K * sizeof(float)));
  WgradArgs a;
  a.in = in; a.dy = dy; a.slabs = static_cast<float*>(ws);
  a.M = static_cast<int>(M); a.K = static_cast<int>(K); a.Cout = d->Cout;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KW = d->KW; a.taps = d->KH * d->KW; a.stride_h = d->stride_h; a.stride_w = d->stride_w;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.dil_h = d->dil_h; a.dil_w = d->dil_w;
  a.rows_per_split = static_cast<int>(((M + splits - 1) / splits + WG_BM - 1) / WG_BM * WG_BM);
  a.in_bytes = static_cast<unsigned>(in_bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((d->Cout + WG_BCO - 1) / WG_BCO, (static_cast<int>(K / 32) + WG_SL - 1) / WG_SL, splits);
  hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, s, a);
  int rc = check_launch("conv_wgrad");
  if (rc) return rc;
  const long n = static_cast<long>(d->Cout) * K;
  long g = (n + 255) / 256;
  g = g > 2048 ? 2048 : g;
  hipLaunchKernelGGL(slab_sum_kernel, dim3(static_cast<int>(g)), dim3(256), 0, s, static_cast<const float*>(ws),
                     dw_packed, n, splits);
  return check_launch("conv_wgrad(sum)");
}

extern "C" int diffsal_colsum(const float* dy, float* out, int M, int C, int seg_rows, void* ws, size_t ws_bytes,
                              diffsal_stream_t stream) {
  DS_REQUIRE(dy && out && ws, DIFFSAL_E_ARG, "colsum: null argument");
  DS_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 4096 && seg_rows > 0 && M % seg_rows == 0, DIFFSAL_E_SHAPE,
             "colsum: bad shape M=%d C=%d seg_rows=%d", M, C, seg_rows);
  const int segs = M / seg_rows;
  int chunks = 2048 / segs;
  chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
  while (chunks > 1 && seg_rows / chunks < 8) chunks >>= 1;
  DS_REQUIRE(ws_bytes >= static_cast<size_t>(segs) * chunks * C * sizeof(double), DIFFSAL_E_ARG,
             "colsum: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(colsum_kernel, dim3(chunks, segs), dim3(256), C * sizeof(double), s, dy, static_cast<double*>(ws), M,
                     C, seg_rows, chunks);
  int rc = check_launch("colsum");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((segs * C + 255) / 256), dim3(256), 0, s,
                     static_cast<const double*>(ws), out, segs, chunks, C);
  return check_launch("colsum(sum)");
}
