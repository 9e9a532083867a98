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
#include <cstdlib>

#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
  const float* in;   // layer input, NHWC
  const float* dy;   // [M][Cout]
  float* slabs;      // [segments * splits][Cout][K]
  double* dbias;     // optional [segments * splits][Cout]: column sums of dY over the split's rows (bias gradient partials)
  float* dbias_out;  // optional [Cout]: the finished bias gradient (written here when there is one split, else by slab_sum_kernel)
  int M, K, Cout;
  int H, W, Cin, Ho, Wo;
  int KW, taps, stride_h, stride_w, pad_t, pad_l, dil_h, dil_w;
  int seg_rows, splits, rows_per_split;   // blockIdx.z = segment * splits + split
  unsigned in_bytes;
  unsigned dy_bytes;   // M * Cout * 4 (< 4 GiB): dY goes through a buffer descriptor too, so that its tail loads need no branch
};

constexpr int WG_BM = 64;     // granularity of the M split (a multiple of every kernel's rows-per-step)

// Tile = (WCO*CT*32 output channels) x (WK*ST K-slices of 32); the 4 waves form a WCO x WK grid and each owns
// CT x ST MFMA 32x32 accumulators.  Global loads of step i+1 are issued into registers before the MFMA loop of
// step i, so HBM/L2 latency hides behind 16 * CT * ST MFMAs per wave.
template <int CT, int WCO, int WK, int ST, int BM = 32>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs p) {
  static_assert(WCO * WK == 4, "four waves");
  constexpr int BCO = WCO * CT * 32, SL = WK * ST, BKI = SL * 32;
  // row pitch = 32 (mod 64) floats: the two half waves of a ds_read_b32 (rows m and m + 1, 32 consecutive floats each) use
  // disjoint halves of the 64 banks; the float4 stores stay row-linear.  (+4 padding: 1-2 % slower, tools/tune_wgrad_mvit.py)
  constexpr int PA = BCO + (96 - BCO % 64) % 64, PB = BKI + (96 - BKI % 64) % 64;
  constexpr int NA = BCO / 32 * (BM / 32), NB = BKI / 32 * (BM / 32);    // float4 loads per thread per step
  __shared__ __attribute__((aligned(16))) float dYs[BM * PA];
  __shared__ __attribute__((aligned(16))) float Xs[BM * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WK, wn = wave % WK;
  const int co0 = blockIdx.x * BCO;
  const int sl0 = blockIdx.y * SL;          // first K slice of this tile
  const int n_slices = p.K / 32;
  const int seg = blockIdx.z / p.splits, sp = blockIdx.z - seg * p.splits;
  // the splits of a segment interleave at step granularity (split sp takes steps sp, sp + splits, ...): at any moment the
  // resident workgroups stream through adjacent chunks of dY / X (halos shared in L2) instead of marching in lockstep at a
  // fixed large stride.  (Measured neutral on MI355X; kept because it bounds the working set.)
  const int m_begin = seg * p.seg_rows + sp * BM;
  const int m_end = min(p.M, (seg + 1) * p.seg_rows);
  const int m_stride = p.splits * BM;
  const int HoWo = p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in), 0, static_cast<int>(p.in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.dy), 0, static_cast<int>(p.dy_bytes), 0x00020000);

  f32x16 acc[CT][ST];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < ST; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // loader mapping: dY tile 32 x BCO floats (NA float4 per thread); X tile 32 x BKI floats (NB float4 per thread).
  // Per-load constants that do not depend on the step: tile row, K slice -> tap displacement.
  int xr[NB], xdy[NB], xdx[NB];
  unsigned xdelta[NB];
  bool xok[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int idx = tid + 256 * q;
    const int r = idx / (BKI / 4), rem = idx - r * (BKI / 4);
    const int s = rem >> 3, c4 = (rem & 7) * 4;
    const int kt = min(sl0 + s, n_slices - 1);
    const int chunk = kt / p.taps, tap = kt - chunk * p.taps;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    xr[q] = r;
    xdy[q] = ky * p.dil_h;
    xdx[q] = kx * p.dil_w;
    xdelta[q] = static_cast<unsigned>((xdy[q] * p.W + xdx[q]) * p.Cin + chunk * 32 + c4) * 4u;
    xok[q] = sl0 + s < n_slices;
  }
  float4 ra[NA], rb[NB];
  auto prefetch = [&](int mb) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int idx = tid + 256 * q;
      const int r = idx / (BCO / 4), c4 = (idx - r * (BCO / 4)) * 4;
      const int m = mb + r, co = co0 + c4;
      // branch-free (a conditional load costs a branch and a full vmcnt drain per piece): rows past the split's end and
      // columns past Cout read zeros through the descriptor's range check.  Cout % 4 == 0.
      const unsigned off = (m < m_end && co < p.Cout) ? static_cast<unsigned>(m * p.Cout + co) * 4u : 0xFFFFFFFFu;
      ra[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, off, 0, 0));
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int m = mb + xr[q];
      unsigned off = 0xFFFFFFFFu;
      if (m < m_end && xok[q]) {
        const int n = m / HoWo, rm = m - n * HoWo;
        const int oy = rm / p.Wo, ox = rm - oy * p.Wo;
        const int iy0 = oy * p.stride_h - p.pad_t, ix0 = ox * p.stride_w - p.pad_l;
        const int iy = iy0 + xdy[q], ix = ix0 + xdx[q];
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
          off = static_cast<unsigned>(((n * p.H + iy0) * p.W + ix0) * p.Cin) * 4u + xdelta[q];
      }
      rb[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
  };

  // bias gradient rides along: the first K-tile's workgroups add up the dY tile they stage anyway (fp64 per thread)
  const bool do_bias = p.dbias != nullptr && blockIdx.y == 0 && tid < BCO;
  double bias_acc = 0.0;
  if (m_begin < m_end) prefetch(m_begin);
  for (int mb = m_begin; mb < m_end; mb += m_stride) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int idx = tid + 256 * q;
      const int r = idx / (BCO / 4), c4 = (idx - r * (BCO / 4)) * 4;
      st4(&dYs[r * PA + c4], ra[q]);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int idx = tid + 256 * q;
      const int r = idx / (BKI / 4), rem = idx - r * (BKI / 4);
      st4(&Xs[r * PB + rem * 4], rb[q]);
    }
    __syncthreads();
    if (mb + m_stride < m_end) prefetch(mb + m_stride);
    if (do_bias) {
      float s4 = 0.f;
#pragma unroll
      for (int r = 0; r < BM; ++r) s4 += dYs[r * PA + tid];
      bias_acc += static_cast<double>(s4);
    }
#pragma unroll
    for (int ks = 0; ks < BM / 2; ++ks) {
      const int mrow = ks * 2 + (lane >> 5);
      float a[CT], b[ST];
#pragma unroll
      for (int i = 0; i < CT; ++i) a[i] = dYs[mrow * PA + (wm * CT + i) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < ST; ++j) b[j] = Xs[mrow * PB + (wn * ST + j) * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < ST; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  if (do_bias && co0 + tid < p.Cout) {
    p.dbias[static_cast<long>(blockIdx.z) * p.Cout + co0 + tid] = bias_acc;
    if (p.dbias_out && gridDim.z == 1) p.dbias_out[co0 + tid] = static_cast<float>(bias_acc);
  }
  // C/D map: col = lane & 31 (k within slice), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (co within tile)
  float* slab = p.slabs + static_cast<long>(blockIdx.z) * p.Cout * p.K;
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < ST; ++j) {
      const int sl = sl0 + wn * ST + j;
      if (sl >= n_slices) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * CT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Cout) slab[static_cast<long>(co) * p.K + sl * 32 + (lane & 31)] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same tile product for PLAIN products (one tap: token GEMMs, 107 of the 124 weight gradients of a training step) with both
// operands staged by LDS-DMA (buffer_load ... lds, as csrc/gemm_dma.hip): no staging registers, no ds_write pass, three stages
// in flight and ONE barrier per 32-row step (placed between the two halves of the step's MFMAs) instead of a single stage
// between two barriers.  A step's tiles are [32 rows][BCO] of dY and [32 rows][BKI] of X, row-linear (a ds_read_b32 of a half
// wave covers 32 consecutive floats of one row: conflict-free on the unpadded pitch); a DMA piece is 1 KiB in lane order, the
// lane's (row, column) fixed for the kernel.  Rows past the split's end, columns past Cout / K: offset 0xFFFFFFFF, the DMA
// writes zeros.  Same accumulation order per output element as wgrad_kernel (rows in ascending order inside a split).
// ---------------------------------------------------------------------------------------------------------------------------
typedef int wg_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wgrad_dma_piece(unsigned lds_addr, unsigned voff, wg_i32x4 rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wgrad_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CT, int WCO, int WK, int ST>
__global__ __launch_bounds__(256, 1) void wgrad_dma_kernel(WgradArgs p) {
  static_assert(WCO * WK == 4, "four waves");
  constexpr int BM = 32, STAGES = 3, P = STAGES - 1;
  constexpr int BCO = WCO * CT * 32, SL = WK * ST, BKI = SL * 32;
  constexpr int A_F = BM * BCO, B_F = BM * BKI, STAGE_F = A_F + B_F;          // floats
  constexpr int APW = A_F / 256 / 4, BPW = B_F / 256 / 4, PPW = APW + BPW;     // 1 KiB pieces per wave and step
  static_assert(A_F % 1024 == 0 && B_F % 1024 == 0, "whole pieces per wave");
  constexpr int WAIT_N = (P - 2 > 0 ? P - 2 : 0) * PPW;
  __shared__ __attribute__((aligned(16))) float smem[STAGES * STAGE_F];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WK, wn = wave % WK;
  const int co0 = blockIdx.x * BCO;
  const int sl0 = blockIdx.y * SL;
  const int n_slices = p.K / 32;
  const int seg = blockIdx.z / p.splits, sp = blockIdx.z - seg * p.splits;
  const int m_begin = seg * p.seg_rows + sp * BM;
  const int m_end = min(p.M, (seg + 1) * p.seg_rows);
  const int m_stride = p.splits * BM;
  const int n_steps = m_begin < m_end ? (m_end - m_begin + m_stride - 1) / m_stride : 0;

  const unsigned long pdy = reinterpret_cast<unsigned long>(p.dy), pin = reinterpret_cast<unsigned long>(p.in);
  const wg_i32x4 rs_dy = {static_cast<int>(pdy), static_cast<int>(pdy >> 32) & 0xFFFF, static_cast<int>(p.dy_bytes), 0x00020000};
  const wg_i32x4 rs_x = {static_cast<int>(pin), static_cast<int>(pin >> 32) & 0xFFFF, static_cast<int>(p.in_bytes), 0x00020000};
  // this lane's slot in every piece it issues: (row of the step, byte offset inside the operand's row), ~0 when the column is
  // outside the matrix
  int a_row[APW], b_row[BPW];
  unsigned a_col[APW], b_col[BPW];
#pragma unroll
  for (int q = 0; q < APW; ++q) {
    const int idx = (wave + 4 * q) * 64 + lane;
    a_row[q] = idx / (BCO / 4);
    const int c = co0 + (idx - a_row[q] * (BCO / 4)) * 4;
    a_col[q] = c < p.Cout ? static_cast<unsigned>(c) * 4u : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int q = 0; q < BPW; ++q) {
    const int idx = (wave + 4 * q) * 64 + lane;
    b_row[q] = idx / (BKI / 4);
    const int k = sl0 * 32 + (idx - b_row[q] * (BKI / 4)) * 4;
    b_col[q] = k < p.K ? static_cast<unsigned>(k) * 4u : 0xFFFFFFFFu;
  }
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)smem)) + wave * 1024u;
  int iss_step = 0;
  auto issue_piece = [&](int stage, int q) __attribute__((always_inline)) {      // q < APW: dY, else X
    const int mb = m_begin + iss_step * m_stride;
    const unsigned st = lds_base + static_cast<unsigned>(stage * STAGE_F * 4);
    if (q < APW) {
      const int m = mb + a_row[q];
      const unsigned off = (m < m_end && iss_step < n_steps) ? static_cast<unsigned>(m) * static_cast<unsigned>(p.Cout * 4) + a_col[q] : 0xFFFFFFFFu;
      wgrad_dma_piece(st + q * 4096u, off | (a_col[q] == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u), rs_dy);
    } else {
      const int qb = q - APW;
      const int m = mb + b_row[qb];
      const unsigned off = (m < m_end && iss_step < n_steps) ? static_cast<unsigned>(m) * static_cast<unsigned>(p.K * 4) + b_col[qb] : 0xFFFFFFFFu;
      wgrad_dma_piece(st + (A_F + qb * 1024) * 4u, off | (b_col[qb] == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u), rs_x);
    }
  };

  f32x16 acc[CT][ST];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < ST; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_bias = p.dbias != nullptr && blockIdx.y == 0 && tid < BCO;
  double bias_acc = 0.0;

#pragma unroll
  for (int s = 0; s < P; ++s) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) issue_piece(s, q);
    ++iss_step;
  }
  wgrad_wait_vmcnt<(P - 1) * PPW>();
  __builtin_amdgcn_s_barrier();

  int stage = 0;
  for (int t = 0; t < n_steps; ++t) {
    const float* dYs = smem + stage * STAGE_F;
    const float* Xs = dYs + A_F;
    const int ist = stage == 0 ? STAGES - 1 : stage - 1;          // the stage freed by the previous step takes step t + P
    if (do_bias) {
      float s4 = 0.f;
#pragma unroll
      for (int r = 0; r < BM; ++r) s4 += dYs[r * BCO + tid];
      bias_acc += static_cast<double>(s4);
    }
    // fragments of k-step ks + 1 are read before the MFMAs of k-step ks (one wave per SIMD: nobody else covers the LDS latency)
    float fa[2][CT], fb[2][ST];
    auto read_frags = [&](int ks, int set) __attribute__((always_inline)) {
      const int mrow = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < CT; ++i) fa[set][i] = dYs[mrow * BCO + (wm * CT + i) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < ST; ++j) fb[set][j] = Xs[mrow * BKI + (wn * ST + j) * 32 + (lane & 31)];
    };
    auto mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < ST; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };
    read_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < BM / 4; ++ks) {
      read_frags(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // step t + 1 (this wave's pieces) has landed; after the barrier every wave's have, and nobody reads stage `ist` any more
    wgrad_wait_vmcnt<WAIT_N>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ks = BM / 4; ks < BM / 2; ++ks) {
      if (ks + 1 < BM / 2) read_frags(ks + 1, (ks + 1) & 1);
      const int h = ks - BM / 4;                                   // DMA pieces of step t + P between the second half's MFMAs
#pragma unroll
      for (int q = h * PPW / (BM / 4); q < (h + 1) * PPW / (BM / 4); ++q) issue_piece(ist, q);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    ++iss_step;
    stage = stage + 1 == STAGES ? 0 : stage + 1;
  }
  if (do_bias && co0 + tid < p.Cout) {
    p.dbias[static_cast<long>(blockIdx.z) * p.Cout + co0 + tid] = bias_acc;
    if (p.dbias_out && gridDim.z == 1) p.dbias_out[co0 + tid] = static_cast<float>(bias_acc);
  }
  float* slab = p.slabs + static_cast<long>(blockIdx.z) * p.Cout * p.K;
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < ST; ++j) {
      const int sl = sl0 + wn * ST + j;
      if (sl >= n_slices) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * CT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Cout) slab[static_cast<long>(co) * p.K + sl * 32 + (lane & 31)] = acc[i][j][r];
      }
    }
}

// Two stages (80 KB: TWO workgroups per CU, one's DMA issue, barriers and slab stores under the other's matrix work) instead of three:
// one barrier per 32-row step at the step boundary, the next step's pieces issued between this step's MFMAs (a step is 4096 matrix
// cycles: ample cover for the DMA latency).  Same products and order as wgrad_dma_kernel.
template <int CT, int WCO, int WK, int ST>
__global__ __launch_bounds__(256, 2) void wgrad_dma2_kernel(WgradArgs p) {
  static_assert(WCO * WK == 4, "four waves");
  constexpr int BM = 32, STAGES = 2;
  constexpr int BCO = WCO * CT * 32, SL = WK * ST, BKI = SL * 32;
  constexpr int A_F = BM * BCO, B_F = BM * BKI, STAGE_F = A_F + B_F;          // floats
  constexpr int APW = A_F / 256 / 4, BPW = B_F / 256 / 4, PPW = APW + BPW;     // 1 KiB pieces per wave and step
  static_assert(A_F % 1024 == 0 && B_F % 1024 == 0, "whole pieces per wave");
  __shared__ __attribute__((aligned(16))) float smem[STAGES * STAGE_F];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WK, wn = wave % WK;
  const int co0 = blockIdx.x * BCO;
  const int sl0 = blockIdx.y * SL;
  const int n_slices = p.K / 32;
  const int seg = blockIdx.z / p.splits, sp = blockIdx.z - seg * p.splits;
  const int m_begin = seg * p.seg_rows + sp * BM;
  const int m_end = min(p.M, (seg + 1) * p.seg_rows);
  const int m_stride = p.splits * BM;
  const int n_steps = m_begin < m_end ? (m_end - m_begin + m_stride - 1) / m_stride : 0;

  const unsigned long pdy = reinterpret_cast<unsigned long>(p.dy), pin = reinterpret_cast<unsigned long>(p.in);
  const wg_i32x4 rs_dy = {static_cast<int>(pdy), static_cast<int>(pdy >> 32) & 0xFFFF, static_cast<int>(p.dy_bytes), 0x00020000};
  const wg_i32x4 rs_x = {static_cast<int>(pin), static_cast<int>(pin >> 32) & 0xFFFF, static_cast<int>(p.in_bytes), 0x00020000};
  // this lane's slot in every piece it issues: (row of the step, byte offset inside the operand's row), ~0 when the column is
  // outside the matrix
  int a_row[APW], b_row[BPW];
  unsigned a_col[APW], b_col[BPW];
#pragma unroll
  for (int q = 0; q < APW; ++q) {
    const int idx = (wave + 4 * q) * 64 + lane;
    a_row[q] = idx / (BCO / 4);
    const int c = co0 + (idx - a_row[q] * (BCO / 4)) * 4;
    a_col[q] = c < p.Cout ? static_cast<unsigned>(c) * 4u : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int q = 0; q < BPW; ++q) {
    const int idx = (wave + 4 * q) * 64 + lane;
    b_row[q] = idx / (BKI / 4);
    const int k = sl0 * 32 + (idx - b_row[q] * (BKI / 4)) * 4;
    b_col[q] = k < p.K ? static_cast<unsigned>(k) * 4u : 0xFFFFFFFFu;
  }
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)smem)) + wave * 1024u;
  int iss_step = 0;
  auto issue_piece = [&](int stage, int q) __attribute__((always_inline)) {      // q < APW: dY, else X
    const int mb = m_begin + iss_step * m_stride;
    const unsigned st = lds_base + static_cast<unsigned>(stage * STAGE_F * 4);
    if (q < APW) {
      const int m = mb + a_row[q];
      const unsigned off = (m < m_end && iss_step < n_steps) ? static_cast<unsigned>(m) * static_cast<unsigned>(p.Cout * 4) + a_col[q] : 0xFFFFFFFFu;
      wgrad_dma_piece(st + q * 4096u, off | (a_col[q] == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u), rs_dy);
    } else {
      const int qb = q - APW;
      const int m = mb + b_row[qb];
      const unsigned off = (m < m_end && iss_step < n_steps) ? static_cast<unsigned>(m) * static_cast<unsigned>(p.K * 4) + b_col[qb] : 0xFFFFFFFFu;
      wgrad_dma_piece(st + (A_F + qb * 1024) * 4u, off | (b_col[qb] == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u), rs_x);
    }
  };

  f32x16 acc[CT][ST];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < ST; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_bias = p.dbias != nullptr && blockIdx.y == 0 && tid < BCO;
  double bias_acc = 0.0;

#pragma unroll
  for (int q = 0; q < PPW; ++q) issue_piece(0, q);
  ++iss_step;
  wgrad_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  for (int t = 0; t < n_steps; ++t) {
    const int stage = t & 1, ist = stage ^ 1;
    const float* dYs = smem + stage * STAGE_F;
    const float* Xs = dYs + A_F;
    if (do_bias) {
      float s4 = 0.f;
#pragma unroll
      for (int r = 0; r < BM; ++r) s4 += dYs[r * BCO + tid];
      bias_acc += static_cast<double>(s4);
    }
    float fa[2][CT], fb[2][ST];
    auto read_frags = [&](int ks, int set) __attribute__((always_inline)) {
      const int mrow = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < CT; ++i) fa[set][i] = dYs[mrow * BCO + (wm * CT + i) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < ST; ++j) fb[set][j] = Xs[mrow * BKI + (wn * ST + j) * 32 + (lane & 31)];
    };
    auto mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < ST; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };
    // the other stage is free (everybody passed the barrier that ended step t - 1): the pieces of step t + 1 go out between this step's
    // MFMAs and have the whole step to land
    read_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < BM / 2; ++ks) {
      if (ks + 1 < BM / 2) read_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int q = ks * PPW / (BM / 2); q < (ks + 1) * PPW / (BM / 2); ++q) issue_piece(ist, q);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    ++iss_step;
    // this wavefront's pieces of step t + 1 have landed and its reads of this stage have returned; after the barrier everybody's
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (do_bias && co0 + tid < p.Cout) {
    p.dbias[static_cast<long>(blockIdx.z) * p.Cout + co0 + tid] = bias_acc;
    if (p.dbias_out && gridDim.z == 1) p.dbias_out[co0 + tid] = static_cast<float>(bias_acc);
  }
  float* slab = p.slabs + static_cast<long>(blockIdx.z) * p.Cout * p.K;
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < ST; ++j) {
      const int sl = sl0 + wn * ST + j;
      if (sl >= n_slices) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * CT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Cout) slab[static_cast<long>(co) * p.K + sl * 32 + (lane & 31)] = acc[i][j][r];
      }
    }
}

// out[seg][i] = sum over the splits of one segment (deterministic: fixed association).  Workgroup = 16 float4 columns x
// 16 split groups: group g adds splits g, g+16, ... in order, then the 16 group sums are added in order.  (One thread per
// column walking all splits serially left 9 workgroups with 256-512 dependent steps each on the small token GEMMs.)
// Blocks past g_main (bias_out != NULL) finish the bias gradient that rode along: bias_out[c] = sum over the splits of the fp64
// partial column sums (16 split groups, then the group sums in order) -- no separate reduction launch per layer.
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                       long n, int splits, int g_main, const double* __restrict__ bias_part,
                                                       float* __restrict__ bias_out, int Cout) {
  if (static_cast<int>(blockIdx.x) >= g_main) {       // 16 columns x 16 split groups, the same fixed association as the slabs
    __shared__ double shb[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = (static_cast<int>(blockIdx.x) - g_main) * 16 + cl;
    double t = 0.0;
    if (blockIdx.y == 0 && c < Cout)
      for (int z = g; z < splits; z += 16) t += bias_part[static_cast<long>(z) * Cout + c];
    shb[g][cl] = t;
    __syncthreads();
    if (g == 0 && blockIdx.y == 0 && c < Cout) {
      double u = shb[0][cl];
#pragma unroll
      for (int q = 1; q < 16; ++q) u += shb[q][cl];
      bias_out[c] = static_cast<float>(u);
    }
    return;
  }
  __shared__ float4 sh[16][17];
  const float* base = slabs + static_cast<long>(blockIdx.y) * splits * n;
  float* o = out + static_cast<long>(blockIdx.y) * n;
  const long n4 = n >> 2;   // n = Cout * K, K % 32 == 0
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long i = static_cast<long>(blockIdx.x) * 16 + cl;
  float4 s = make_float4(0, 0, 0, 0);
  if (i < n4)
    for (int k = g; k < splits; k += 16) {
      const float4 v = ld4(base + static_cast<long>(k) * n + i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  sh[g][cl] = s;
  __syncthreads();
  if (g == 0 && i < n4) {
    float4 t = sh[0][cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) { t.x += sh[q][cl].x; t.y += sh[q][cl].y; t.z += sh[q][cl].z; t.w += sh[q][cl].w; }
    st4(o + i * 4, t);
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
  const int tpr = c4n < 256 ? c4n : 256;          // threads across a row; wider rows are walked in column blocks
  const int rpp = 256 / tpr;
  const int cc = threadIdx.x % tpr, rs = threadIdx.x / tpr;
  const long r0 = static_cast<long>(seg) * seg_rows;
  const long rb = r0 + static_cast<long>(seg_rows) * chunk / chunks;
  const long re = min(static_cast<long>(M), r0 + static_cast<long>(seg_rows) * (chunk + 1) / chunks);
  for (int i = threadIdx.x; i < C; i += 256) shd[i] = 0.0;
  __syncthreads();
  if (rs < rpp)
    for (int c4 = cc; c4 < c4n; c4 += tpr) {
      float4 s = make_float4(0, 0, 0, 0);
      for (long m = rb + rs; m < re; m += rpp) {
        const float4 v = ld4(dy + m * C + c4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      atomicAdd(&shd[c4 * 4 + 0], static_cast<double>(s.x)); atomicAdd(&shd[c4 * 4 + 1], static_cast<double>(s.y));
      atomicAdd(&shd[c4 * 4 + 2], static_cast<double>(s.z)); atomicAdd(&shd[c4 * 4 + 3], static_cast<double>(s.w));
    }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) part[(static_cast<long>(seg) * chunks + chunk) * C + i] = shd[i];
}

// out[g][c] (fp32) = sum over chunks of part[g][chunk][c] (fp64).  Workgroup = (8 channels, segment); 32 thread groups
// take every 32nd chunk, then the 32 group sums are added in a fixed order.
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, float* __restrict__ out,
                                                           int segs, int chunks, int C) {
  __shared__ double sh[32][9];
  const int cl = threadIdx.x & 7, kg = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl, g = blockIdx.y;
  double s = 0.0;
  if (c < C)
    for (int k = kg; k < chunks; k += 32) s += part[(static_cast<long>(g) * chunks + k) * C + c];
  sh[kg][cl] = s;
  __syncthreads();
  if (kg == 0 && c < C) {
    double t = sh[0][cl];
#pragma unroll
    for (int j = 1; j < 32; ++j) t += sh[j][cl];
    out[static_cast<long>(g) * C + c] = static_cast<float>(t);
  }
}

}  // namespace diffsal

using namespace diffsal;

namespace {

struct WgradCfg { int bco, sl, mfma_per_kstep; };
constexpr int kNumWgradCfgs = 5;
constexpr WgradCfg kWgradCfgs[kNumWgradCfgs] = {{128, 6, 6}, {96, 8, 6}, {64, 12, 6}, {32, 12, 3}, {96, 4, 3}};

struct WgradPlan { int cfg, splits, rows_per_split; };

// Pick the tile shape and the M split from a small cost model (cycles):
//  * a workgroup step (32 rows) is MFMA-bound at 16 k-steps x mfma_per_kstep x 64 cycles; two workgroups share a CU,
//    512 run at once, so `rounds` of 512 each take 2 x steps x step_cycles;
//  * every split writes and re-reads a Cout x K slab (~1250 B/cycle of HBM).
// A workgroup count just above a multiple of 512 costs a whole extra round -- the reason this is not a fixed target.
//  * plain products (one tap: token GEMMs): every K tile re-reads dY and every Cout tile re-reads X through L2 -- at ~1200
//    bytes per cycle chip-wide that, not the matrix pipe, bounds the narrow 96 x 128 tile on the wide layers of the video
//    encoder (tools/tune_wgrad_mvit.py: Cout or K >= 1536 are 7-20 % faster on 128 x 192).  Convolutions keep the pure MFMA
//    model: their taps re-use the input tile in L2 and were tuned on it (tools/tune_wgrad.py).
WgradPlan wgrad_plan(int Cout, long K, long seg_rows, int segments, bool one_tap = false) {
  WgradPlan pl{0, 1, 0};
  double best = 1e300;
  const long max_splits = (seg_rows + 127) / 128;
  int only = -1;
  if (tune(TUNE_WGRAD_CFG) >= 0) only = tune(TUNE_WGRAD_CFG) % kNumWgradCfgs;   // tuning aid
  for (int c = 0; c < kNumWgradCfgs; ++c) {
    if (only >= 0 && c != only) continue;
    const WgradCfg cf = kWgradCfgs[c];
    // the narrow-Cout tiles re-read the input once per 32/64 output channels (no memory term in the model below):
    // only for layers that are that narrow
    if (only < 0 && ((c == 3 && Cout > 48) || (c == 2 && Cout > 64))) continue;
    const long tiles = ((Cout + cf.bco - 1) / cf.bco) * ((K / 32 + cf.sl - 1) / cf.sl) * segments;
    const double step_cycles = 16.0 * cf.mfma_per_kstep * 64.0;
    // the LDS-DMA form of the 128 x 192 tile (plain products) holds three 40 KB stages: ONE workgroup per CU, 256 at once -- planned
    // in rounds of 256 (the rounds of 512 of the register-staged kernels gave it twice the splits it can run at once: two rounds of
    // half the length and twice the slab traffic)
    const bool dma = c == 0 && one_tap && tune(TUNE_WGRAD_DMA) != 0;
    const bool dma2 = dma && tune(TUNE_WGRAD_DMA) != 1;     // the two-stage form (default): two workgroups per CU; DIFFSAL_WGRAD_DMA=1: the three-stage form
    const long slots = (dma && !dma2) ? 256 : 512;
    for (int r = 1; r <= (dma ? 16 : 8); ++r) {
      long sp = r * slots / tiles;
      sp = sp < 1 ? 1 : (sp > max_splits ? max_splits : sp);
      sp = sp > 512 ? 512 : sp;
      const long wgs = tiles * sp;
      const long rounds = (wgs + slots - 1) / slots;
      const long rps = ((seg_rows + sp - 1) / sp + WG_BM - 1) / WG_BM * WG_BM;
      const double steps = static_cast<double>(rps) / 32.0 + 4.0;   // in 32-row steps
      const double occ = (dma && !dma2) ? 1.0 : (wgs >= 512 ? 2.0 : (wgs > 256 ? 2.0 * wgs / 512.0 : 1.0));
      double t = rounds * occ * steps * step_cycles;
      if (sp > 1) t += 2.0 * sp * segments * Cout * static_cast<double>(K) * 4.0 / 1250.0;
      if (one_tap) {
        const double tiles_k = static_cast<double>((K / 32 + cf.sl - 1) / cf.sl), tiles_co = static_cast<double>((Cout + cf.bco - 1) / cf.bco);
        const double t_mem = 4.0 * seg_rows * segments * (Cout * tiles_k + K * tiles_co) / 1200.0;
        t = 0.5 * (t > t_mem ? t : t_mem) + 0.5 * (t + t_mem);
      }
      if (t < best) { best = t; pl.cfg = c; pl.splits = static_cast<int>(sp); pl.rows_per_split = static_cast<int>(rps); }
    }
  }
  if (tune(TUNE_WGRAD_SPLITS) > 0) {   // tuning aid
    long sp = tune(TUNE_WGRAD_SPLITS);
    sp = sp < 1 ? 1 : (sp > max_splits ? max_splits : sp);
    pl.splits = static_cast<int>(sp);
    pl.rows_per_split = static_cast<int>(((seg_rows + sp - 1) / sp + WG_BM - 1) / WG_BM * WG_BM);
  }
  if (tune(TUNE_WGRAD_VERBOSE) == 1)
    fprintf(stderr, "wgrad plan: Cout=%d K=%ld seg_rows=%ld segs=%d -> cfg %d splits %d rows/split %d model %.0f cycles\n",
            Cout, K, seg_rows, segments, pl.cfg, pl.splits, pl.rows_per_split, best);
  return pl;
}

int wgrad_launch(WgradArgs a, int segments, float* out, hipStream_t s) {
  const WgradPlan pl = wgrad_plan(a.Cout, a.K, a.seg_rows, segments, a.taps == 1);
  a.splits = pl.splits;
  a.rows_per_split = pl.rows_per_split;
  const WgradCfg cf = kWgradCfgs[pl.cfg];
  const dim3 grid((a.Cout + cf.bco - 1) / cf.bco, (a.K / 32 + cf.sl - 1) / cf.sl, segments * pl.splits);
  const long n = static_cast<long>(a.Cout) * a.K;
  if (pl.splits == 1) a.slabs = out;   // one split per segment: the tile goes straight to its destination
  const bool plain = a.taps == 1 && a.stride_h == 1 && a.stride_w == 1 && a.pad_t == 0 && a.pad_l == 0 && a.Ho == a.H && a.Wo == a.W;
  if (plain && pl.cfg == 0 && tune(TUNE_WGRAD_DMA) != 0) {
    if (tune(TUNE_WGRAD_DMA) != 1) hipLaunchKernelGGL((wgrad_dma2_kernel<2, 2, 2, 3>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad_dma_kernel<2, 2, 2, 3>), grid, dim3(256), 0, s, a);
  } else
  switch (pl.cfg) {
    case 0: hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 3>), grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL((wgrad_kernel<3, 1, 4, 2>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((wgrad_kernel<2, 1, 4, 3>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((wgrad_kernel<1, 1, 4, 3>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((wgrad_kernel<3, 1, 4, 1>), grid, dim3(256), 0, s, a); break;
  }
  int rc = check_launch("conv_wgrad");
  if (rc || pl.splits == 1) return rc;
  const long g = (n / 4 + 15) / 16;
  const bool bias = a.dbias && a.dbias_out && segments == 1;
  const int extra = bias ? (a.Cout + 15) / 16 : 0;
  hipLaunchKernelGGL(slab_sum_kernel, dim3(static_cast<int>(g) + extra, segments), dim3(256), 0, s,
                     static_cast<const float*>(a.slabs), out, n, pl.splits, static_cast<int>(g), a.dbias, a.dbias_out, a.Cout);
  return check_launch("conv_wgrad(sum)");
}

}  // namespace

extern "C" size_t diffsal_conv_wgrad_ws_bytes(const diffsal_conv_desc* d) {
  if (!d || d->Cin <= 0 || d->Cin % 32) return 0;
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const long K = static_cast<long>(d->KH) * d->KW * d->Cin;
  const WgradPlan pl = wgrad_plan(d->Cout, K, M, 1, d->KH * d->KW == 1);
  return static_cast<size_t>(pl.splits) * d->Cout * K * sizeof(float);
}

extern "C" int diffsal_conv_wgrad_splits(const diffsal_conv_desc* d) {
  if (!d || d->Cin <= 0 || d->Cin % 32) return 0;
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const long K = static_cast<long>(d->KH) * d->KW * d->Cin;
  return wgrad_plan(d->Cout, K, M, 1, d->KH * d->KW == 1).splits;
}

extern "C" int diffsal_conv_wgrad(const diffsal_conv_desc* d, const float* in, const float* dy, float* dw_packed,
                                  double* dbias_part, float* dbias_out, void* ws, size_t ws_bytes, diffsal_stream_t stream) {
  DS_REQUIRE(d && in && dy && dw_packed && ws, DIFFSAL_E_ARG, "conv_wgrad: null argument");
  DS_REQUIRE(!dbias_out || dbias_part, DIFFSAL_E_ARG, "conv_wgrad: dbias_out needs dbias_part (the per-split partial sums)");
  DS_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0 && d->Cout % 4 == 0, DIFFSAL_E_SHAPE,
             "conv_wgrad: Cin=%d must be a multiple of 32 and Cout=%d of 4", d->Cin, d->Cout);
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const long K = static_cast<long>(d->KH) * d->KW * d->Cin;
  const long in_bytes = static_cast<long>(d->N) * d->H * d->W * d->Cin * 4;
  DS_REQUIRE(M > 0 && M < (1L << 31) && in_bytes < (1L << 32) - 16 && M * d->Cout * 4 < (1L << 32) - 16 && d->KH * d->KW <= 32,
             DIFFSAL_E_SHAPE, "conv_wgrad: problem too large");
  const size_t need = diffsal_conv_wgrad_ws_bytes(d);
  DS_REQUIRE(ws_bytes >= need && aligned16(ws) && aligned16(dy) && aligned16(in) && aligned16(dw_packed), DIFFSAL_E_ARG,
             "conv_wgrad: needs %zu bytes of 16-byte aligned workspace", need);
  WgradArgs a;
  a.in = in; a.dy = dy; a.slabs = static_cast<float*>(ws); a.dbias = dbias_part; a.dbias_out = dbias_out;
  a.M = static_cast<int>(M); a.K = static_cast<int>(K); a.Cout = d->Cout;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KW = d->KW; a.taps = d->KH * d->KW; a.stride_h = d->stride_h; a.stride_w = d->stride_w;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.dil_h = d->dil_h; a.dil_w = d->dil_w;
  a.seg_rows = static_cast<int>(M);
  a.in_bytes = static_cast<unsigned>(in_bytes);
  a.dy_bytes = static_cast<unsigned>(M * d->Cout * 4);
  return wgrad_launch(a, 1, dw_packed, static_cast<hipStream_t>(stream));
}

extern "C" size_t diffsal_wgrad_segmented_ws_bytes(int segments, int seg_rows, int K, int Cout) {
  if (segments <= 0 || seg_rows <= 0 || K <= 0 || K % 32 || Cout <= 0) return 0;
  const WgradPlan pl = wgrad_plan(Cout, K, seg_rows, segments, true);
  return static_cast<size_t>(segments) * pl.splits * Cout * K * sizeof(float);
}

extern "C" int diffsal_wgrad_segmented(const float* x, const float* dy, float* out, int segments, int seg_rows, int K,
                                       int Cout, void* ws, size_t ws_bytes, diffsal_stream_t stream) {
  DS_REQUIRE(x && dy && out && ws, DIFFSAL_E_ARG, "wgrad_segmented: null argument");
  DS_REQUIRE(segments > 0 && seg_rows > 0 && K > 0 && K % 32 == 0 && Cout > 0 && Cout % 4 == 0, DIFFSAL_E_SHAPE,
             "wgrad_segmented: bad shape segments=%d seg_rows=%d K=%d Cout=%d", segments, seg_rows, K, Cout);
  const long M = static_cast<long>(segments) * seg_rows;
  const long in_bytes = M * K * 4;
  DS_REQUIRE(M < (1L << 31) && in_bytes < (1L << 32) - 16 && M * Cout * 4 < (1L << 32) - 16, DIFFSAL_E_SHAPE,
             "wgrad_segmented: problem too large");
  const size_t need = diffsal_wgrad_segmented_ws_bytes(segments, seg_rows, K, Cout);
  DS_REQUIRE(ws_bytes >= need && aligned16(ws) && aligned16(dy) && aligned16(x) && aligned16(out), DIFFSAL_E_ARG,
             "wgrad_segmented: needs %zu bytes of 16-byte aligned workspace", need);
  WgradArgs a;
  a.in = x; a.dy = dy; a.slabs = static_cast<float*>(ws); a.dbias = nullptr; a.dbias_out = nullptr;
  a.M = static_cast<int>(M); a.K = K; a.Cout = Cout;
  a.H = 1; a.W = static_cast<int>(M); a.Cin = K; a.Ho = 1; a.Wo = static_cast<int>(M);
  a.KW = 1; a.taps = 1; a.stride_h = 1; a.stride_w = 1; a.pad_t = 0; a.pad_l = 0; a.dil_h = 1; a.dil_w = 1;
  a.seg_rows = seg_rows;
  a.in_bytes = static_cast<unsigned>(in_bytes);
  a.dy_bytes = static_cast<unsigned>(M * Cout * 4);
  return wgrad_launch(a, segments, out, static_cast<hipStream_t>(stream));
}

extern "C" int diffsal_colsum(const float* dy, float* out, int M, int C, int seg_rows, void* ws, size_t ws_bytes,
                              diffsal_stream_t stream) {
  DS_REQUIRE(dy && out && ws, DIFFSAL_E_ARG, "colsum: null argument");
  DS_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 4096 && seg_rows > 0 && M % seg_rows == 0, DIFFSAL_E_SHAPE,
             "colsum: bad shape M=%d C=%d seg_rows=%d", M, C, seg_rows);
  const int segs = M / seg_rows;
  int chunks = 1024 / segs;
  chunks = chunks < 1 ? 1 : (chunks > 512 ? 512 : chunks);
  while (chunks > 1 && seg_rows / chunks < 32) chunks >>= 1;
  DS_REQUIRE(ws_bytes >= static_cast<size_t>(segs) * chunks * C * sizeof(double), DIFFSAL_E_ARG,
             "colsum: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(colsum_kernel, dim3(chunks, segs), dim3(256), C * sizeof(double), s, dy, static_cast<double*>(ws), M,
                     C, seg_rows, chunks);
  int rc = check_launch("colsum");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 7) / 8, segs), dim3(256), 0, s,
                     static_cast<const double*>(ws), out, segs, chunks, C);
  return check_launch("colsum(sum)");
}
