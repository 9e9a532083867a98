// Streaming GEMM for the short-K token layers of the finest decoder stage:
//   out[m, n] = act(sum_k x[m, k] w[n, k] + bias[n]) + residual[m, n],   K in {96, 192}, N in {96, 192}, M huge.
//
// These layers (q / proj / fc1 / fc2 at C = 96: R/models/saliency_decoder/attention.py:78-83,
// common_block.py:125-147) are at the fp32-MFMA / HBM ridge (32 FLOP/B): the generic tiled kernel spends more
// time in per-tile prologues, barriers and store tails than in MFMAs.  Here the whole weight matrix lives in
// LDS for the lifetime of the workgroup, each WAVEFRONT owns 32-row tiles of x on its own and streams them
// through registers (double-buffered, one 96-wide K half ahead), so the steady state has NO barriers, no LDS
// traffic for x and statically counted vmcnt waits (loads are always older than the stores of the previous tile).
#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LinStreamArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* residual;
  float* out;
  int M, act;
};

constexpr int KHALF = 96;            // K is processed in halves of 96 (12 groups of 8)
constexpr int GROUPS = KHALF / 8;

template <int KH, int TN, bool HAS_RES>
__global__ __launch_bounds__(256, (TN <= 3 ? 2 : 1)) void lin_stream_kernel(LinStreamArgs p) {
  constexpr int K = KHALF * KH;
  constexpr int N = 32 * TN;
  constexpr int KP = K + 4;  // LDS pitch in floats: (K+4) mod 64 is 36 or 4 -> conflict-free ds_read_b128 per 16-lane group
  extern __shared__ __attribute__((aligned(16))) float Ws[];  // [N][KP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < N * (K / 4); i += 256) {
    const int n = i / (K / 4), k4 = i - n * (K / 4);
    st4(Ws + n * KP + k4 * 4, ld4(p.w + static_cast<long>(n) * K + k4 * 4));
  }
  __syncthreads();

  const int n_tiles = (p.M + 31) / 32;
  const int wave_id = blockIdx.x * 4 + wave;
  const int n_waves = gridDim.x * 4;
  const int lrow = lane & 31;
  const int lk = (lane >> 5) * 4;
  const float* wfrag = Ws + lrow * KP + lk;

  float4 xa[2][GROUPS];
  auto load_half = [&](int tile, int half, int set) {
    int m = tile * 32 + lrow;
    m = m < p.M ? m : p.M - 1;  // rows past M (and tiles past the end) read valid memory and are never stored
    const float* src = p.x + static_cast<long>(m) * K + half * KHALF + lk;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) xa[set][g] = ld4(src + g * 8);
  };

  f32x16 acc[TN];
  auto compute_half = [&](int half, int set) {
    float4 bf[2][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = ld4(wfrag + j * 32 * KP + half * KHALF);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
      if (g + 1 < GROUPS) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(g + 1) & 1][j] = ld4(wfrag + j * 32 * KP + half * KHALF + (g + 1) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep exactly one group of weight fragments in flight (register budget)
      const float4 a = xa[set][g];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = s == 0 ? a.x : s == 1 ? a.y : s == 2 ? a.z : a.w;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float4 b = bf[g & 1][j];
          const float bv = s == 0 ? b.x : s == 1 ? b.y : s == 2 ? b.z : b.w;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
  float bi[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bi[j] = p.bias ? p.bias[j * 32 + col_l] : 0.f;
  const float* __restrict__ resid = p.residual;
  float* __restrict__ outp = p.out;

  int tile = wave_id;
  if (tile < n_tiles) load_half(tile, 0, 0);
  for (; tile < n_tiles; tile += n_waves) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float res[HAS_RES ? TN : 1][16];
#pragma unroll
    for (int hf = 0; hf < KH; ++hf) {
      // next K half (of this tile, or the first half of this wave's next tile) starts its trip now
      if (hf + 1 < KH) load_half(tile, hf + 1, (hf + 1) & 1);
      else load_half(tile + n_waves, 0, (hf + 1) & 1);
      if (HAS_RES && hf == KH - 1) {  // residual of this tile: in flight during the last half's MFMAs
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int m = tile * 32 + (r & 3) + 8 * (r >> 2) + row_h;
            m = m < p.M ? m : p.M - 1;
            res[j][r] = resid[static_cast<long>(m) * N + j * 32 + col_l];
          }
      }
      compute_half(hf, hf & 1);
    }
    // epilogue: C/D map col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); 128-byte row segments
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = tile * 32 + (r & 3) + 8 * (r >> 2) + row_h;
        float v = acc[j][r] + bi[j];
        if (p.act == DIFFSAL_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == DIFFSAL_ACT_GELU_ERF) v = gelu_erf(v);
        else if (p.act == DIFFSAL_ACT_SIGMOID) v = sigmoidf_(v);
        if (HAS_RES) v += res[j][r];
        if (m < p.M) outp[static_cast<long>(m) * N + j * 32 + col_l] = v;
      }
    }
    static_assert(KH == 1 || KH == 2, "the register double buffer alternates per K half");
    if (KH == 1) {  // one half per tile: the buffer that was just filled becomes the current one
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) xa[0][g] = xa[1][g];
    }
  }
}

template <int KH, int TN, bool HAS_RES>
static int launch_stream(const LinStreamArgs& a, hipStream_t s) {
  constexpr int K = KHALF * KH, N = 32 * TN;
  const size_t lds = static_cast<size_t>(N) * (K + 4) * sizeof(float);
  static bool raised = false;
  if (lds > 64 * 1024 && !raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lin_stream_kernel<KH, TN, HAS_RES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = true;
  }
  const int per_cu = (TN <= 3 && lds <= 80 * 1024) ? 2 : 1;  // TN = 6 needs > 256 registers: one wave per SIMD
  const int n_tiles = (a.M + 31) / 32;
  int grid = 256 * per_cu;
  if (grid * 4 > n_tiles) grid = (n_tiles + 3) / 4;
  hipLaunchKernelGGL((lin_stream_kernel<KH, TN, HAS_RES>), dim3(grid), dim3(256), lds, s, a);
  return check_launch("linear_stream");
}

// Returns 1 if the shape is handled here (and launches), 0 if the caller should use the tiled kernel, <0 on error.
int try_linear_stream(const float* x, const float* w, const float* bias, const float* residual, float* out, long M,
                      int K, int N, int act, hipStream_t s) {
  if (M < 65536 || M >= (1L << 31) / 192) return 0;  // below ~256 tiles per CU-wave the tiled kernel wins
  if (!((K == 96 || K == 192) && (N == 96 || N == 192)) || (K == 192 && N == 192)) return 0;
  LinStreamArgs a{x, w, bias, residual, out, static_cast<int>(M), act};
  int rc;
  if (K == 96 && N == 96) rc = residual ? launch_stream<1, 3, true>(a, s) : launch_stream<1, 3, false>(a, s);
  else if (K == 96 && N == 192) rc = residual ? launch_stream<1, 6, true>(a, s) : launch_stream<1, 6, false>(a, s);
  else rc = residual ? launch_stream<2, 3, true>(a, s) : launch_stream<2, 3, false>(a, s);
  return rc == DIFFSAL_OK ? 1 : rc;
}

}  // namespace diffsal
