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
  if (lds > 64 * 1024) DS_RAISE_DYNAMIC_LDS((lin_stream_kernel<KH, TN, HAS_RES>), 160 * 1024);
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

// =================================================================================================================
// Fused MLP half of the finest decoder stage's TransformerBlock (+ the norm that feeds ReduceTemp):
//     x2 = x1 + fc2( gelu( fc1( LayerNorm_2(x1) ) ) )            R/models/saliency_decoder/transformer.py:153-157,
//     z  = LayerNorm_mts(x2)   (optional, frames < t_keep only)   common_block.py:125-147, sal_unet.py:447,473
// for C = 96, hidden = 192 (the only stage whose two weight matrices fit in LDS: 147 KB).  Replaces four launches
// (LayerNorm, fc1 + GELU, fc2 + residual, LayerNorm) and removes three round trips of the 74 MB token tensor and one of
// the 149 MB hidden tensor per step (B = 4): the hidden activations never leave registers.
//
// One wavefront owns 32 tokens at a time and works TRANSPOSED (hidden^T = W1 xn^T, out^T = W2 hidden^T) so that the
// C/D layout of one product (column = lane & 31 = token, 16 rows per lane) is directly the B operand of the next one
// (v_mfma_f32_32x32x2_f32 takes its two k values from the two lane halves; the k order of a contraction is free as long
// as A and B agree, so half h pairs "its" rows {4h + (r & 3) + 8 (r >> 2)} with the same columns of the weight).  Both
// weights stay in LDS for the lifetime of the workgroup; steady state has no barriers.  Exact fp32.
// =================================================================================================================
struct MlpBlockArgs {
  const float* x1;
  const float* g2; const float* be2;      // norm2
  const float* w1; const float* b1;       // fc1 [192][96]
  const float* w2; const float* b2;       // fc2 [96][192]
  const float* gz; const float* bez;      // norm_mts (z output), may be null with z
  float* x2;
  float* z;
  int M;
  float eps2, epsz;
  int hw, T, t_keep;                      // z is written for tokens whose frame (m / hw) % T < t_keep
};

// Eight wavefronts per workgroup (two per SIMD) share the resident weights: while one wave of a SIMD is in its LayerNorm / GELU /
// store phase (VALU: ~2000 instructions of erf-GELU per tile) the other issues MFMAs.  With four waves (one per SIMD, 346
// registers each, the next tile prefetched into registers) the matrix pipe sat idle through every VALU phase: 0.55 busy.
// The second resident wave also covers the tile loads, so the register prefetch is gone and a wave fits in 256 registers.
__global__ __launch_bounds__(512) void mlp_block_kernel(MlpBlockArgs p) {
  constexpr int C = 96, HID = 192;
  constexpr int P1 = C + 4, P2 = HID + 4;           // LDS pitches (floats): conflict-free ds_read_b128 per 16-lane group
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* W1s = sm;                                   // [HID][P1]
  float* W2s = W1s + HID * P1;                       // [C][P2]
  float* vec = W2s + C * P2;                         // g2 | be2 | b2 | gz | bez (C each), b1 (HID)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < HID * (C / 4); i += 512) {
    const int n = i / (C / 4), c4 = (i - n * (C / 4)) * 4;
    st4(W1s + n * P1 + c4, ld4(p.w1 + n * C + c4));
  }
  for (int i = tid; i < C * (HID / 4); i += 512) {
    const int c = i / (HID / 4), n4 = (i - c * (HID / 4)) * 4;
    st4(W2s + c * P2 + n4, ld4(p.w2 + c * HID + n4));
  }
  for (int i = tid; i < C; i += 512) {
    vec[i] = p.g2[i]; vec[C + i] = p.be2[i]; vec[2 * C + i] = p.b2[i];
    vec[3 * C + i] = p.z ? p.gz[i] : 0.f; vec[4 * C + i] = p.z ? p.bez[i] : 0.f;
  }
  for (int i = tid; i < HID; i += 512) vec[5 * C + i] = p.b1[i];
  __syncthreads();

  const int ml = lane & 31, hf = lane >> 5;
  const int n_tiles = (p.M + 31) / 32;
  const int n_waves = gridDim.x * 8;
  const float* w1frag = W1s + ml * P1 + 4 * hf;      // + 32 t * P1 + 8 i : W1[n = 32t + ml][c = 8i + 4hf ..]
  const float* w2frag = W2s + ml * P2 + 4 * hf;      // + 32 u * P2 + 32 t + 8 g : W2[c = 32u + ml][n = 32t + 8g + 4hf ..]

  float4 xa[12];
  auto load_tile = [&](int tile, float4 (&dst)[12]) {
    int m = tile * 32 + ml;
    m = m < p.M ? m : p.M - 1;
    const float* src = p.x1 + static_cast<long>(m) * C + 4 * hf;
#pragma unroll
    for (int i = 0; i < 12; ++i) dst[i] = ld4(src + 8 * i);          // channels 8i + 4hf .. + 3
  };
  auto ln_rows = [&](const float4 (&v)[12], float& mean, float& rstd, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    s += lane_xor32(s);
    mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
    q += lane_xor32(q);
    rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
  };

  for (int tile = blockIdx.x * 8 + wave; tile < n_tiles; tile += n_waves) {
    load_tile(tile, xa);
    // ---- LayerNorm_2 in registers
    float mean, rstd;
    ln_rows(xa, mean, rstd, p.eps2);
    // ---- hidden^T = W1 xn^T + b1, GELU:  6 tiles of 32 hidden units, column = this lane's token
    f32x16 hid[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = vec[5 * C + 32 * t + 4 * hf + (r & 3) + 8 * (r >> 2)];
    {
      // the normalised pieces (B operands); weight fragments are read one group ahead of their MFMAs (mfma_groups_f32)
      float4 xnp[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float4 g = ld4(vec + 8 * i + 4 * hf), b = ld4(vec + C + 8 * i + 4 * hf);
        xnp[i] = make_float4((xa[i].x - mean) * rstd * g.x + b.x, (xa[i].y - mean) * rstd * g.y + b.y,
                             (xa[i].z - mean) * rstd * g.z + b.z, (xa[i].w - mean) * rstd * g.w + b.w);
      }
      mfma_groups_f32<72>([&](int k) { return w1frag + 32 * (k % 6) * P1 + 8 * (k / 6); },          // k = 6 i + t
                          [&](int k, float4 a) {
                            const int i = k / 6, t = k % 6;
                            DS_MFMA4(hid[t], a, xnp[i].x, xnp[i].y, xnp[i].z, xnp[i].w);
                          });
    }
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = gelu_erf(hid[t][r]);
    // x1 again for the residual (L2-resident: this wave read it a moment ago): its registers were given up during the first
    // product so that two waves fit on a SIMD; the reload lands during the second product
    load_tile(tile, xa);
    // ---- out^T = W2 hidden^T: 3 tiles of 32 channels; k pairs = (half 0's row, half 1's row) of each hidden register
    f32x16 o[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][r] = 0.f;
    mfma_groups_f32<72>([&](int k) { return w2frag + 32 * (k % 3) * P2 + 8 * (k / 3); },              // k = 3 (4 t + g) + u
                        [&](int k, float4 a) {
                          const int u = k % 3, tg = k / 3, t = tg >> 2, g = tg & 3;
                          DS_MFMA4(o[u], a, hid[t][4 * g + 0], hid[t][4 * g + 1], hid[t][4 * g + 2], hid[t][4 * g + 3]);
                        });
    // ---- x2 = out + b2 + x1: this lane holds channels c = 32u + 8g + 4hf .. + 3 of its token = piece i = 4u + g of xa
    const int m = tile * 32 + ml;
    float4 (&y)[12] = xa;                       // x2 overwrites x1 in place (x1 is not needed afterwards)
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = 4 * u + g;
        const float4 b = ld4(vec + 2 * C + 8 * i + 4 * hf);
        y[i].x = o[u][4 * g + 0] + b.x + xa[i].x; y[i].y = o[u][4 * g + 1] + b.y + xa[i].y;
        y[i].z = o[u][4 * g + 2] + b.z + xa[i].z; y[i].w = o[u][4 * g + 3] + b.w + xa[i].w;
      }
    if (m < p.M) {
      float* dst = p.x2 + static_cast<long>(m) * C + 4 * hf;
#pragma unroll
      for (int i = 0; i < 12; ++i) st4(dst + 8 * i, y[i]);
    }
    if (p.z) {
      float mz, rz;
      ln_rows(y, mz, rz, p.epsz);
      const bool keep = m < p.M && ((m / p.hw) % p.T) < p.t_keep;
      if (keep) {
        float* dz = p.z + static_cast<long>(m) * C + 4 * hf;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const float4 g = ld4(vec + 3 * C + 8 * i + 4 * hf), b = ld4(vec + 4 * C + 8 * i + 4 * hf);
          st4(dz + 8 * i, make_float4((y[i].x - mz) * rz * g.x + b.x, (y[i].y - mz) * rz * g.y + b.y,
                                      (y[i].z - mz) * rz * g.z + b.z, (y[i].w - mz) * rz * g.w + b.w));
        }
      }
    }
  }
}

}  // namespace diffsal

extern "C" int diffsal_mlp_block(const float* x1, const float* g2, const float* be2, float eps2, const float* w1,
                                 const float* b1, const float* w2, const float* b2, float* x2, float* z, const float* gz,
                                 const float* bez, float epsz, long M, int C, int hidden, int hw, int T, int t_keep,
                                 diffsal_stream_t stream) {
  using namespace diffsal;
  DS_REQUIRE(x1 && g2 && be2 && w1 && b1 && w2 && b2 && x2, DIFFSAL_E_ARG, "mlp_block: null argument");
  DS_REQUIRE(C == 96 && hidden == 192, DIFFSAL_E_SHAPE, "mlp_block: built for C = 96, hidden = 192 (got %d, %d)", C, hidden);
  DS_REQUIRE(M > 0 && M < (1L << 31) / 96, DIFFSAL_E_SHAPE, "mlp_block: M = %ld", M);
  DS_REQUIRE(!z || (gz && bez && hw > 0 && T > 0 && t_keep > 0), DIFFSAL_E_ARG, "mlp_block: z needs its norm and the frame geometry");
  DS_REQUIRE(aligned16(x1) && aligned16(x2) && aligned16(w1) && aligned16(w2) && (!z || aligned16(z)), DIFFSAL_E_ALIGN,
             "mlp_block: misaligned pointer");
  DS_REQUIRE(x1 != x2, DIFFSAL_E_ARG, "mlp_block: in-place operation is not supported (tiles are prefetched)");
  MlpBlockArgs a{x1, g2, be2, w1, b1, w2, b2, gz, bez, x2, z, static_cast<int>(M), eps2, epsz, hw > 0 ? hw : 1, T > 0 ? T : 1, t_keep};
  const size_t lds = (static_cast<size_t>(192) * 100 + 96 * 196 + 5 * 96 + 192) * sizeof(float);
  DS_RAISE_DYNAMIC_LDS((mlp_block_kernel), 160 * 1024);
  const int n_tiles = static_cast<int>((M + 31) / 32);
  int grid = 256;
  if (grid * 8 > n_tiles) grid = (n_tiles + 7) / 8;
  hipLaunchKernelGGL(mlp_block_kernel, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), a);
  return check_launch("mlp_block");
}
