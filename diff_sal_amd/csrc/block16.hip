// Fused second half of the finest stage's TransformerBlock on bf16 / fp16 storage (C = 96, hidden = 192):
//
//     x1 = x + proj(o)                                   R/models/saliency_decoder/attention.py:110, transformer.py:151-152
//     x2 = x1 + fc2( gelu( fc1( LayerNorm_2(x1) ) ) )    transformer.py:153-157, common_block.py:125-147
//     z  = LayerNorm_mts(x2)  (frames < t_keep only)     sal_unet.py:447,473
//
// In 16-bit storage all three weight matrices fit in LDS together (99 KB), so the five launches of the unfused path
// (proj GEMM, LayerNorm, fc1 GEMM, fc2 GEMM, LayerNorm) and their seven round trips of token-sized tensors collapse into one
// streaming kernel: read o and x once, write x2 and z once; x1, the normalised activations and the 192-wide hidden layer
// never leave registers.
//
// A wavefront owns 32 tokens and works TRANSPOSED, like the fp32 mlp_block kernel: every product is  Y^T = W X^T  with the
// weight as the A operand (rows = output features, read from LDS) and the activations as the B operand (column = this
// lane's token).  The C/D layout of one v_mfma_f32_32x32x16 (column = lane & 31, rows 4h + (r & 3) + 8 (r >> 2), h = lane
// half) is turned into the B operand of the next one without any data movement: for k-step (u, j) lane half h supplies
// the eight accumulator registers r = 8j .. 8j + 7 of row tile u, which are features 32u + 16j + 4h + {0..3, 8..11}; the
// weight's columns are stored in LDS in exactly that permuted order so that the A operand is one ds_read_b128.
// fp32 accumulation, fp32 LayerNorm statistics, fp32 bias / residual adds; one rounding per stored value.
#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Blk16;
template <> struct Blk16<__bf16> {
  typedef bf16x8 vec;
  static __device__ __forceinline__ f32x16 mma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Blk16<_Float16> {
  typedef f16x8 vec;
  static __device__ __forceinline__ f32x16 mma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T>
struct Block16Args {
  const T* o; const T* x;
  const T* wp; const float* bp;
  const float* g2; const float* be2;
  const T* w1; const float* b1;
  const T* w2; const float* b2;
  const float* gz; const float* bez;
  T* x2; T* z;
  int M;
  float eps2, epsz;
  int hw, Tn, t_keep;
};

// four storage elements held in two dwords -> fp32
template <typename T> __device__ __forceinline__ float4 unpack4(uint2 u);
template <> __device__ __forceinline__ float4 unpack4<__bf16>(uint2 u) {
  return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xFFFF0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xFFFF0000u));
}
template <> __device__ __forceinline__ float4 unpack4<_Float16>(uint2 u) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 h = __builtin_bit_cast(h4, u);
  return make_float4(static_cast<float>(h.x), static_cast<float>(h.y), static_cast<float>(h.z), static_cast<float>(h.w));
}

// column permutation inside a 32-feature block: LDS position q = 16 j + 8 h + e  <-  feature 16 j + 4 h + (e < 4 ? e : e + 4)
__device__ __forceinline__ int perm32(int q) {
  const int j = q >> 4, h = (q >> 3) & 1, e = q & 7;
  return 16 * j + 4 * h + (e < 4 ? e : e + 4);
}

// NW wavefronts per workgroup (one workgroup per CU: the weights are staged once).  NW = 8 -- two wavefronts per SIMD, 256 registers
// each, no next-tile prefetch registers: the chain of a tile is MFMA -> LayerNorm / GELU on the VALU -> MFMA, and a single wavefront per
// SIMD leaves the matrix pipe idle through the ~100 erf-GELUs per lane (and the VALU idle through the MFMAs and the loads); the second
// wavefront fills both.
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64, 1) void block16_kernel(Block16Args<T> p) {
  constexpr int NT = NW * 64;
  constexpr bool PREFETCH = NW == 4;
  typedef typename Blk16<T>::vec vec;
  constexpr int C = 96, HID = 192;
  constexpr int P1 = C + 8, P2 = HID + 8;          // LDS row pitches in elements: 208 B and 400 B = odd multiples of 16 B
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T* Wps = reinterpret_cast<T*>(smraw);            // [C][P1]   natural column order (its B operand comes from memory)
  T* W1s = Wps + C * P1;                           // [HID][P1] columns permuted per 32-block
  T* W2s = W1s + HID * P1;                         // [C][P2]   columns permuted per 32-block
  float* vecs = reinterpret_cast<float*>(W2s + C * P2);   // bp | g2 | be2 | b2 | gz | bez (C each) | b1 (HID)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < C * C; i += NT) { const int r = i / C, c = i - r * C; Wps[r * P1 + c] = p.wp[i]; }
  for (int i = tid; i < HID * C; i += NT) {
    const int r = i / C, q = i - r * C;
    W1s[r * P1 + q] = p.w1[r * C + (q & ~31) + perm32(q & 31)];
  }
  for (int i = tid; i < C * HID; i += NT) {
    const int r = i / HID, q = i - r * HID;
    W2s[r * P2 + q] = p.w2[r * HID + (q & ~31) + perm32(q & 31)];
  }
  for (int i = tid; i < C; i += NT) {
    vecs[i] = p.bp[i]; vecs[C + i] = p.g2[i]; vecs[2 * C + i] = p.be2[i]; vecs[3 * C + i] = p.b2[i];
    vecs[4 * C + i] = p.z ? p.gz[i] : 0.f; vecs[5 * C + i] = p.z ? p.bez[i] : 0.f;
  }
  for (int i = tid; i < HID; i += NT) vecs[6 * C + i] = p.b1[i];
  __syncthreads();

  const int ml = lane & 31, hf = lane >> 5;
  const int n_tiles = (p.M + 31) / 32;
  const int n_waves = gridDim.x * NW;
  const T* wpf = Wps + ml * P1 + 8 * hf;           // + 32u*P1 + 16 s      : Wp[c = 32u + ml][k = 16s + 8hf ..]
  const T* w1f = W1s + ml * P1 + 8 * hf;           // + 32t*P1 + 32u + 16j : W1[n = 32t + ml][perm block u, step j]
  const T* w2f = W2s + ml * P2 + 8 * hf;           // + 32u*P2 + 32t + 16j : W2[c = 32u + ml][perm block t, step j]

  uint4 oa[6], onext[6];        // raw 16-byte pieces of o:   elements 16 s + 8 hf .. + 7
  uint2 xa[12], xnext[12];      // raw 8-byte pieces of x:    channels 8 i + 4 hf .. + 3  (i = 4u + g)
  auto load_tile = [&](int tile, uint4 (&od)[6], uint2 (&xd)[12]) {
    int m = tile * 32 + ml;
    m = m < p.M ? m : p.M - 1;
    const T* os = p.o + static_cast<long>(m) * C + 8 * hf;
    const T* xs = p.x + static_cast<long>(m) * C + 4 * hf;
#pragma unroll
    for (int s = 0; s < 6; ++s) od[s] = *reinterpret_cast<const uint4*>(os + 16 * s);
#pragma unroll
    for (int i = 0; i < 12; ++i) xd[i] = *reinterpret_cast<const uint2*>(xs + 8 * i);
  };
  // B operand of k-step (tile u, half j) from the fp32 C/D registers of a previous product
  auto b_from = [&](const f32x16& a, int j) -> vec {
    f32x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = a[8 * j + e];
    return __builtin_convertvector(t, vec);
  };
  auto ln_stats = [&](const f32x16 (&v)[3], float& mean, float& rstd, float eps) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += v[u][r];
    s += lane_xor32(s);
    mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = v[u][r] - mean; q = fmaf(d, d, q); }
    q += lane_xor32(q);
    rstd = 1.0f / sqrtf(fmaf(q, 1.0f / C, eps));       // (explicit FMAs here and below: the two instantiations of this kernel round alike)
  };
  // channel of accumulator register r of row tile u for this lane
  auto chan = [&](int u, int r) { return 32 * u + 4 * hf + (r & 3) + 8 * (r >> 2); };

  int tile = blockIdx.x * NW + wave;
  if (PREFETCH && tile < n_tiles) load_tile(tile, oa, xa);
  for (; tile < n_tiles; tile += n_waves) {
    if constexpr (PREFETCH) {
      if (tile + n_waves < n_tiles) load_tile(tile + n_waves, onext, xnext);
    } else {
      load_tile(tile, oa, xa);
    }
    // ---- x1^T = Wp o^T + bp + x^T
    f32x16 x1[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) x1[u][r] = vecs[chan(u, r)];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const vec b = __builtin_bit_cast(vec, oa[s]);
#pragma unroll
      for (int u = 0; u < 3; ++u)
        x1[u] = Blk16<T>::mma(*reinterpret_cast<const vec*>(wpf + 32 * u * P1 + 16 * s), b, x1[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 r4 = unpack4<T>(xa[4 * u + g]);                           // 4 consecutive channels 32u + 8g + 4hf
        x1[u][4 * g + 0] += r4.x; x1[u][4 * g + 1] += r4.y; x1[u][4 * g + 2] += r4.z; x1[u][4 * g + 3] += r4.w;
      }
    // ---- LayerNorm_2 (fp32 statistics over the 96 channels of this lane's token: 48 here + 48 in the partner lane)
    float mean, rstd;
    ln_stats(x1, mean, rstd, p.eps2);
    // the normalised activations go straight into the packed B operands of fc1 (24 registers instead of 48 fp32 ones)
    vec xb[3][2];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      f32x16 xn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = chan(u, r);
        xn[r] = fmaf((x1[u][r] - mean) * rstd, vecs[C + c], vecs[2 * C + c]);
      }
      xb[u][0] = b_from(xn, 0);
      xb[u][1] = b_from(xn, 1);
    }
    // ---- hidden^T = gelu(W1 xn^T + b1): 6 row tiles, 6 k-steps (3 channel blocks x 2)
    f32x16 hid[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = vecs[6 * C + chan(t, r)];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const vec b = xb[u][j];
#pragma unroll
        for (int t = 0; t < 6; ++t)
          hid[t] = Blk16<T>::mma(*reinterpret_cast<const vec*>(w1f + 32 * t * P1 + 32 * u + 16 * j), b, hid[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = gelu_erf(hid[t][r]);
    // ---- x2^T = W2 hidden^T + b2 + x1^T: 3 row tiles, 12 k-steps
    f32x16 y[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[u][r] = vecs[3 * C + chan(u, r)] + x1[u][r];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const vec b = b_from(hid[t], j);
#pragma unroll
        for (int u = 0; u < 3; ++u)
          y[u] = Blk16<T>::mma(*reinterpret_cast<const vec*>(w2f + 32 * u * P2 + 32 * t + 16 * j), b, y[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    // ---- stores: 4 consecutive channels (8 bytes) per (u, g)
    const int m = tile * 32 + ml;
    if (m < p.M) {
      T* dst = p.x2 + static_cast<long>(m) * C + 4 * hf;
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          st4(dst + 32 * u + 8 * g, make_float4(y[u][4 * g + 0], y[u][4 * g + 1], y[u][4 * g + 2], y[u][4 * g + 3]));
    }
    if (p.z) {
      float mz, rz;
      ln_stats(y, mz, rz, p.epsz);
      if (m < p.M && ((m / p.hw) % p.Tn) < p.t_keep) {
        T* dz = p.z + static_cast<long>(m) * C + 4 * hf;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = 32 * u + 8 * g + 4 * hf + e;
              v4[e] = fmaf((y[u][4 * g + e] - mz) * rz, vecs[4 * C + c], vecs[5 * C + c]);
            }
            st4(dz + 32 * u + 8 * g, make_float4(v4[0], v4[1], v4[2], v4[3]));
          }
      }
    }
    if constexpr (PREFETCH) {
#pragma unroll
      for (int s = 0; s < 6; ++s) oa[s] = onext[s];
#pragma unroll
      for (int i = 0; i < 12; ++i) xa[i] = xnext[i];
    }
  }
}

template <typename T>
static int launch_block16(const Block16Args<T>& a, hipStream_t s) {
  const size_t lds = (static_cast<size_t>(96) * 104 + 192 * 104 + 96 * 200) * sizeof(T) + (6 * 96 + 192) * sizeof(float);
  const int n_tiles = (a.M + 31) / 32;
  // eight wavefronts per workgroup from two tiles per wavefront on (DIFFSAL_BLOCK16_WAVES = 4 / 8 forces)
  const int forced = tune(TUNE_BLOCK16_WAVES);
  const bool eight = forced == 8 || (forced != 4 && n_tiles >= 256 * 8 * 2);
  int grid = 256;
  if (eight) {
    DS_RAISE_DYNAMIC_LDS((block16_kernel<T, 8>), 160 * 1024);
    if (grid * 8 > n_tiles) grid = (n_tiles + 7) / 8;
    hipLaunchKernelGGL((block16_kernel<T, 8>), dim3(grid), dim3(512), lds, s, a);
  } else {
    DS_RAISE_DYNAMIC_LDS((block16_kernel<T, 4>), 160 * 1024);
    if (grid * 4 > n_tiles) grid = (n_tiles + 3) / 4;
    hipLaunchKernelGGL((block16_kernel<T, 4>), dim3(grid), dim3(256), lds, s, a);
  }
  return check_launch("block16");
}

}  // namespace diffsal

extern "C" int diffsal_block16(const void* o, const void* x, const void* wp, const float* bp, const float* g2, const float* be2,
                               float eps2, const void* w1, const float* b1, const void* w2, const float* b2, void* x2, void* z,
                               const float* gz, const float* bez, float epsz, long M, int C, int hidden, int hw, int T,
                               int t_keep, int dtype, diffsal_stream_t stream) {
  using namespace diffsal;
  DS_REQUIRE(o && x && wp && bp && g2 && be2 && w1 && b1 && w2 && b2 && x2, DIFFSAL_E_ARG, "block16: null argument");
  DS_REQUIRE(C == 96 && hidden == 192, DIFFSAL_E_SHAPE, "block16: built for C = 96, hidden = 192 (got %d, %d)", C, hidden);
  DS_REQUIRE(dtype == DIFFSAL_BF16 || dtype == DIFFSAL_F16, DIFFSAL_E_ARG, "block16: 16-bit storage only (dtype %d)", dtype);
  DS_REQUIRE(M > 0 && M < (1L << 31) / 96, DIFFSAL_E_SHAPE, "block16: M = %ld", M);
  DS_REQUIRE(!z || (gz && bez && hw > 0 && T > 0 && t_keep > 0), DIFFSAL_E_ARG, "block16: z needs its norm and the frame geometry");
  DS_REQUIRE(aligned16(o) && aligned16(x) && aligned16(x2) && (!z || aligned16(z)), DIFFSAL_E_ALIGN, "block16: misaligned pointer");
  DS_REQUIRE(x2 != x && x2 != o, DIFFSAL_E_ARG, "block16: in-place operation is not supported (tiles are prefetched)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == DIFFSAL_BF16) {
    Block16Args<__bf16> a{static_cast<const __bf16*>(o), static_cast<const __bf16*>(x), static_cast<const __bf16*>(wp), bp, g2, be2,
                          static_cast<const __bf16*>(w1), b1, static_cast<const __bf16*>(w2), b2, gz, bez,
                          static_cast<__bf16*>(x2), static_cast<__bf16*>(z), static_cast<int>(M), eps2, epsz, hw > 0 ? hw : 1,
                          T > 0 ? T : 1, t_keep};
    return launch_block16(a, s);
  }
  Block16Args<_Float16> a{static_cast<const _Float16*>(o), static_cast<const _Float16*>(x), static_cast<const _Float16*>(wp), bp, g2,
                          be2, static_cast<const _Float16*>(w1), b1, static_cast<const _Float16*>(w2), b2, gz, bez,
                          static_cast<_Float16*>(x2), static_cast<_Float16*>(z), static_cast<int>(M), eps2, epsz, hw > 0 ? hw : 1,
                          T > 0 ? T : 1, t_keep};
  return launch_block16(a, s);
}
