// K11 on bf16 / fp16 storage for the wide stages (head dim 192 / 384 = stages 1 / 0; R/models/saliency_decoder/attention.py:97-108, scale
// C^-1/2 of the FULL width, quirk Q6) on the matrix cores.
//
// attention16_kernel (misc.hip) forms the 18 scores and the 18-term output of a query on the VALU from K / V rows in LDS: 2 x 18 x d
// multiply-adds per query with one 16-byte LDS read per four of them -- at d = 384 the launch is bound by LDS reads (55 KB per query and
// head), 1.7 TB/s of q / o at 64 clips.  Here, as in block_front's phase D, a wavefront owns 32 queries and works TRANSPOSED:
//     S^T = K Q^T      A = K rows (keys, padded to 32) from LDS, B = this lane's query, 16 bytes per k-step straight from memory
//     P^T = softmax over the rows (keys) of S^T: down the 16 registers of a lane + one lane ^ 32 exchange
//     O^T = V^T P^T    A = V^T rows (channels) from LDS, key columns stored in the order the C/D registers of S^T come in, so
//                      that P^T goes from accumulator registers to B operand without data movement
// A workgroup (4 wavefronts) = one (image, head) and 128 queries; K and V^T of the head in LDS (56 KB at d = 384: two workgroups per
// CU).  fp32 accumulation and softmax; the probabilities enter O^T as a 16-bit hi + lo pair (no rounding to speak of), one rounding on
// the output.
#include "common.h"

namespace diffsal {

typedef float am_f32x16 __attribute__((ext_vector_type(16)));
typedef float am_f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 am_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 am_f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct AmMma;
template <> struct AmMma<__bf16> {
  typedef am_bf16x8 vec;
  static __device__ __forceinline__ am_f32x16 run(vec a, vec b, am_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct AmMma<_Float16> {
  typedef am_f16x8 vec;
  static __device__ __forceinline__ am_f32x16 run(vec a, vec b, am_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// key -> its column in V^T's LDS rows: position 16 j + 8 h + e holds key 16 j + 4 h + (e < 4 ? e : e + 4), the order in which the keys
// sit in the C/D registers of S^T (register r of lane half h = key 4 h + (r & 3) + 8 (r >> 2))
__device__ __forceinline__ int am_key_col(int key) {
  const int j = key >> 4, t = key & 15;
  const int h = (t >> 2) & 1, e = (t & 3) + ((t >> 3) << 2);
  return 16 * j + 8 * h + e;
}

template <typename T, int D>
__global__ __launch_bounds__(256, 2) void attention16_mfma_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                  T* __restrict__ o, int Lq, int Lk, int C, int heads, float scale) {
  typedef typename AmMma<T>::vec vec;
  constexpr int PW = D + 8, PV = 40;             // row pitches in elements: (D + 8) * 2 and 80 bytes, odd multiples of 16 bytes
  constexpr int KS = D / 16, NU = D / 32;        // k-steps of S^T, row tiles of O^T
  extern __shared__ __attribute__((aligned(16))) unsigned char am_smem[];
  T* Ks = reinterpret_cast<T*>(am_smem);         // [32][PW]  keys (rows past Lk: zero)
  T* Vts = Ks + 32 * PW;                         // [D][PV]   V^T, key columns in register order (columns of keys past Lk: zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ml = lane & 31, hf = lane >> 5;
  const int n = blockIdx.x / heads, hd = blockIdx.x - n * heads;
  const int cb = hd * D;
  // ---- this lane's query (token ml of the wavefront's 32), requested before the K / V staging: its latency runs under it
  const int tok = (blockIdx.y * 4 + wave) * 32 + ml;
  const bool live = tok < Lq;
  const T* qrow = q + (static_cast<long>(n) * Lq + (live ? tok : Lq - 1)) * C + cb + 8 * hf;
  vec qb[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qb[s] = *reinterpret_cast<const vec*>(qrow + 16 * s);
  // ---- K rows and V^T columns of this head
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (int it = tid; it < 32 * (D / 8); it += 256) {
    const int j = it / (D / 8), oc = it - j * (D / 8);
    const uint4 kr = j < Lk ? *reinterpret_cast<const uint4*>(k + (static_cast<long>(n) * Lk + j) * C + cb + oc * 8) : zero;
    *reinterpret_cast<uint4*>(Ks + j * PW + oc * 8) = kr;
    const uint4 vr = j < Lk ? *reinterpret_cast<const uint4*>(v + (static_cast<long>(n) * Lk + j) * C + cb + oc * 8) : zero;
    const T* ve = reinterpret_cast<const T*>(&vr);
    const int col = am_key_col(j);
#pragma unroll
    for (int e = 0; e < 8; ++e) Vts[(oc * 8 + e) * PV + col] = ve[e];
  }
  __syncthreads();
  if ((blockIdx.y * 4 + wave) * 32 >= Lq) return;           // a wavefront past the last query (uniform)
  // ---- S^T = K Q^T: rows = keys, this lane's column = its query
  am_f32x16 st;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = 0.f;
  const T* kf = Ks + ml * PW + 8 * hf;
#pragma unroll
  for (int s = 0; s < KS; ++s) st = AmMma<T>::run(*reinterpret_cast<const vec*>(kf + 16 * s), qb[s], st);
  // ---- softmax over the keys: register r of lane half hf = key 4 hf + (r & 3) + 8 (r >> 2)
  float mx = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool ok = (r & 3) + 8 * (r >> 2) + 4 * hf < Lk;
    mx = ok ? fmaxf(mx, st[r]) : mx;
  }
  mx = fmaxf(mx, lane_xor32(mx));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool ok = (r & 3) + 8 * (r >> 2) + 4 * hf < Lk;
    const float e = ok ? __expf((st[r] - mx) * scale) : 0.f;
    st[r] = e;
    sum += e;
  }
  sum += lane_xor32(sum);
  const float inv = 1.0f / sum;
  // the probabilities as TWO 16-bit B operands, hi + lo (lo = the rounding error of hi): a single rounding of P doubled the error of the
  // operator on bf16 storage (4-5e-3 of the maximum against 2-3e-3 with fp32 probabilities); the second pair of MFMAs is free here
  vec pb[2], pl[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    am_f32x8 t8, r8;
#pragma unroll
    for (int e = 0; e < 8; ++e) t8[e] = st[8 * j + e] * inv;
    pb[j] = __builtin_convertvector(t8, vec);
    const am_f32x8 back = __builtin_convertvector(pb[j], am_f32x8);
#pragma unroll
    for (int e = 0; e < 8; ++e) r8[e] = t8[e] - back[e];
    pl[j] = __builtin_convertvector(r8, vec);
  }
  // ---- O^T = V^T P^T, a 32-channel row tile at a time; register r = channel 32 u + 4 hf + (r & 3) + 8 (r >> 2): four 8-byte stores
  const T* vf = Vts + ml * PV + 8 * hf;
  T* orow = o + (static_cast<long>(n) * Lq + tok) * C + cb + 4 * hf;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    am_f32x16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
    const vec v0 = *reinterpret_cast<const vec*>(vf + 32 * u * PV);
    ot = AmMma<T>::run(v0, pl[0], ot);
    ot = AmMma<T>::run(v0, pb[0], ot);
    if (Lk > 16) {
      const vec v1 = *reinterpret_cast<const vec*>(vf + 32 * u * PV + 16);
      ot = AmMma<T>::run(v1, pl[1], ot);
      ot = AmMma<T>::run(v1, pb[1], ot);
    }
    if (live) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st4(orow + 32 * u + 8 * g, make_float4(ot[4 * g + 0], ot[4 * g + 1], ot[4 * g + 2], ot[4 * g + 3]));
    }
  }
}

// 1 if launched, 0 if the shape is not this kernel's (the caller goes on), < 0 on error
int try_attention16_mfma(const void* q, const void* k, const void* v, void* o, int N, int Lq, int Lk, int C, int heads, float scale,
                         int dtype, hipStream_t s) {
  if (dtype == DIFFSAL_F32 || tune(TUNE_NO_STREAM16) == 1 || tune(TUNE_NO_ATTN16_MFMA) == 1) return 0;
  const int d = heads > 0 ? C / heads : 0;
  if ((d != 192 && d != 384) || d * heads != C || Lk < 1 || Lk > 32 || Lq < 1) return 0;
  const long wgs = static_cast<long>(N) * heads;
  if (wgs >= 65536) return 0;
  const dim3 grid(static_cast<unsigned>(wgs), static_cast<unsigned>((Lq + 127) / 128));
#define AM_LAUNCH(TT, DD)                                                                                                   \
  do {                                                                                                                      \
    const size_t lds = (32 * (DD + 8) + DD * 40) * sizeof(TT);                                                               \
    DS_RAISE_DYNAMIC_LDS((attention16_mfma_kernel<TT, DD>), 160 * 1024);                                                     \
    hipLaunchKernelGGL((attention16_mfma_kernel<TT, DD>), grid, dim3(256), lds, s, static_cast<const TT*>(q),                \
                       static_cast<const TT*>(k), static_cast<const TT*>(v), static_cast<TT*>(o), Lq, Lk, C, heads, scale);  \
  } while (0)
  if (dtype == DIFFSAL_BF16) { if (d == 192) AM_LAUNCH(__bf16, 192); else AM_LAUNCH(__bf16, 384); }
  else { if (d == 192) AM_LAUNCH(_Float16, 192); else AM_LAUNCH(_Float16, 384); }
#undef AM_LAUNCH
  const int rc = check_launch("attention(16-bit, MFMA)");
  return rc == DIFFSAL_OK ? 1 : rc;
}

}  // namespace diffsal
