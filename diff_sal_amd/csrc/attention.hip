// General softmax attention on the fp32 matrix cores (flash-style: one pass over the keys, online softmax):
//
//   out[b, l, h*DV + :] = softmax_t( [q | q_extra][b,h,l,:] . [k | k_extra][b,h,t,:] ) v[b,h,t,:]  (+ residual[b,h,l,:])
//
// Used by the once-per-clip encoders around the denoiser:
//   * MViTv2 pooling attention with decomposed relative-position bias, R/models/mvit.py:363-410 (add_decomposed_rel_pos)
//     and :548-605 (MultiScaleAttention.forward).  The bias  rel_t[q, kt] + rel_h[q, kh] + rel_w[q, kw]  is not added
//     element-wise: it is folded into the QK^T contraction by appending E extra columns -- q_extra = (q.Rt, q.Rh, q.Rw)
//     per query, k_extra = one-hot(kt), one-hot(kh), one-hot(kw) per key (a table shared by all batches / heads) -- so
//     the bias costs E/D more MFMAs and zero vector instructions.  `residual` is the pooled q of residual pooling
//     (mvit.py:596-600), `skip_first` leaves row 0 (the class token) without it.
//   * AudioAttnNet's self-attention, R/models/audio_attention.py:30-60 (heads 2, dim_head 64, 756 tokens).
//
// Work split: a workgroup = 4 wavefronts = 128 queries of one (batch, head); each wavefront owns 32 queries.  K and V
// tiles of 32 keys are staged in LDS (shared by the 4 waves).  Everything is computed TRANSPOSED so that a lane always
// works for ONE query:  S^T = K Q^T  has C/D layout (column = lane & 31 = query, 16 rows per lane = keys), so the
// softmax maxima / sums over keys are in-lane reductions plus one exchange with the partner lane (lane ^ 32), the
// rescale of the running output is a per-lane scalar, and P^T is already in the B-operand layout of the second product
// O^T = V^T P^T  (v_mfma_f32_32x32x2_f32 takes its two k values from the two lane halves: the k order of a contraction
// is free as long as A and B agree, so half h pairs "its" key rows {4h + (r & 3) + 8 (r >> 2)} with the same rows of V).
// Exact fp32 products and sums; exp via expf.
#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
  const float* q;        // [B,H,Lq,D] via strides
  const float* q_extra;  // [B,H,Lq,E] contiguous, or null when E == 0
  const float* k;
  const float* k_extra;  // [Lk,E] contiguous (shared), or null
  const float* v;
  const float* residual; // same indexing as q, or null
  float* out;            // [B, Lq, H*DV]
  long q_sb, q_sh, q_sl;  // element strides of q (batch, head, row)
  long k_sb, k_sh, k_sl;
  long v_sb, v_sh, v_sl;
  long r_sb, r_sh, r_sl;
  int H, Lq, Lk;
  float scale;
  int skip_first;
};

template <int D, int E, int DV>
__global__ __launch_bounds__(256) void attention_fwd_kernel(AttnArgs p) {
  constexpr int DQ = D + E;             // contraction length of QK^T
  static_assert(DQ % 8 == 0 && DV % 32 == 0 && D % 4 == 0 && E % 4 == 0, "shape");
  constexpr int HQ = DQ / 2;            // per lane half
  constexpr int KP = DQ + 4;            // LDS pitches (floats)
  constexpr int VP = DV + 4;
  constexpr int NT = DV / 32;
  __shared__ __attribute__((aligned(16))) float Ks[32 * KP];
  __shared__ __attribute__((aligned(16))) float Vs[32 * VP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int ql = lane & 31, hf = lane >> 5;
  const int qi = q0 + ql;
  const int qc = qi < p.Lq ? qi : p.Lq - 1;   // clamped: lanes past the end compute garbage that is never stored

  // ---- this lane's query fragment: elements [hf*HQ, hf*HQ + HQ) of [q * scale | q_extra]
  float qf[HQ];
  {
    const float* qr = p.q + b * p.q_sb + h * p.q_sh + static_cast<long>(qc) * p.q_sl;
    const float* qe = E ? p.q_extra + (static_cast<long>(bh) * p.Lq + qc) * E : nullptr;
#pragma unroll
    for (int j4 = 0; j4 < HQ / 4; ++j4) {
      const int e0 = hf * HQ + j4 * 4;   // D, HQ multiples of 4: a float4 piece never straddles q / q_extra
      float4 t;
      if (e0 < D) {
        t = ld4(qr + e0);
        t.x *= p.scale; t.y *= p.scale; t.z *= p.scale; t.w *= p.scale;
      } else {
        t = ld4(qe + (e0 - D));
      }
      qf[j4 * 4 + 0] = t.x; qf[j4 * 4 + 1] = t.y; qf[j4 * 4 + 2] = t.z; qf[j4 * 4 + 3] = t.w;
    }
  }

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -3.0e38f, l_run = 0.f;   // running max (shared by the two halves) and this lane's partial sum

  const float* kb = p.k + b * p.k_sb + h * p.k_sh;
  const float* vb = p.v + b * p.v_sb + h * p.v_sh;
  const int n_tiles = (p.Lk + 31) / 32;
  constexpr int KF4 = 32 * DQ / 4, VF4 = 32 * DV / 4;          // float4 pieces per tile
  constexpr int KPT = (KF4 + 255) / 256, VPT = (VF4 + 255) / 256;
  float4 kreg[KPT], vreg[VPT];
  auto fetch = [&](int tile) {
    const int key0 = tile * 32;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DQ / 4), c4 = (idx - row * (DQ / 4)) * 4;
      const int key = key0 + row;
      float4 t = make_float4(0, 0, 0, 0);
      if (idx < KF4 && key < p.Lk) t = c4 < D ? ld4(kb + static_cast<long>(key) * p.k_sl + c4)
                                             : ld4(p.k_extra + static_cast<long>(key) * E + (c4 - D));
      kreg[i] = t;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DV / 4), c4 = (idx - row * (DV / 4)) * 4;
      const int key = key0 + row;
      float4 t = make_float4(0, 0, 0, 0);
      if (idx < VF4 && key < p.Lk) t = ld4(vb + static_cast<long>(key) * p.v_sl + c4);
      vreg[i] = t;
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DQ / 4), c4 = (idx - row * (DQ / 4)) * 4;
      if (idx < KF4) st4(&Ks[row * KP + c4], kreg[i]);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DV / 4), c4 = (idx - row * (DV / 4)) * 4;
      if (idx < VF4) st4(&Vs[row * VP + c4], vreg[i]);
    }
  };

  fetch(0);
  park();
  __syncthreads();
  for (int tile = 0; tile < n_tiles; ++tile) {
    if (tile + 1 < n_tiles) fetch(tile + 1);   // lands while this tile is computed
    // ---- S^T = K Q^T : A = K[key = lane & 31][hf*HQ + j] (LDS), B = qf[j]
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* krow = &Ks[ql * KP + hf * HQ];
#pragma unroll
    for (int j4 = 0; j4 < HQ / 4; ++j4) {
      const float4 a = ld4(krow + j4 * 4);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[j4 * 4 + 0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[j4 * 4 + 1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[j4 * 4 + 2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[j4 * 4 + 3], s, 0, 0, 0);
    }
    // ---- online softmax for this lane's query over its 16 keys (+ the partner half's 16)
    const int key_base = tile * 32 + hf * 4;
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key_base + (r & 3) + 8 * (r >> 2);
      s[r] = key < p.Lk ? s[r] : -3.0e38f;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key_base + (r & 3) + 8 * (r >> 2);
      const float e = key < p.Lk ? expf(s[r] - m_new) : 0.f;
      s[r] = e;
      ps += e;
    }
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // ---- O^T += V^T P^T : A = V[key row of (r, hf)][32 t + (lane & 31)] (LDS), B = p[r]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow_i = hf * 4 + (r & 3) + 8 * (r >> 2);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[krow_i * VP + t * 32 + ql], s[r], o[t], 0, 0, 0);
    }
    __syncthreads();                 // every wave is done with this tile
    if (tile + 1 < n_tiles) {
      park();
      __syncthreads();
    }
  }

  // ---- finish: out[q][d] = O^T[d][q] / l (+ residual); this lane holds d = 32 t + 4 hf + (r & 3) + 8 (r >> 2)
  const float l_tot = l_run + __shfl_xor(l_run, 32, kWave);
  const float inv = 1.0f / l_tot;
  if (qi < p.Lq) {
    float* orow = p.out + (static_cast<long>(b) * p.Lq + qi) * (static_cast<long>(p.H) * DV) + h * DV;
    const bool add_res = p.residual && !(p.skip_first && qi == 0);
    const float* rr = p.residual ? p.residual + b * p.r_sb + h * p.r_sh + static_cast<long>(qi) * p.r_sl : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 4 * hf + 8 * g;
        float4 v4 = make_float4(o[t][4 * g + 0] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
        if (add_res) {
          const float4 r4 = ld4(rr + d);
          v4.x += r4.x; v4.y += r4.y; v4.z += r4.z; v4.w += r4.w;
        }
        st4(orow + d, v4);
      }
  }
}

}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_attention_general(const float* q, const float* q_extra, const float* k, const float* k_extra,
                                         const float* v, const float* residual, float* out, int B, int H, int Lq, int Lk,
                                         int D, int E, int DV, const long* q_strides, const long* k_strides,
                                         const long* v_strides, const long* r_strides, float scale, int skip_first,
                                         diffsal_stream_t stream) {
  DS_REQUIRE(q && k && v && out && q_strides && k_strides && v_strides, DIFFSAL_E_ARG, "attention_general: null argument");
  DS_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && static_cast<long>(B) * H < 65536, DIFFSAL_E_SHAPE,
             "attention_general: bad shape B=%d H=%d Lq=%d Lk=%d", B, H, Lq, Lk);
  DS_REQUIRE((E == 0) == (q_extra == nullptr) && (E == 0) == (k_extra == nullptr), DIFFSAL_E_ARG,
             "attention_general: q_extra / k_extra must be given exactly when E > 0");
  DS_REQUIRE(!residual || r_strides, DIFFSAL_E_ARG, "attention_general: residual needs its strides");
  DS_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && (!residual || aligned16(residual)) &&
                 (!q_extra || aligned16(q_extra)) && (!k_extra || aligned16(k_extra)),
             DIFFSAL_E_ALIGN, "attention_general: misaligned pointer");
  for (int i = 0; i < 3; ++i)
    DS_REQUIRE(q_strides[i] % 4 == 0 && k_strides[i] % 4 == 0 && v_strides[i] % 4 == 0 && (!residual || r_strides[i] % 4 == 0),
               DIFFSAL_E_ALIGN, "attention_general: strides must be multiples of 4 elements");
  AttnArgs a;
  a.q = q; a.q_extra = q_extra; a.k = k; a.k_extra = k_extra; a.v = v; a.residual = residual; a.out = out;
  a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sl = q_strides[2];
  a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sl = k_strides[2];
  a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sl = v_strides[2];
  a.r_sb = residual ? r_strides[0] : 0; a.r_sh = residual ? r_strides[1] : 0; a.r_sl = residual ? r_strides[2] : 0;
  a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.skip_first = skip_first;
  const dim3 grid((Lq + 127) / 128, B * H);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (D == 96 && E == 48 && DV == 96) hipLaunchKernelGGL((attention_fwd_kernel<96, 48, 96>), grid, dim3(256), 0, s, a);
  else if (D == 96 && E == 0 && DV == 96) hipLaunchKernelGGL((attention_fwd_kernel<96, 0, 96>), grid, dim3(256), 0, s, a);
  else if (D == 64 && E == 0 && DV == 64) hipLaunchKernelGGL((attention_fwd_kernel<64, 0, 64>), grid, dim3(256), 0, s, a);
  else if (D == 32 && E == 0 && DV == 32) hipLaunchKernelGGL((attention_fwd_kernel<32, 0, 32>), grid, dim3(256), 0, s, a);
  else {
    set_error("attention_general: (D, E, DV) = (%d, %d, %d) is not built: (96,48,96), (96,0,96), (64,0,64), (32,0,32)", D, E, DV);
    return DIFFSAL_E_SHAPE;
  }
  return check_launch("attention_general");
}
