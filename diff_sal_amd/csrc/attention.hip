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

// Staging discipline of the three kernels below.  The next tile's pieces are fetched with UNCONDITIONAL 16-byte loads from
// addresses clamped into valid memory and kept raw in registers; whether a piece is real (inside the tile, key / query inside
// the sequence) is decided when it is parked in LDS, after the tile's MFMAs.  Any use of a loaded value next to its load
// (a select, a scale) made hipcc wait for every load before issuing the next one -- twelve dependent L2 round trips per tile
// in front of the matrix work, with one wave per SIMD nothing to hide them.  LDS fragments are read one group ahead of the
// MFMAs that consume them (mfma_groups_f32 / mfma_groups_scalar_f32, common.h) for the same reason.
__device__ __forceinline__ float4 keep_or_zero(float4 t, bool ok) { return ok ? t : make_float4(0.f, 0.f, 0.f, 0.f); }

struct AttnArgs {
  const float* q;        // [B,H,Lq,D] via strides
  const float* q_extra;  // [B,H,Lq,E] contiguous, or null when E == 0
  const float* k;
  const float* k_extra;  // [Lk,E] contiguous (shared), or null
  // slot form of a ONE-HOT k_extra (MViT's decomposed relative-position bias: a key has one column per axis): k_slots[key][0..2] =
  // the columns that are 1 (E = none: the class token), [3] unused.  The bias  q_extra . k_extra[key]  is then the sum of (at most)
  // three gathered q_extra values, added to S on the vector unit from LDS, and the E / D extra MFMAs of the contraction form (a fifth
  // of the forward's at E = 48) are not issued: forward 565 -> 515 us on MViT's first stage.  Forward only: in the dk / dv kernel
  // (one wavefront per SIMD) the gathers cost what the 24 MFMAs saved, and a dq kernel that sums dS per column by LDS adds ran 4x slower
  // (ds_add_f32 retires lane by lane) -- both measured and removed (profiles/NOTES.md).
  const int* k_slots;
  const float* v;
  const float* residual; // same indexing as q, or null
  float* out;            // [B, Lq, H*DV]
  float* lse;            // [B,H,Lq] log-sum-exp of every row (training: the backward recomputes P from it), or null
  long q_sb, q_sh, q_sl;  // element strides of q (batch, head, row)
  long k_sb, k_sh, k_sl;
  long v_sb, v_sh, v_sl;
  long r_sb, r_sh, r_sl;
  int H, Lq, Lk;
  float scale;
  int skip_first;
  // tail mode (as in the dq kernel of the backward): blocks [0, qb_full) of the (query tile, batch*head) list run whole; each
  // later block is cut into qb_split pieces over the key tiles, which leave their normalised partial output and its
  // log-sum-exp in the slabs; attention_fwd_tail_kernel merges them (softmax of the pieces' log-sum-exps)
  int qb_x, qb_full, qb_split, qb_bh;
  float* o_slabs;        // [qb_split][B][Lq][H*DV]
  float* l_slabs;        // [qb_split][B*H][Lq]
};

template <int D, int E, int DV, bool SL = false>
__global__ __launch_bounds__(256, 2) void attention_fwd_kernel(AttnArgs p) {
  constexpr int DQ = SL ? D : D + E;    // contraction length of QK^T (slot form: the bias is gathered, not contracted)
  static_assert(DQ % 8 == 0 && DV % 32 == 0 && D % 4 == 0 && E % 8 == 0 && (!SL || E > 0), "shape");
  constexpr int HQ = DQ / 2;            // per lane half
  constexpr int KP = DQ + 4;            // LDS pitches (floats)
  constexpr int VP = DV + 4;
  constexpr int NT = DV / 32;
  __shared__ __attribute__((aligned(16))) float Ks[32 * KP];
  __shared__ __attribute__((aligned(16))) float Vs[32 * VP];
  // slot form: this wavefront's q_extra transposed [column][query] (+ an all-zero column E) and the key tile's slots
  __shared__ float QeT[SL ? 4 : 1][SL ? (E + 1) * 32 : 1];
  __shared__ int4 Ss[SL ? 32 : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int blk = blockIdx.x, piece = -1;
  if (blk >= p.qb_full) {
    const int t = blk - p.qb_full;
    blk = p.qb_full + t / p.qb_split;
    piece = t - (blk - p.qb_full) * p.qb_split;
  }
  const int bh = blk / p.qb_x, b = bh / p.H, h = bh - b * p.H;
  const int q0 = (blk - bh * p.qb_x) * 128 + wave * 32;
  const int ql = lane & 31, hf = lane >> 5;
  const int qi = q0 + ql;
  const int qc = qi < p.Lq ? qi : p.Lq - 1;   // clamped: lanes past the end compute garbage that is never stored

  // ---- this lane's query fragment: elements [hf*HQ, hf*HQ + HQ) of [q * scale | q_extra]
  float qf[HQ];
  {
    const float* qr = p.q + b * p.q_sb + h * p.q_sh + static_cast<long>(qc) * p.q_sl;
    const float* qe = E ? p.q_extra + (static_cast<long>(bh) * p.Lq + qc) * E : nullptr;
    // all pieces are requested before any is touched (a scale or a branch next to a load serialises the loads)
    float4 raw[HQ / 4];
#pragma unroll
    for (int j4 = 0; j4 < HQ / 4; ++j4) {
      const int e0 = hf * HQ + j4 * 4;   // D, HQ multiples of 4: a float4 piece never straddles q / q_extra
      const float* src = qr + (e0 < D ? e0 : 0);
      if constexpr (E > 0 && !SL) src = e0 < D ? src : qe + (e0 - D);
      raw[j4] = ld4(src);
    }
    if constexpr (SL) {                  // this lane's half of its query's q_extra row -> QeT[column][query]
      float4 re[E / 8];
#pragma unroll
      for (int j4 = 0; j4 < E / 8; ++j4) re[j4] = ld4(qe + hf * (E / 2) + j4 * 4);
      float* qt = &QeT[wave][0];
#pragma unroll
      for (int j4 = 0; j4 < E / 8; ++j4) {
        const int c = hf * (E / 2) + j4 * 4;
        qt[(c + 0) * 32 + ql] = re[j4].x; qt[(c + 1) * 32 + ql] = re[j4].y; qt[(c + 2) * 32 + ql] = re[j4].z; qt[(c + 3) * 32 + ql] = re[j4].w;
      }
      if (hf == 0) qt[E * 32 + ql] = 0.f;
    }
#pragma unroll
    for (int j4 = 0; j4 < HQ / 4; ++j4) {
      const float sc = hf * HQ + j4 * 4 < D ? p.scale : 1.0f;
      qf[j4 * 4 + 0] = raw[j4].x * sc; qf[j4 * 4 + 1] = raw[j4].y * sc; qf[j4 * 4 + 2] = raw[j4].z * sc; qf[j4 * 4 + 3] = raw[j4].w * sc;
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
  const int all_tiles = (p.Lk + 31) / 32;
  const int per_piece = (all_tiles + p.qb_split - 1) / p.qb_split;
  const int t_begin = piece < 0 ? 0 : piece * per_piece;
  const int n_tiles = piece < 0 ? all_tiles : min(all_tiles, t_begin + per_piece);      // one past the last key tile of this block / piece
  constexpr int KF4 = 32 * DQ / 4, VF4 = 32 * DV / 4;          // float4 pieces per tile
  constexpr int KPT = (KF4 + 255) / 256, VPT = (VF4 + 255) / 256;
  float4 kreg[KPT], vreg[VPT];
  int4 sreg = make_int4(E, E, E, E);
  auto fetch = [&](int tile) {
    const int key0 = tile * 32;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DQ / 4), c4 = (idx - row * (DQ / 4)) * 4;
      const int key = key0 + (row < 32 ? row : 31);              // pieces past the tile read a valid row (never parked)
      const long kc = key < p.Lk ? key : p.Lk - 1;
      const float* src = kb + kc * p.k_sl + (c4 < D ? c4 : 0);
      if constexpr (E > 0 && !SL) src = c4 < D ? src : p.k_extra + kc * E + (c4 - D);
      kreg[i] = ld4(src);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DV / 4), c4 = (idx - row * (DV / 4)) * 4;
      const int key = key0 + (row < 32 ? row : 31);
      const long kc = key < p.Lk ? key : p.Lk - 1;
      vreg[i] = ld4(vb + kc * p.v_sl + c4);
    }
    if constexpr (SL) {
      if (tid < 32) {
        const int key = key0 + tid;
        sreg = *reinterpret_cast<const int4*>(p.k_slots + 4L * (key < p.Lk ? key : p.Lk - 1));
      }
    }
  };
  auto park = [&](int tile) {
    const int key0 = tile * 32;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DQ / 4), c4 = (idx - row * (DQ / 4)) * 4;
      if (idx < KF4) st4(&Ks[row * KP + c4], keep_or_zero(kreg[i], key0 + row < p.Lk));
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DV / 4), c4 = (idx - row * (DV / 4)) * 4;
      if (idx < VF4) st4(&Vs[row * VP + c4], keep_or_zero(vreg[i], key0 + row < p.Lk));
    }
    if constexpr (SL) {
      if (tid < 32) Ss[tid] = key0 + tid < p.Lk ? sreg : make_int4(E, E, E, E);
    }
  };

  fetch(t_begin);
  park(t_begin);
  __syncthreads();
  for (int tile = t_begin; tile < n_tiles; ++tile) {
    if (tile + 1 < n_tiles) fetch(tile + 1);   // lands while this tile is computed
    // ---- S^T = K Q^T : A = K[key = lane & 31][hf*HQ + j] (LDS), B = qf[j]
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* krow = &Ks[ql * KP + hf * HQ];
    mfma_groups_f32<HQ / 4>([&](int j4) { return krow + j4 * 4; },
                            [&](int j4, float4 a) { DS_MFMA4(s, a, qf[j4 * 4 + 0], qf[j4 * 4 + 1], qf[j4 * 4 + 2], qf[j4 * 4 + 3]); });
    if constexpr (SL) {              // + q_extra[query][the key's columns]
      const float* qt = &QeT[wave][ql];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int4 sl = Ss[hf * 4 + (r & 3) + 8 * (r >> 2)];
        s[r] += (qt[sl.x * 32] + qt[sl.y * 32]) + qt[sl.z * 32];
      }
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
    mx = fmaxf(mx, lane_xor32(mx));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key_base + (r & 3) + 8 * (r >> 2);
      const float e = key < p.Lk ? __expf(s[r] - m_new) : 0.f;   // v_exp_f32 form: ~1e-6 relative, a quarter of the VALU work of expf
      s[r] = e;
      ps += e;
    }
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    // ---- O^T += V^T P^T : A = V[key row of (r, hf)][32 t + (lane & 31)] (LDS, one scalar per MFMA, eight read ahead), B = p[r]
    mfma_groups_scalar_f32<2 * NT, 8>(                         // group = (t, half of the 16 key rows): two resident waves share 512 registers
        [&](int g, int j) { const int r = 8 * (g & 1) + j; return &Vs[(hf * 4 + (r & 3) + 8 * (r >> 2)) * VP + (g >> 1) * 32 + ql]; },
        [&](int g, const float (&a)[8]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[g >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], s[8 * (g & 1) + j], o[g >> 1], 0, 0, 0);
        });
    __syncthreads();                 // every wave is done with this tile
    if (tile + 1 < n_tiles) {
      park(tile + 1);
      __syncthreads();
    }
  }

  // ---- finish: out[q][d] = O^T[d][q] / l (+ residual); this lane holds d = 32 t + 4 hf + (r & 3) + 8 (r >> 2)
  const float l_tot = l_run + lane_xor32(l_run);
  const float inv = 1.0f / l_tot;
  if (piece >= 0) {                  // tail mode: this piece's normalised output and log-sum-exp, merged by attention_fwd_tail_kernel
    if (qi < p.Lq) {
      if (hf == 0) p.l_slabs[(static_cast<long>(piece) * p.qb_bh + bh) * p.Lq + qi] = m_run + logf(l_tot);
      float* orow = p.o_slabs + ((static_cast<long>(piece) * (p.qb_bh / p.H) + b) * p.Lq + qi) * (static_cast<long>(p.H) * DV) + h * DV;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          st4(orow + 32 * t + 4 * hf + 8 * g,
              make_float4(o[t][4 * g + 0] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv));
    }
    return;
  }
  if (p.lse && qi < p.Lq && hf == 0) p.lse[static_cast<long>(bh) * p.Lq + qi] = m_run + logf(l_tot);
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

// tail mode of the forward: rows of the blocks >= qb_full: out = sum_i w_i O_i (+ residual), w = softmax over the pieces'
// log-sum-exps; lse = log sum_i exp(lse_i).  One thread per (row, 4 channels).
template <int DV>
__global__ __launch_bounds__(256) void attention_fwd_tail_kernel(AttnArgs p, int n_tail_blocks) {
  constexpr int C4 = DV / 4;
  const long items = static_cast<long>(n_tail_blocks) * 128 * C4;
  const long l_slab = static_cast<long>(p.qb_bh) * p.Lq;
  const long o_slab = l_slab * DV;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < items; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % C4) * 4;
    const long r = i / C4;
    const int blk = p.qb_full + static_cast<int>(r >> 7);
    const int bh = blk / p.qb_x, b = bh / p.H, h = bh - b * p.H;
    const int qi = (blk - bh * p.qb_x) * 128 + static_cast<int>(r & 127);
    if (qi >= p.Lq) continue;
    const long lrow = static_cast<long>(bh) * p.Lq + qi;
    const long orow = (static_cast<long>(b) * p.Lq + qi) * (static_cast<long>(p.H) * DV) + h * DV + c;
    float m = -3.0e38f;
    for (int s = 0; s < p.qb_split; ++s) m = fmaxf(m, p.l_slabs[s * l_slab + lrow]);
    float den = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < p.qb_split; ++s) {
      const float w = __expf(p.l_slabs[s * l_slab + lrow] - m);
      const float4 u = ld4(p.o_slabs + s * o_slab + orow);
      den += w;
      acc.x += w * u.x; acc.y += w * u.y; acc.z += w * u.z; acc.w += w * u.w;
    }
    const float inv = 1.0f / den;
    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    if (p.residual && !(p.skip_first && qi == 0)) {
      const float4 r4 = ld4(p.residual + b * p.r_sb + h * p.r_sh + static_cast<long>(qi) * p.r_sl + c);
      acc.x += r4.x; acc.y += r4.y; acc.z += r4.z; acc.w += r4.w;
    }
    st4(p.out + orow, acc);
    if (p.lse && c == 0) p.lse[lrow] = m + logf(den);
  }
}

// =================================================================================================================
// Backward of the attention above (training of the encoders).  Flash-style: P is recomputed from the
// saved row log-sum-exp; delta[q] = sum_d dO[q,d] * O_attn[q,d] (O_attn = out - residual) comes from a small pre-pass.
//   dS = P * (dP - delta),  dP = dO V^T
//   dq = scale * dS K (+ dO on residual rows),  dq_extra = dS k_extra,  dK = scale-folded dS^T Q',  dV = P^T dO
// Two kernels, both in the transposed one-lane-one-row formulation of the forward:
//   attention_bwd_q_kernel : a wavefront owns 32 queries, loops over key tiles   -> dq, dq_extra
//   attention_bwd_kv_kernel: a wavefront owns 32 keys,    loops over query tiles -> dk, dv
// No atomics; every output element is produced by exactly one lane (deterministic).
// =================================================================================================================
struct AttnBwdArgs {
  const float* q; const float* q_extra; const float* k; const float* k_extra; const float* v;
  const float* dout;     // [B, Lq, H*DV]
  const float* out;      // forward output (same layout as dout)
  const float* residual; // forward residual or null
  const float* lse;      // [B,H,Lq]
  float* delta;          // [B,H,Lq] scratch
  float* dq;             // [B,H,Lq,D] contiguous
  float* dq_extra;       // [B,H,Lq,E] contiguous or null
  float* dk;             // [B,H,Lk,D] contiguous
  float* dv;             // [B,H,Lk,DV] contiguous
  float* kv_part;        // [q_splits][B*H*Lk*(D+DV)] partial dk | dv when q_splits > 1 (summed by attention_bwd_kv_sum_kernel)
  int q_splits;          // the kv kernel's query range is cut into this many pieces (few keys, many queries: MViT's early layers)
  // dq kernel, tail mode: blocks [0, qb_full) of the (query tile, batch*head) list run whole; each later block is cut into
  // qb_split pieces over the key tiles, which write their partial dq | dq_extra rows to q_slabs (attention_bwd_q_tail_kernel
  // adds them up): the last, partly filled round of workgroups (one per CU) fills the chip instead of a third of it.
  int qb_x, qb_full, qb_split, qb_bh;
  float* q_slabs;        // [qb_split][B*H*Lq][D + E]
  // dS form: the dk / dv kernel leaves dS [B*H][Lq][Lkp] (Lkp = Lk rounded up to 32, zeros past Lk) and the dq kernel is the
  // one product dS K' left over -- 4 + 1 tile products per (query, key) tile pair instead of 4 + 3 (288 GB of HBM: the 0.5 GB
  // of the largest MViT layer is written and read once at a quarter of the memory rate while the matrix pipe works)
  float* ds;
  int Lkp;
  long q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, r_sb, r_sh, r_sl;
  int H, Lq, Lk;
  float scale;
  int skip_first;
};

// delta[b,h,l] = sum_d dout[b,l,h*DV+d] * (out[b,l,h*DV+d] - residual[b,h,l,d]); G lanes per row.
template <int DV>
__global__ __launch_bounds__(256) void attention_bwd_delta_kernel(AttnBwdArgs p, long rows) {
  constexpr int G = 8;
  const int gl = threadIdx.x % G;
  const long row = static_cast<long>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  const bool live = row < rows;
  const long rc = live ? row : rows - 1;
  const int l = static_cast<int>(rc % p.Lq);
  const long bh = rc / p.Lq;
  const int h = static_cast<int>(bh % p.H), b = static_cast<int>(bh / p.H);
  const long o = (static_cast<long>(b) * p.Lq + l) * (static_cast<long>(p.H) * DV) + h * DV;
  const bool has_res = p.residual && !(p.skip_first && l == 0);
  const float* rr = p.residual ? p.residual + b * p.r_sb + h * p.r_sh + static_cast<long>(l) * p.r_sl : nullptr;
  float s = 0.f;
  for (int d = gl * 4; d < DV; d += G * 4) {
    const float4 g = ld4(p.dout + o + d);
    float4 y = ld4(p.out + o + d);
    if (has_res) { const float4 r4 = ld4(rr + d); y.x -= r4.x; y.y -= r4.y; y.z -= r4.z; y.w -= r4.w; }
    s += (g.x * y.x + g.y * y.y) + (g.z * y.z + g.w * y.w);
  }
  s = group_sum<G>(s);
  if (live && gl == 0) p.delta[rc] = s;
}

// batch * heads of a launch (stored with the dq tail plan)
__device__ __forceinline__ int gridDimBH(const AttnBwdArgs& p) { return p.qb_bh; }

// tail mode of the dq kernel: rows of the blocks >= qb_full: dq | dq_extra = sum of the pieces' partial rows (+ the residual path)
template <int D, int E, int DV>
__global__ __launch_bounds__(256) void attention_bwd_q_tail_kernel(AttnBwdArgs p, int n_tail_blocks) {
  constexpr int DQ = D + E, C4 = DQ / 4;
  const long items = static_cast<long>(n_tail_blocks) * 128 * C4;
  const long slab = static_cast<long>(p.qb_bh) * p.Lq * DQ;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < items; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % C4) * 4;
    const long r = i / C4;
    const int blk = p.qb_full + static_cast<int>(r >> 7);
    const int bh = blk / p.qb_x, b = bh / p.H, h = bh - b * p.H;
    const int qi = (blk - bh * p.qb_x) * 128 + static_cast<int>(r & 127);
    if (qi >= p.Lq) continue;
    const long row = static_cast<long>(bh) * p.Lq + qi;
    float4 v = ld4(p.q_slabs + row * DQ + c);
    for (int s = 1; s < p.qb_split; ++s) {
      const float4 u = ld4(p.q_slabs + s * slab + row * DQ + c);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (c < D) {
      if (p.residual && !(p.skip_first && qi == 0)) {
        const float4 r4 = ld4(p.dout + (static_cast<long>(b) * p.Lq + qi) * (static_cast<long>(p.H) * DV) + h * DV + c);
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
      }
      st4(p.dq + row * D + c, v);
    } else {
      st4(p.dq_extra + row * E + (c - D), v);
    }
  }
}

template <int D, int E, int DV>
__global__ __launch_bounds__(256) void attention_bwd_q_kernel(AttnBwdArgs p) {
  constexpr int DQ = D + E;
  constexpr int HQ = DQ / 2, HV = DV / 2;
  constexpr int NQT = (DQ + 31) / 32;            // output tiles of dQ'^T (rows = contraction index of QK^T)
  constexpr int KP = NQT * 32 + 4;               // K' rows padded with zeros up to a whole tile
  constexpr int VP = DV + 4;
  __shared__ __attribute__((aligned(16))) float Ks[32 * KP];
  __shared__ __attribute__((aligned(16))) float Vs[32 * VP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int blk = blockIdx.x, piece = -1;
  if (blk >= p.qb_full) {
    const int t = blk - p.qb_full;
    blk = p.qb_full + t / p.qb_split;
    piece = t - (blk - p.qb_full) * p.qb_split;
  }
  const int bh = blk / p.qb_x, b = bh / p.H, h = bh - b * p.H;
  const int ql = lane & 31, hf = lane >> 5;
  const int qi = (blk - bh * p.qb_x) * 128 + wave * 32 + ql;
  const int qc = qi < p.Lq ? qi : p.Lq - 1;

  float qf[HQ], gf[HV];   // this lane's halves of [q*scale | q_extra] and of dO
  {
    const float* qr = p.q + b * p.q_sb + h * p.q_sh + static_cast<long>(qc) * p.q_sl;
    const float* qe = E ? p.q_extra + (static_cast<long>(bh) * p.Lq + qc) * E : nullptr;
    // all pieces are requested before any is touched (a scale or a branch next to a load serialises the loads)
    float4 raw[HQ / 4];
#pragma unroll
    for (int j4 = 0; j4 < HQ / 4; ++j4) {
      const int e0 = hf * HQ + j4 * 4;
      const float* src = qr + (e0 < D ? e0 : 0);
      if constexpr (E > 0) src = e0 < D ? src : qe + (e0 - D);
      raw[j4] = ld4(src);
    }
#pragma unroll
    for (int j4 = 0; j4 < HQ / 4; ++j4) {
      const float sc = hf * HQ + j4 * 4 < D ? p.scale : 1.0f;
      qf[j4 * 4 + 0] = raw[j4].x * sc; qf[j4 * 4 + 1] = raw[j4].y * sc; qf[j4 * 4 + 2] = raw[j4].z * sc; qf[j4 * 4 + 3] = raw[j4].w * sc;
    }
    const float* gr = p.dout + (static_cast<long>(b) * p.Lq + qc) * (static_cast<long>(p.H) * DV) + h * DV + hf * HV;
#pragma unroll
    for (int j4 = 0; j4 < HV / 4; ++j4) {
      const float4 t = ld4(gr + j4 * 4);
      gf[j4 * 4 + 0] = t.x; gf[j4 * 4 + 1] = t.y; gf[j4 * 4 + 2] = t.z; gf[j4 * 4 + 3] = t.w;
    }
  }
  const float lse = p.lse[static_cast<long>(bh) * p.Lq + qc];
  const float delta = p.delta[static_cast<long>(bh) * p.Lq + qc];

  f32x16 acc[NQT];
#pragma unroll
  for (int t = 0; t < NQT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* kb = p.k + b * p.k_sb + h * p.k_sh;
  const float* vb = p.v + b * p.v_sb + h * p.v_sh;
  const int all_tiles = (p.Lk + 31) / 32;
  const int per_piece = (all_tiles + p.qb_split - 1) / p.qb_split;
  const int t_begin = piece < 0 ? 0 : piece * per_piece;
  const int n_tiles = piece < 0 ? all_tiles : min(all_tiles, t_begin + per_piece);      // one past the last key tile of this block / piece
  constexpr int KC4 = NQT * 8;                                   // float4 pieces per padded K' row
  constexpr int KF4 = 32 * KC4, VF4 = 32 * DV / 4;
  constexpr int KPT = (KF4 + 255) / 256, VPT = (VF4 + 255) / 256;
  float4 kreg[KPT], vreg[VPT];
  auto fetch = [&](int tile) {
    const int key0 = tile * 32;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / KC4, c4 = (idx - row * KC4) * 4;
      const int key = key0 + (row < 32 ? row : 31);
      const long kc = key < p.Lk ? key : p.Lk - 1;
      const float* src = kb + kc * p.k_sl + (c4 < D ? c4 : 0);
      if constexpr (E > 0) src = (c4 < D || c4 >= DQ) ? src : p.k_extra + kc * E + (c4 - D);
      kreg[i] = ld4(src);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DV / 4), c4 = (idx - row * (DV / 4)) * 4;
      const int key = key0 + (row < 32 ? row : 31);
      const long kc = key < p.Lk ? key : p.Lk - 1;
      vreg[i] = ld4(vb + kc * p.v_sl + c4);
    }
  };
  auto park = [&](int tile) {
    const int key0 = tile * 32;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / KC4, c4 = (idx - row * KC4) * 4;
      if (idx < KF4) st4(&Ks[row * KP + c4], keep_or_zero(kreg[i], key0 + row < p.Lk && c4 < DQ));   // zero padding columns up to a tile
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DV / 4), c4 = (idx - row * (DV / 4)) * 4;
      if (idx < VF4) st4(&Vs[row * VP + c4], keep_or_zero(vreg[i], key0 + row < p.Lk));
    }
  };
  fetch(t_begin);
  park(t_begin);
  __syncthreads();
  for (int tile = t_begin; tile < n_tiles; ++tile) {
    if (tile + 1 < n_tiles) fetch(tile + 1);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    const float* krow = &Ks[ql * KP + hf * HQ];
    mfma_groups_f32<HQ / 4>([&](int j4) { return krow + j4 * 4; },
                            [&](int j4, float4 a) { DS_MFMA4(s, a, qf[j4 * 4 + 0], qf[j4 * 4 + 1], qf[j4 * 4 + 2], qf[j4 * 4 + 3]); });
    const float* vrow = &Vs[ql * VP + hf * HV];                 // dP^T = V dO^T
    mfma_groups_f32<HV / 4>([&](int j4) { return vrow + j4 * 4; },
                            [&](int j4, float4 a) { DS_MFMA4(dp, a, gf[j4 * 4 + 0], gf[j4 * 4 + 1], gf[j4 * 4 + 2], gf[j4 * 4 + 3]); });
    const int key_base = tile * 32 + hf * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key_base + (r & 3) + 8 * (r >> 2);
      const float pr = key < p.Lk ? __expf(s[r] - lse) : 0.f;
      s[r] = pr * (dp[r] - delta);                                // dS^T
    }
    // dQ'^T += K'^T dS^T : A = K'[key row of (r, hf)][32 t + (lane & 31)], one scalar per MFMA, sixteen read ahead
    mfma_groups_scalar_f32<NQT, 16>(
        [&](int t, int r) { return &Ks[(hf * 4 + (r & 3) + 8 * (r >> 2)) * KP + t * 32 + ql]; },
        [&](int t, const float (&a)[16]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], s[r], acc[t], 0, 0, 0);
        });
    __syncthreads();
    if (tile + 1 < n_tiles) {
      park(tile + 1);
      __syncthreads();
    }
  }
  if (qi >= p.Lq) return;
  if (piece >= 0) {       // partial sums of a tail piece: scaled, no residual (the tail kernel adds it once)
    float* sl = p.q_slabs + ((static_cast<long>(piece) * gridDimBH(p) + bh) * p.Lq + qi) * DQ;
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * t + 4 * hf + 8 * g;
        if (c >= DQ) continue;
        const float sc = c < D ? p.scale : 1.0f;
        st4(sl + c, make_float4(acc[t][4 * g + 0] * sc, acc[t][4 * g + 1] * sc, acc[t][4 * g + 2] * sc, acc[t][4 * g + 3] * sc));
      }
    return;
  }
  // this lane holds contraction rows c = 32 t + 4 hf + (r & 3) + 8 (r >> 2) of its query
  float* dqr = p.dq + (static_cast<long>(bh) * p.Lq + qi) * D;
  float* der = E ? p.dq_extra + (static_cast<long>(bh) * p.Lq + qi) * E : nullptr;
  const bool add_res = p.residual && !(p.skip_first && qi == 0);
  const float* gres = p.dout + (static_cast<long>(b) * p.Lq + qi) * (static_cast<long>(p.H) * DV) + h * DV;
#pragma unroll
  for (int t = 0; t < NQT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * t + 4 * hf + 8 * g;
      float4 v4 = make_float4(acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
      if (c < D) {
        v4.x *= p.scale; v4.y *= p.scale; v4.z *= p.scale; v4.w *= p.scale;
        if (add_res) { const float4 r4 = ld4(gres + c); v4.x += r4.x; v4.y += r4.y; v4.z += r4.z; v4.w += r4.w; }
        st4(dqr + c, v4);
      } else if (c < DQ) {
        st4(der + (c - D), v4);
      }
    }
}

// dq kernel of the dS form: dQ'^T += K'^T dS^T with dS read back from the dk / dv kernel's buffer -- the last third of
// attention_bwd_q_kernel (same tiles, same tail mode, same lane ownership: the sums run in the same order, so dq is
// bit-identical to the recomputing kernel's given the same dS).
template <int D, int E, int DV>
__global__ __launch_bounds__(256, 2) void attention_bwd_q_ds_kernel(AttnBwdArgs p) {
  constexpr int DQ = D + E;
  constexpr int NQT = (DQ + 31) / 32;
  constexpr int KP = NQT * 32 + 4;
  __shared__ __attribute__((aligned(16))) float Ks2[2][32 * KP];   // two stages, one barrier per key tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int blk = blockIdx.x, piece = -1;
  if (blk >= p.qb_full) {
    const int t = blk - p.qb_full;
    blk = p.qb_full + t / p.qb_split;
    piece = t - (blk - p.qb_full) * p.qb_split;
  }
  const int bh = blk / p.qb_x, b = bh / p.H, h = bh - b * p.H;
  const int ql = lane & 31, hf = lane >> 5;
  const int qi = (blk - bh * p.qb_x) * 128 + wave * 32 + ql;
  const int qc = qi < p.Lq ? qi : p.Lq - 1;

  f32x16 acc[NQT];
#pragma unroll
  for (int t = 0; t < NQT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* kb = p.k + b * p.k_sb + h * p.k_sh;
  const float* dsrow = p.ds + (static_cast<long>(bh) * p.Lq + qc) * p.Lkp + hf * 4;
  const int all_tiles = (p.Lk + 31) / 32;
  const int per_piece = (all_tiles + p.qb_split - 1) / p.qb_split;
  const int t_begin = piece < 0 ? 0 : piece * per_piece;
  const int n_tiles = piece < 0 ? all_tiles : min(all_tiles, t_begin + per_piece);
  constexpr int KC4 = NQT * 8;
  constexpr int KF4 = 32 * KC4;
  constexpr int KPT = (KF4 + 255) / 256;
  float4 kreg[KPT], dreg[4];
  auto fetch = [&](int tile) {
    const int key0 = tile * 32;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / KC4, c4 = (idx - row * KC4) * 4;
      const int key = key0 + (row < 32 ? row : 31);
      const long kc = key < p.Lk ? key : p.Lk - 1;
      const float* src = kb + kc * p.k_sl + (c4 < D ? c4 : 0);
      if constexpr (E > 0) src = (c4 < D || c4 >= DQ) ? src : p.k_extra + kc * E + (c4 - D);
      kreg[i] = ld4(src);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) dreg[g] = ld4(dsrow + key0 + 8 * g);     // keys hf*4 + 8 g + (0..3) of this lane's query
  };
  auto park = [&](int tile, int st) {
    float* Ks = Ks2[st];
    const int key0 = tile * 32;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / KC4, c4 = (idx - row * KC4) * 4;
      if (idx < KF4) st4(&Ks[row * KP + c4], keep_or_zero(kreg[i], key0 + row < p.Lk && c4 < DQ));
    }
  };
  fetch(t_begin);
  park(t_begin, 0);
  __syncthreads();
  for (int tile = t_begin; tile < n_tiles; ++tile) {
    const int cur = (tile - t_begin) & 1;
    const float* Ks = Ks2[cur];
    float s[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) { s[4 * g + 0] = dreg[g].x; s[4 * g + 1] = dreg[g].y; s[4 * g + 2] = dreg[g].z; s[4 * g + 3] = dreg[g].w; }
    if (tile + 1 < n_tiles) fetch(tile + 1);
    mfma_groups_scalar_f32<NQT, 16>(
        [&](int t, int r) { return &Ks[(hf * 4 + (r & 3) + 8 * (r >> 2)) * KP + t * 32 + ql]; },
        [&](int t, const float (&a)[16]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], s[r], acc[t], 0, 0, 0);
        });
    if (tile + 1 < n_tiles) park(tile + 1, cur ^ 1);
    __syncthreads();
  }
  if (qi >= p.Lq) return;
  if (piece >= 0) {
    float* sl = p.q_slabs + ((static_cast<long>(piece) * gridDimBH(p) + bh) * p.Lq + qi) * DQ;
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * t + 4 * hf + 8 * g;
        if (c >= DQ) continue;
        const float sc = c < D ? p.scale : 1.0f;
        st4(sl + c, make_float4(acc[t][4 * g + 0] * sc, acc[t][4 * g + 1] * sc, acc[t][4 * g + 2] * sc, acc[t][4 * g + 3] * sc));
      }
    return;
  }
  float* dqr = p.dq + (static_cast<long>(bh) * p.Lq + qi) * D;
  float* der = E ? p.dq_extra + (static_cast<long>(bh) * p.Lq + qi) * E : nullptr;
  const bool add_res = p.residual && !(p.skip_first && qi == 0);
  const float* gres = p.dout + (static_cast<long>(b) * p.Lq + qi) * (static_cast<long>(p.H) * DV) + h * DV;
#pragma unroll
  for (int t = 0; t < NQT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * t + 4 * hf + 8 * g;
      float4 v4 = make_float4(acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
      if (c < D) {
        v4.x *= p.scale; v4.y *= p.scale; v4.z *= p.scale; v4.w *= p.scale;
        if (add_res) { const float4 r4 = ld4(gres + c); v4.x += r4.x; v4.y += r4.y; v4.z += r4.z; v4.w += r4.w; }
        st4(dqr + c, v4);
      } else if (c < DQ) {
        st4(der + (c - D), v4);
      }
    }
}


template <int D, int E, int DV, bool WRITE_DS>
__global__ __launch_bounds__(256) void attention_bwd_kv_kernel(AttnBwdArgs p) {
  constexpr int DQ = D + E;
  constexpr int HQ = DQ / 2, HV = DV / 2;
  constexpr int NKT = D / 32, NVT = DV / 32;
  constexpr int QP = DQ + 4, GP = DV + 4;
  static_assert(D % 32 == 0, "dK tiles");
  // two stages: the next query tile is parked between the two halves of this tile's MFMAs (its loads were issued at the top of
  // the iteration) -- one barrier per tile and no wait for memory behind it
  __shared__ __attribute__((aligned(16))) float Qs2[2][32 * QP];   // [q*scale | q_extra] of a query tile
  __shared__ __attribute__((aligned(16))) float Gs2[2][32 * GP];   // its dO
  __shared__ float Ls2[2][32], Ds2[2][32];                          // lse, delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int kl = lane & 31, hf = lane >> 5;
  const int ki = blockIdx.x * 128 + wave * 32 + kl;
  const int kc = ki < p.Lk ? ki : p.Lk - 1;

  float kf[HQ], vf[HV];   // this lane's halves of [k | k_extra] and v for its key
  {
    const float* kr = p.k + b * p.k_sb + h * p.k_sh + static_cast<long>(kc) * p.k_sl;
    const float* ke = E ? p.k_extra + static_cast<long>(kc) * E : nullptr;
#pragma unroll
    for (int j4 = 0; j4 < HQ / 4; ++j4) {
      const int e0 = hf * HQ + j4 * 4;
      const float* src = kr + (e0 < D ? e0 : 0);
      if constexpr (E > 0) src = e0 < D ? src : ke + (e0 - D);
      const float4 t = ld4(src);
      kf[j4 * 4 + 0] = t.x; kf[j4 * 4 + 1] = t.y; kf[j4 * 4 + 2] = t.z; kf[j4 * 4 + 3] = t.w;
    }
    const float* vr = p.v + b * p.v_sb + h * p.v_sh + static_cast<long>(kc) * p.v_sl + hf * HV;
#pragma unroll
    for (int j4 = 0; j4 < HV / 4; ++j4) {
      const float4 t = ld4(vr + j4 * 4);
      vf[j4 * 4 + 0] = t.x; vf[j4 * 4 + 1] = t.y; vf[j4 * 4 + 2] = t.z; vf[j4 * 4 + 3] = t.w;
    }
  }
  f32x16 ak[NKT], av[NVT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ak[t][r] = 0.f;
#pragma unroll
  for (int t = 0; t < NVT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) av[t][r] = 0.f;

  const float* qb = p.q + b * p.q_sb + h * p.q_sh;
  const int all_tiles = (p.Lq + 31) / 32;
  const int tile_lo = static_cast<int>(static_cast<long>(all_tiles) * blockIdx.z / p.q_splits);
  const int tile_hi = static_cast<int>(static_cast<long>(all_tiles) * (blockIdx.z + 1) / p.q_splits);
  constexpr int QF4 = 32 * DQ / 4, GF4 = 32 * DV / 4;
  constexpr int QPT = (QF4 + 255) / 256, GPT = (GF4 + 255) / 256;
  float4 qreg[QPT], greg[GPT];
  float lreg = 0.f, dreg = 0.f;
  auto fetch = [&](int tile) {
    const int q0 = tile * 32;
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DQ / 4), c4 = (idx - row * (DQ / 4)) * 4;
      const int qq = q0 + (row < 32 ? row : 31);
      const long qcl = qq < p.Lq ? qq : p.Lq - 1;
      const float* src = qb + qcl * p.q_sl + (c4 < D ? c4 : 0);
      if constexpr (E > 0) src = c4 < D ? src : p.q_extra + (static_cast<long>(bh) * p.Lq + qcl) * E + (c4 - D);
      qreg[i] = ld4(src);                                        // scaled and masked when it is parked
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DV / 4), c4 = (idx - row * (DV / 4)) * 4;
      const int qq = q0 + (row < 32 ? row : 31);
      const long qcl = qq < p.Lq ? qq : p.Lq - 1;
      greg[i] = ld4(p.dout + (static_cast<long>(b) * p.Lq + qcl) * (static_cast<long>(p.H) * DV) + h * DV + c4);
    }
    if (tid < 32) {
      const int qq = q0 + tid;
      const long qcl = qq < p.Lq ? qq : p.Lq - 1;
      lreg = p.lse[static_cast<long>(bh) * p.Lq + qcl];
      dreg = p.delta[static_cast<long>(bh) * p.Lq + qcl];
    }
  };
  auto park = [&](int tile, int st) {
    float* Qs = Qs2[st]; float* Gs = Gs2[st]; float* Ls = Ls2[st]; float* Ds = Ds2[st];
    const int q0 = tile * 32;
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DQ / 4), c4 = (idx - row * (DQ / 4)) * 4;
      float4 t = keep_or_zero(qreg[i], q0 + row < p.Lq);
      const float sc = c4 < D ? p.scale : 1.0f;
      t.x *= sc; t.y *= sc; t.z *= sc; t.w *= sc;
      if (idx < QF4) st4(&Qs[row * QP + c4], t);
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (DV / 4), c4 = (idx - row * (DV / 4)) * 4;
      if (idx < GF4) st4(&Gs[row * GP + c4], keep_or_zero(greg[i], q0 + row < p.Lq));
    }
    if (tid < 32) {
      const bool in = q0 + tid < p.Lq;
      Ls[tid] = in ? lreg : 3.0e38f;                             // exp(s - huge) = 0 for rows past the end
      Ds[tid] = in ? dreg : 0.f;
    }
  };
  fetch(tile_lo);
  park(tile_lo, 0);
  __syncthreads();
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int cur = (tile - tile_lo) & 1;
    const float* Qs = Qs2[cur]; const float* Gs = Gs2[cur]; const float* Ls = Ls2[cur]; const float* Ds = Ds2[cur];
    if (tile + 1 < tile_hi) fetch(tile + 1);
    // S = Q' K'^T (rows = queries, column = this lane's key):  A = Q'[query = lane & 31][hf*HQ + j], B = kf[j]
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    const float* qrow = &Qs[kl * QP + hf * HQ];
    mfma_groups_f32<HQ / 4>([&](int j4) { return qrow + j4 * 4; },
                            [&](int j4, float4 a) { DS_MFMA4(s, a, kf[j4 * 4 + 0], kf[j4 * 4 + 1], kf[j4 * 4 + 2], kf[j4 * 4 + 3]); });
    const float* grow = &Gs[kl * GP + hf * HV];                  // dP = dO V^T
    mfma_groups_f32<HV / 4>([&](int j4) { return grow + j4 * 4; },
                            [&](int j4, float4 a) { DS_MFMA4(dp, a, vf[j4 * 4 + 0], vf[j4 * 4 + 1], vf[j4 * 4 + 2], vf[j4 * 4 + 3]); });
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow_i = hf * 4 + (r & 3) + 8 * (r >> 2);
      const float pr = __expf(s[r] - Ls[qrow_i]);
      s[r] = pr;                                                 // P
      dp[r] = pr * (dp[r] - Ds[qrow_i]);                         // dS
    }
    if constexpr (WRITE_DS) {                                    // dS[query][this lane's key]: 32 lanes = 128 contiguous bytes per row
      if (ki < p.Lkp) {
        const int qbase = tile * 32 + hf * 4;
        float* dst = p.ds + (static_cast<long>(bh) * p.Lq + qbase) * p.Lkp + ki;
        const bool key_ok = ki < p.Lk;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dq_row = (r & 3) + 8 * (r >> 2);
          if (qbase + dq_row < p.Lq) dst[static_cast<long>(dq_row) * p.Lkp] = key_ok ? dp[r] : 0.f;
        }
      }
    }
    if (tile + 1 < tile_hi) park(tile + 1, cur ^ 1);             // nobody reads that stage before the barrier below
    // dV^T += dO^T P and dK^T += (scale q)^T dS : A = dO / Q'[query row of (r, hf)][32 t + (lane & 31)], scalars read ahead
    mfma_groups_scalar_f32<2 * (NVT + NKT), 8>(                // group = (output tile, half of the 16 query rows)
        [&](int gg, int j) {
          const int g = gg >> 1, r = 8 * (gg & 1) + j;
          const int qrow_i = hf * 4 + (r & 3) + 8 * (r >> 2);
          return g < NVT ? &Gs[qrow_i * GP + g * 32 + kl] : &Qs[qrow_i * QP + (g - NVT) * 32 + kl];
        },
        [&](int gg, const float (&a)[8]) {
          const int g = gg >> 1, r0 = 8 * (gg & 1);
          if (g < NVT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) av[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], s[r0 + j], av[g], 0, 0, 0);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) ak[g - NVT] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], dp[r0 + j], ak[g - NVT], 0, 0, 0);
          }
        });
    __syncthreads();
  }
  if (ki >= p.Lk) return;
  float* dkr = p.dk + (static_cast<long>(bh) * p.Lk + ki) * D;
  float* dvr = p.dv + (static_cast<long>(bh) * p.Lk + ki) * DV;
  if (p.q_splits > 1) {     // partial sums of this query range: [split][dk (all rows) | dv (all rows)]
    const long n_k = static_cast<long>(gridDim.y) * p.Lk;
    float* base = p.kv_part + static_cast<long>(blockIdx.z) * n_k * (D + DV);
    dkr = base + (static_cast<long>(bh) * p.Lk + ki) * D;
    dvr = base + n_k * D + (static_cast<long>(bh) * p.Lk + ki) * DV;
  }
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      st4(dkr + 32 * t + 4 * hf + 8 * g, make_float4(ak[t][4 * g + 0], ak[t][4 * g + 1], ak[t][4 * g + 2], ak[t][4 * g + 3]));
#pragma unroll
  for (int t = 0; t < NVT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      st4(dvr + 32 * t + 4 * hf + 8 * g, make_float4(av[t][4 * g + 0], av[t][4 * g + 1], av[t][4 * g + 2], av[t][4 * g + 3]));
}

// dk | dv = sum over the query splits, in split order (deterministic)
__global__ __launch_bounds__(256) void attention_bwd_kv_sum_kernel(const float* __restrict__ part, float* __restrict__ dk,
                                                                   float* __restrict__ dv, long n_dk, long n_dv, int splits) {
  const long n = n_dk + n_dv;
  for (long i = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 4; i < n; i += static_cast<long>(gridDim.x) * 1024) {
    float4 a = ld4(part + i);
    for (int s = 1; s < splits; ++s) {
      const float4 b = ld4(part + s * n + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    st4(i < n_dk ? dk + i : dv + (i - n_dk), a);
  }
}

template <int D, int E, int DV>
static int launch_attention_bwd_kv(AttnBwdArgs& a, int B, hipStream_t s) {
  const dim3 grid((a.Lk + 127) / 128, B * a.H, a.q_splits);
  if (a.ds) hipLaunchKernelGGL((attention_bwd_kv_kernel<D, E, DV, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attention_bwd_kv_kernel<D, E, DV, false>), grid, dim3(256), 0, s, a);
  int rc = check_launch("attention_general_bwd(dk, dv)");
  if (rc || a.q_splits == 1) return rc;
  const long n_dk = static_cast<long>(B) * a.H * a.Lk * D, n_dv = static_cast<long>(B) * a.H * a.Lk * DV;
  long g = ((n_dk + n_dv) / 4 + 255) / 256;
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(attention_bwd_kv_sum_kernel, dim3(static_cast<int>(g)), dim3(256), 0, s, a.kv_part, a.dk, a.dv, n_dk, n_dv,
                     a.q_splits);
  return check_launch("attention_general_bwd(sum)");
}

template <int D, int E, int DV>
static int launch_attention_bwd(AttnBwdArgs& a, int B, hipStream_t s) {
  const long rows = static_cast<long>(B) * a.H * a.Lq;
  hipLaunchKernelGGL((attention_bwd_delta_kernel<DV>), dim3(static_cast<unsigned>((rows + 31) / 32)), dim3(256), 0, s, a, rows);
  int rc = check_launch("attention_general_bwd(delta)");
  if (rc) return rc;
  if (a.ds) {                // dS form: dk / dv first (it writes dS), then dq = dS K'
    rc = launch_attention_bwd_kv<D, E, DV>(a, B, s);
    if (rc) return rc;
  }
  {
    const int total = a.qb_x * a.qb_bh;
    const int grid = a.qb_full + (total - a.qb_full) * a.qb_split;
    if (a.ds) hipLaunchKernelGGL((attention_bwd_q_ds_kernel<D, E, DV>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attention_bwd_q_kernel<D, E, DV>), dim3(grid), dim3(256), 0, s, a);
    rc = check_launch("attention_general_bwd(dq)");
    if (rc) return rc;
    if (a.qb_split > 1) {
      const int n_tail = total - a.qb_full;
      const long items = static_cast<long>(n_tail) * 128 * ((D + E) / 4);
      long g = (items + 255) / 256;
      g = g > 2048 ? 2048 : g;
      hipLaunchKernelGGL((attention_bwd_q_tail_kernel<D, E, DV>), dim3(static_cast<unsigned>(g)), dim3(256), 0, s, a, n_tail);
      rc = check_launch("attention_general_bwd(dq tail)");
      if (rc) return rc;
    }
  }
  return a.ds ? 0 : launch_attention_bwd_kv<D, E, DV>(a, B, s);
}

}  // namespace diffsal

using namespace diffsal;

// Tail mode plan of the forward and the dq kernel: pieces per tail block (1 = off) and the first tail block.  On when the
// last round of workgroups (one per CU) fills at most half of the chip and there are enough key tiles to cut; a launch
// of less than one round is cut whole (every block a tail block) when that shortens the longest CU's share by >= 15 %.
static void attention_tail_plan(int BH, int Lq, int Lk, int* qb_x, int* qb_full, int* qb_split) {
  const int gx = (Lq + 127) / 128, total = gx * BH, n_tiles = (Lk + 31) / 32;
  *qb_x = gx; *qb_full = total; *qb_split = 1;
  if (total < 256) {
    int best = 1;
    float best_t = 1.0f;
    for (int sp = 2; sp <= 4; ++sp) {
      if (n_tiles / sp < 4) break;
      const float t = static_cast<float>((total * sp + 255) / 256) / sp;
      if (t < best_t - 1e-6f) { best_t = t; best = sp; }
    }
    if (best > 1 && best_t <= 0.85f) { *qb_full = 0; *qb_split = best; }
    return;
  }
  const int full = (total / 256) * 256, rest = total - full;
  if (rest == 0 || rest > 128) return;
  int sp = 256 / rest;
  while (sp > 1 && n_tiles / sp < 4) --sp;
  if (sp < 2) return;
  *qb_full = full; *qb_split = sp;
}

// floats of scratch for the forward's tail mode (0: not used for this shape): the pieces' outputs and log-sum-exps
extern "C" size_t diffsal_attention_general_tail_floats(int B, int H, int Lq, int Lk, int DV) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  int gx, full, sp;
  attention_tail_plan(B * H, Lq, Lk, &gx, &full, &sp);
  return sp > 1 ? static_cast<size_t>(sp) * B * H * Lq * (DV + 1) : 0;
}

extern "C" int diffsal_attention_general(const float* q, const float* q_extra, const float* k, const float* k_extra, const int* k_slots,
                                         const float* v, const float* residual, float* out, float* lse, int B, int H, int Lq, int Lk,
                                         int D, int E, int DV, const long* q_strides, const long* k_strides,
                                         const long* v_strides, const long* r_strides, float scale, int skip_first,
                                         float* tail_ws, size_t tail_ws_floats, diffsal_stream_t stream) {
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
  a.q = q; a.q_extra = q_extra; a.k = k; a.k_extra = k_extra; a.v = v; a.residual = residual; a.out = out; a.lse = lse;
  a.k_slots = (k_slots && E > 0 && aligned16(k_slots) && tune(TUNE_NO_ATTN_SLOTS) <= 0) ? k_slots : nullptr;
  a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sl = q_strides[2];
  a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sl = k_strides[2];
  a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sl = v_strides[2];
  a.r_sb = residual ? r_strides[0] : 0; a.r_sh = residual ? r_strides[1] : 0; a.r_sl = residual ? r_strides[2] : 0;
  a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.skip_first = skip_first;
  a.qb_bh = B * H;
  attention_tail_plan(B * H, Lq, Lk, &a.qb_x, &a.qb_full, &a.qb_split);
  // no scratch, or less than this plan's pieces need (diffsal_attention_general_tail_floats): every block whole
  if (!tail_ws || !aligned16(tail_ws) || tail_ws_floats < diffsal_attention_general_tail_floats(B, H, Lq, Lk, DV)) {
    a.qb_full = a.qb_x * a.qb_bh;
    a.qb_split = 1;
    tail_ws = nullptr;
  }
  a.o_slabs = tail_ws;
  a.l_slabs = tail_ws ? tail_ws + static_cast<long>(a.qb_split) * B * H * Lq * DV : nullptr;
  const int total = a.qb_x * a.qb_bh, n_tail = total - a.qb_full;
  const dim3 grid(static_cast<unsigned>(a.qb_full + n_tail * a.qb_split));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (D == 96 && E == 48 && DV == 96 && a.k_slots) hipLaunchKernelGGL((attention_fwd_kernel<96, 48, 96, true>), grid, dim3(256), 0, s, a);
  else if (D == 96 && E == 32 && DV == 96 && a.k_slots) hipLaunchKernelGGL((attention_fwd_kernel<96, 32, 96, true>), grid, dim3(256), 0, s, a);
  else if (D == 96 && E == 48 && DV == 96) hipLaunchKernelGGL((attention_fwd_kernel<96, 48, 96>), grid, dim3(256), 0, s, a);
  else if (D == 96 && E == 32 && DV == 96) hipLaunchKernelGGL((attention_fwd_kernel<96, 32, 96>), grid, dim3(256), 0, s, a);
  else if (D == 96 && E == 0 && DV == 96) hipLaunchKernelGGL((attention_fwd_kernel<96, 0, 96>), grid, dim3(256), 0, s, a);
  else if (D == 64 && E == 0 && DV == 64) hipLaunchKernelGGL((attention_fwd_kernel<64, 0, 64>), grid, dim3(256), 0, s, a);
  else if (D == 32 && E == 0 && DV == 32) hipLaunchKernelGGL((attention_fwd_kernel<32, 0, 32>), grid, dim3(256), 0, s, a);
  else {
    set_error("attention_general: (D, E, DV) = (%d, %d, %d) is not built: (96,48,96), (96,32,96), (96,0,96), (64,0,64), (32,0,32)", D, E, DV);
    return DIFFSAL_E_SHAPE;
  }
  int rc = check_launch("attention_general");
  if (rc || n_tail == 0) return rc;
  long g = (static_cast<long>(n_tail) * 128 * (DV / 4) + 255) / 256;
  g = g > 2048 ? 2048 : g;
  if (DV == 96) hipLaunchKernelGGL(attention_fwd_tail_kernel<96>, dim3(static_cast<unsigned>(g)), dim3(256), 0, s, a, n_tail);
  else if (DV == 64) hipLaunchKernelGGL(attention_fwd_tail_kernel<64>, dim3(static_cast<unsigned>(g)), dim3(256), 0, s, a, n_tail);
  else hipLaunchKernelGGL(attention_fwd_tail_kernel<32>, dim3(static_cast<unsigned>(g)), dim3(256), 0, s, a, n_tail);
  return check_launch("attention_general(tail)");
}

// Query splits of the dk / dv kernel: enough workgroups to fill the chip when there are few keys (Lk = 673 at B*H = 4 is
// 24 workgroups), at most one split per 8 query tiles.
extern "C" int diffsal_attention_general_bwd_splits(int B, int H, int Lq, int Lk) {
  const long wgs = static_cast<long>((Lk + 127) / 128) * B * H;
  long s = (768 + wgs - 1) / wgs;
  const long max_s = ((Lq + 31) / 32 + 7) / 8;
  s = s > max_s ? max_s : s;
  s = s > 64 ? 64 : s;
  return static_cast<int>(s < 1 ? 1 : s);
}

// floats of scratch for the dq kernel's tail mode (0: not used for this shape)
extern "C" size_t diffsal_attention_general_bwd_qtail_floats(int B, int H, int Lq, int Lk, int D, int E) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  int gx, full, sp;
  attention_tail_plan(B * H, Lq, Lk, &gx, &full, &sp);
  return sp > 1 ? static_cast<size_t>(sp) * B * H * Lq * (D + E) : 0;
}

// floats of the dS buffer of the backward's dS form: [B*H][Lq][Lk rounded up to 32]
extern "C" size_t diffsal_attention_general_bwd_ds_floats(int B, int H, int Lq, int Lk) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return 0;
  return static_cast<size_t>(B) * H * Lq * ((Lk + 31) / 32 * 32);
}

extern "C" int diffsal_attention_general_bwd(const float* q, const float* q_extra, const float* k, const float* k_extra,
                                             const float* v, const float* residual, const float* out, const float* lse,
                                             const float* dout, float* delta_ws, float* kv_part_ws, float* q_tail_ws,
                                             size_t q_tail_ws_floats, float* ds_ws, size_t ds_ws_floats, float* dq, float* dq_extra,
                                             float* dk, float* dv, int B, int H, int Lq, int Lk, int D, int E, int DV,
                                             const long* q_strides, const long* k_strides, const long* v_strides,
                                             const long* r_strides, float scale, int skip_first, diffsal_stream_t stream) {
  DS_REQUIRE(q && k && v && out && lse && dout && delta_ws && dq && dk && dv && q_strides && k_strides && v_strides,
             DIFFSAL_E_ARG, "attention_general_bwd: null argument");
  DS_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && static_cast<long>(B) * H < 65536, DIFFSAL_E_SHAPE,
             "attention_general_bwd: bad shape");
  DS_REQUIRE((E == 0) == (q_extra == nullptr) && (E == 0) == (k_extra == nullptr) && (E == 0) == (dq_extra == nullptr),
             DIFFSAL_E_ARG, "attention_general_bwd: q_extra / k_extra / dq_extra must be given exactly when E > 0");
  DS_REQUIRE(!residual || r_strides, DIFFSAL_E_ARG, "attention_general_bwd: residual needs its strides");
  AttnBwdArgs a;
  a.q = q; a.q_extra = q_extra; a.k = k; a.k_extra = k_extra; a.v = v; a.dout = dout; a.out = out; a.residual = residual;
  a.lse = lse; a.delta = delta_ws; a.dq = dq; a.dq_extra = dq_extra; a.dk = dk; a.dv = dv;
  a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sl = q_strides[2];
  a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sl = k_strides[2];
  a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sl = v_strides[2];
  a.r_sb = residual ? r_strides[0] : 0; a.r_sh = residual ? r_strides[1] : 0; a.r_sl = residual ? r_strides[2] : 0;
  a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.skip_first = skip_first;
  a.q_splits = diffsal_attention_general_bwd_splits(B, H, Lq, Lk);
  a.kv_part = kv_part_ws;
  a.qb_bh = B * H;
  attention_tail_plan(B * H, Lq, Lk, &a.qb_x, &a.qb_full, &a.qb_split);
  // no scratch, or less than this plan's pieces need (diffsal_attention_general_bwd_qtail_floats): every block whole
  if (!q_tail_ws || !aligned16(q_tail_ws) || q_tail_ws_floats < diffsal_attention_general_bwd_qtail_floats(B, H, Lq, Lk, D, E)) {
    a.qb_full = a.qb_x * a.qb_bh;
    a.qb_split = 1;
    q_tail_ws = nullptr;
  }
  a.q_slabs = q_tail_ws;
  // dS form when the caller brings the buffer (NULL or a smaller one: the recomputing dq kernel)
  a.ds = ds_ws && aligned16(ds_ws) && ds_ws_floats >= diffsal_attention_general_bwd_ds_floats(B, H, Lq, Lk) && tune(TUNE_NO_ATTN_BWD_DS) <= 0
             ? ds_ws : nullptr;
  a.Lkp = (Lk + 31) / 32 * 32;
  DS_REQUIRE(a.q_splits == 1 || (kv_part_ws && aligned16(kv_part_ws)), DIFFSAL_E_ARG,
             "attention_general_bwd: %d query splits need kv_part_ws of splits * B*H*Lk*(D+DV) floats", a.q_splits);
  DS_REQUIRE((static_cast<long>(B) * H * Lk * D) % 4 == 0, DIFFSAL_E_SHAPE, "attention_general_bwd: dk size");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (D == 96 && E == 48 && DV == 96) return launch_attention_bwd<96, 48, 96>(a, B, s);
  if (D == 96 && E == 32 && DV == 96) return launch_attention_bwd<96, 32, 96>(a, B, s);
  if (D == 96 && E == 0 && DV == 96) return launch_attention_bwd<96, 0, 96>(a, B, s);
  if (D == 64 && E == 0 && DV == 64) return launch_attention_bwd<64, 0, 64>(a, B, s);
  if (D == 32 && E == 0 && DV == 32) return launch_attention_bwd<32, 0, 32>(a, B, s);
  set_error("attention_general_bwd: (D, E, DV) = (%d, %d, %d) is not built", D, E, DV);
  return DIFFSAL_E_SHAPE;
}
