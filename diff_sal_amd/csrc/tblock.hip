// Fused first half of a TransformerBlock of the saliency decoder (the C = 96 stage), one persistent launch instead of five
// (LayerNorm, depthwise-q + LayerNorm, proj_q GEMM, attention, proj GEMM) and their eight round trips of the token tensor:
//
//     xn = LayerNorm_1(x)                                         R/models/saliency_decoder/transformer.py:150
//     q  = proj_q( LayerNorm_q( dwconv3x3(xn) ) )                 attention.py:36-47 (centre temporal slice, Q8), :87-90, :97
//     o  = softmax( q k^T * C^-1/2 ) v        per head            attention.py:99-108 (scale uses the full C, Q6)
//     x1 = x + proj(o)                        (fp32 storage)      attention.py:110, transformer.py:151-152
//
// k, v are the block's PROJECTED keys / values ([N, Lk <= 32, C]; Lk = 18 in every reference configuration), produced by the
// pooled branch (diffsal_qkv_prep + diffsal_linear_pair).  On 16-bit storage the kernel stops at o: the existing block16 kernel
// owns proj + residual + LayerNorm_2 + MLP.
//
// Work decomposition.  A workgroup (4 wavefronts) owns an 8 x 16 pixel tile of one frame; a wavefront owns 2 x 16 pixels, one
// pixel (token) per lane pair (lane, lane ^ 32).  Phase A: the tile plus a one-pixel halo is loaded, normalised (LayerNorm_1,
// fp32 statistics) and parked in LDS in the storage type -- what the unfused path wrote to HBM.  Phase B: every lane forms the
// depthwise 3x3 response and LayerNorm_q of ITS token for the 48 channels its lane half owns (channel groups 8g + 4h .. + 3),
// which is exactly the B-operand layout of the following MFMA, so q_in never touches LDS.  Phase C: the frame's K and V^T
// replace the (now dead) halo tile in LDS.  Phase D works TRANSPOSED like mlp_block / block16 (weights are the A operand, the
// C/D registers of one product are the B operand of the next): Q^T = Wq q_in^T, S^T = K Q^T per head, softmax down the
// registers (+ one lane ^ 32 exchange), O^T = V^T P^T, X1^T = Wp O^T.  fp32 storage multiplies on v_mfma_f32_32x32x2_f32 (exact
// fp32), 16-bit storage on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation; one rounding per value the unfused path
// stored (xn, q_in, q, o) plus one on the probabilities (the 16-bit B operand).
#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mma16;
template <> struct Mma16<__bf16> {
  typedef bf16x8 vec;
  static __device__ __forceinline__ f32x16 mma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma16<_Float16> {
  typedef f16x8 vec;
  static __device__ __forceinline__ f32x16 mma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma16<float> { typedef f32x8 vec; };   // placeholder: the fp32 instantiation never touches the 16-bit branch

template <typename T>
struct FrontArgs {
  const T* x;            // [N, H, W, C]
  const T* k;            // [N, Lk, C] projected keys
  const T* v;            // [N, Lk, C] projected values
  const float* g1; const float* b1;      // LayerNorm_1
  const float* w9;       // [9][C] depthwise taps, tap = 3 * ky + kx
  const float* gq; const float* bq;      // LayerNorm_q
  const T* wq; const float* bias_q;      // [C][C]
  const T* wp; const float* bias_p;      // [C][C] (fp32 storage only)
  T* out;                // fp32 storage: x1; 16-bit storage: o
  int N, H, W, Lk;
  float eps1, epsq, scale;
  int tiles_y, tiles_x;
#ifdef DIFFSAL_DEV_STAMPS
  unsigned long long* stamps;   // development build only: 8 wall-clock stamps per (workgroup, tile) of the first 8 tiles
#endif
};

#ifdef DIFFSAL_DEV_STAMPS
static unsigned long long* g_front_stamps = nullptr;
static size_t g_front_stamp_bytes = 0;
#endif

// LDS column permutation of a 16-bit A operand inside a 32-column block (see block16.hip): position q = 16 j + 8 h + e holds
// column 16 j + 4 h + (e < 4 ? e : e + 4)
__device__ __forceinline__ int perm32f(int q) {
  const int j = q >> 4, h = (q >> 3) & 1, e = q & 7;
  return 16 * j + 4 * h + (e < 4 ? e : e + 4);
}
__device__ __forceinline__ int iperm32f(int c) {   // inverse: column c of the block -> LDS position
  const int j = c >> 4, r = c & 15, h = (r >> 2) & 1, e = (r & 3) + ((r >> 3) << 2);
  return 16 * j + 8 * h + e;
}

template <typename T> __device__ __forceinline__ T from_f(float v) { return static_cast<T>(v); }

// a 4-element piece as it travels through registers: 16 bytes of fp32, 8 bytes of 16-bit storage
template <typename T> struct RawPiece { typedef uint2 type; };
template <> struct RawPiece<float> { typedef float4 type; };
__device__ __forceinline__ float4 raw_to_f4_impl(float4 r, const float*) { return r; }
__device__ __forceinline__ float4 raw_to_f4_impl(uint2 r, const bf16_t*) {
  return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xFFFF0000u),
                     __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xFFFF0000u));
}
__device__ __forceinline__ float4 raw_to_f4_impl(uint2 r, const f16_t*) {
  const f16x4_t h = __builtin_bit_cast(f16x4_t, r);
  const f32x4_t f = __builtin_convertvector(h, f32x4_t);
  return make_float4(f.x, f.y, f.z, f.w);
}

constexpr int FT_TH = 8, FT_TW = 16;                 // tile of a workgroup (pixels)
constexpr int FT_HW = FT_TW + 2;                     // halo tile width
constexpr int FT_TOK = (FT_TH + 2) * FT_HW;          // 180 halo-tile tokens

template <typename T, int C> struct FrontLds {
  static constexpr bool F32 = sizeof(T) == 4;
  static constexpr int PW = F32 ? C + 4 : C + 8;           // weight / K row pitch (elements): odd multiple of 16 B
  static constexpr int PX = C + (F32 ? 4 : 4);             // halo-tile token pitch: odd multiple of the 4-element piece
  static constexpr int PV = F32 ? 36 : 40;                 // V^T row pitch (32 key columns + pad)
  static constexpr size_t w_bytes = static_cast<size_t>(F32 ? 2 : 1) * C * PW * sizeof(T);
  static constexpr size_t vec_bytes = static_cast<size_t>(9 * C + 4 * C) * sizeof(float);   // w9 | gq | bq | bias_q | bias_p
  static constexpr size_t tile_bytes = static_cast<size_t>(FT_TOK) * PX * sizeof(T);
  static constexpr size_t kv_bytes = (static_cast<size_t>(32) * PW + static_cast<size_t>(C) * PV) * sizeof(T);
  static constexpr size_t region_bytes = ((tile_bytes > kv_bytes ? tile_bytes : kv_bytes) + 15) / 16 * 16;
  static constexpr size_t total = w_bytes + vec_bytes + region_bytes;
};

template <typename T, int C>
__global__ __launch_bounds__(256, ((sizeof(T) == 4 || C > 96) ? 1 : 2)) void block_front_kernel(FrontArgs<T> p) {
  typedef FrontLds<T, C> L;
  constexpr bool F32 = L::F32;
  // fp32 storage runs one workgroup per CU (LDS) with 512 registers per lane: the next tile's halo pieces are fetched a tile
  // ahead.  16-bit storage fits two workgroups per CU, which cover each other's load phases; no look-ahead (256 registers)
  // C = 192 on 16-bit storage: Wq + the halo tile fill the LDS: one workgroup per CU as well, same look-ahead
  constexpr bool AHEAD = F32 || C > 96;
  typedef typename Mma16<T>::vec vec;
  typedef typename RawPiece<T>::type raw_t;
  auto raw_to_f4 = [](raw_t r) { return raw_to_f4_impl(r, static_cast<const T*>(nullptr)); };
  constexpr int NU = C / 32;                 // 32-channel row tiles
  constexpr int NG = C / 8;                  // channel groups (4 channels) per lane
  constexpr int HEADS = 2, D = C / HEADS;
  constexpr int NP = C / 4;                  // 4-element pieces per token
  constexpr int G = NP / 3;                  // phase A: lanes per token (3 pieces each)
  static_assert(C % 32 == 0 && NP % 3 == 0 && (G == 8 || G == 16), "C = 96 or 192");
  constexpr int PW = L::PW, PX = L::PX, PV = L::PV;

  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  T* Wqs = reinterpret_cast<T*>(smraw);                         // [C][PW]
  T* Wps = Wqs + C * PW;                                        // [C][PW]  (fp32 storage only)
  float* vecs = reinterpret_cast<float*>(smraw + L::w_bytes);   // w9 [9][C] | gq | bq | bias_q | bias_p
  T* region = reinterpret_cast<T*>(smraw + L::w_bytes + L::vec_bytes);
  T* xns = region;                                              // phase A/B: [FT_TOK][PX]
  T* Ks = region;                                               // phase C/D: [32][PW]
  T* Vts = region + 32 * PW;                                    //            [C][PV]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ml = lane & 31, hf = lane >> 5;

  // ---- weights and vectors, once per workgroup
  for (int i = tid; i < C * C; i += 256) {
    const int r = i / C, q = i - r * C;
    if constexpr (F32) {
      Wqs[r * PW + q] = p.wq[i];
      Wps[r * PW + q] = p.wp[i];
    } else {
      Wqs[r * PW + q] = p.wq[r * C + (q & ~31) + perm32f(q & 31)];
    }
  }
  for (int i = tid; i < 9 * C; i += 256) vecs[i] = p.w9[i];
  for (int i = tid; i < C; i += 256) {
    vecs[9 * C + i] = p.gq[i]; vecs[10 * C + i] = p.bq[i]; vecs[11 * C + i] = p.bias_q[i];
    vecs[12 * C + i] = F32 ? p.bias_p[i] : 0.f;
  }
  // phase A: this thread's pieces of LayerNorm_1's affine
  const int a_slot = tid / G, a_sub = tid - a_slot * G;
  float4 ga[3], ba[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { ga[i] = ld4(p.g1 + 4 * (a_sub + G * i)); ba[i] = ld4(p.b1 + 4 * (a_sub + G * i)); }
  __syncthreads();

  const int n_tiles = p.N * p.tiles_y * p.tiles_x;
  const int trow = 2 * wave + (ml >> 4), tcol = ml & 15;             // this lane's pixel inside the tile
  const int centre = (trow + 1) * FT_HW + (tcol + 1);
  const T* wqf = Wqs + ml * PW + (F32 ? 4 : 8) * hf;
  const T* wpf = Wps + ml * PW + 4 * hf;
  const T* kf = Ks + ml * PW + (F32 ? 4 : 8) * hf;
  const T* vtf = Vts + ml * PV + (F32 ? 4 : 8) * hf;
  auto chan = [&](int u, int r) { return 32 * u + 4 * hf + (r & 3) + 8 * (r >> 2); };

  // phase A's global loads, issued one tile ahead: pass ps covers halo tokens ps * TPP + a_slot; returns the "inside the frame" bits
  constexpr int TPP = 256 / G;                                       // tokens per pass
  constexpr int NPASS = (FT_TOK + TPP - 1) / TPP;
  // NAH passes are fetched a tile ahead and wait in registers through phase D; the rest are fetched in phase A, two in flight
  // (slots NAH, NAH + 1).  fp32: all of them ahead.  C = 192 on 16-bit storage: half -- with all twelve passes (72 registers)
  // beside the six Q^T accumulators the kernel needed 512 registers and spilled 12-15 of them to scratch
  constexpr int NAH = !AHEAD ? 0 : (F32 ? NPASS : NPASS / 2);
  // round 6: the passes that are not fetched ahead are ALL requested at the top of phase A (one slot each: 6 registers per pass on
  // 16-bit storage, free there -- the accumulators of phase D do not exist yet).  With two in flight phase A was six (twelve at
  // C = 192: six) exposed memory latencies in a row, 5.6 of a tile's 16 us
  constexpr int NFL = NAH < NPASS ? NPASS - NAH : 0;
  raw_t pre[NAH + NFL][3];
  unsigned inside_mask = 0;
  auto fetch_pass = [&](int tile_, int ps, raw_t (&dst)[3]) -> bool {
    const int tx_ = tile_ % p.tiles_x, t2_ = tile_ / p.tiles_x;
    const int ty_ = t2_ % p.tiles_y, n_ = t2_ / p.tiles_y;
    const T* ximg_ = p.x + static_cast<long>(n_) * p.H * p.W * C;
    const int tok = ps * TPP + a_slot;
    const int hy = tok / FT_HW, hx = tok - hy * FT_HW;
    const int gy = ty_ * FT_TH + hy - 1, gx = tx_ * FT_TW + hx - 1;
    const bool inside = tok < FT_TOK && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const T* src = ximg_ + (static_cast<long>(inside ? gy : 0) * p.W + (inside ? gx : 0)) * C;
#pragma unroll
    for (int i = 0; i < 3; ++i) dst[i] = *reinterpret_cast<const raw_t*>(src + 4 * (a_sub + G * i));
    return inside;
  };
  auto fetch_tile = [&](int tile_) -> unsigned {
    unsigned mask = 0;
#pragma unroll
    for (int ps = 0; ps < NAH; ++ps) mask |= (fetch_pass(tile_, ps, pre[ps]) ? 1u : 0u) << ps;
    return mask;
  };
  // XCD-aware tile walk: workgroups are dealt round-robin over the eight XCDs, so virtual tile v = 8 s + xcd is tile xcd * (n_tiles / 8)
  // + s -- an XCD owns a contiguous run of tiles and the halo rows two neighbouring tiles share are fetched into ONE L2 (a bijection
  // when both counts are multiples of 8, else the identity)
  const bool xcd_walk = (n_tiles & 7) == 0 && (gridDim.x & 7) == 0;
  auto tile_of = [&](int v) { return xcd_walk ? (v & 7) * (n_tiles >> 3) + (v >> 3) : v; };
  if (AHEAD && static_cast<int>(blockIdx.x) < n_tiles) inside_mask = fetch_tile(tile_of(blockIdx.x));

#ifdef DIFFSAL_DEV_STAMPS
  int stamp_it = 0;
  auto stamp = [&](int k) {
    if (p.stamps && tid == 0 && stamp_it < 8) p.stamps[(static_cast<long>(blockIdx.x) * 8 + stamp_it) * 8 + k] = wall_clock64();
  };
#else
  auto stamp = [](int) {};
#endif
  for (int vt = blockIdx.x; vt < n_tiles; vt += gridDim.x) {
    const int tile = tile_of(vt);
    stamp(0);
    const int tx = tile % p.tiles_x, t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
    const int y0 = ty * FT_TH, x0 = tx * FT_TW;
    if constexpr (NAH < NPASS) {      // the passes that were not fetched ahead (their mask bits replace stale ones)
      inside_mask &= (1u << NAH) - 1u;
#pragma unroll
      for (int ps = NAH; ps < NPASS; ++ps) inside_mask |= (fetch_pass(tile, ps, pre[ps]) ? 1u : 0u) << ps;
    }

    // ---------------- phase A: halo tile (fetched while the previous tile was in phase D) -> LayerNorm_1 -> LDS; zeros outside
    // the frame are the convolution's padding
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int tok = ps * TPP + a_slot;
      const bool inside = (inside_mask >> ps) & 1u;
      float4 vv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) vv[i] = raw_to_f4(pre[ps][i]);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) s += (vv[i].x + vv[i].y) + (vv[i].z + vv[i].w);
      s = group_sum<G>(s);
      const float mean = s * (1.0f / C);
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float a = vv[i].x - mean, b = vv[i].y - mean, c = vv[i].z - mean, d = vv[i].w - mean;
        qq += (a * a + b * b) + (c * c + d * d);
      }
      qq = group_sum<G>(qq);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + p.eps1);
      if (tok < FT_TOK) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          float4 o;
          o.x = inside ? (vv[i].x - mean) * rstd * ga[i].x + ba[i].x : 0.f;
          o.y = inside ? (vv[i].y - mean) * rstd * ga[i].y + ba[i].y : 0.f;
          o.z = inside ? (vv[i].z - mean) * rstd * ga[i].z + ba[i].z : 0.f;
          o.w = inside ? (vv[i].w - mean) * rstd * ga[i].w + ba[i].w : 0.f;
          st4(xns + tok * PX + 4 * (a_sub + G * i), o);
        }
      }
    }
    __syncthreads();
    stamp(1);
    // this frame's K / V pieces: in flight during phase B, parked in LDS in phase C
    constexpr int NKP = 32 * NP / 256;                                // K (and V) pieces per thread: 32 key rows of NP pieces
    raw_t kraw[NKP], vraw[NKP];
    auto fetch_kv = [&]() {
      const T* ksrc = p.k + static_cast<long>(n) * p.Lk * C;
      const T* vsrc = p.v + static_cast<long>(n) * p.Lk * C;
#pragma unroll
      for (int i = 0; i < NKP; ++i) {
        const int pc = tid + 256 * i;                                 // piece: key j = pc / NP, channels 4 (pc % NP) ..
        const int j = pc / NP;
        const int off = (j < p.Lk ? j : 0) * C + 4 * (pc - j * NP);
        kraw[i] = *reinterpret_cast<const raw_t*>(ksrc + off);
        vraw[i] = *reinterpret_cast<const raw_t*>(vsrc + off);
      }
    };
    fetch_kv();

    // ---------------- phase B: depthwise 3x3 + LayerNorm_q of this lane's token, channel groups 8 g + 4 hf
    float4 qin[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) qin[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      // software-pipelined by hand: the LDS reads of half a tap (NG / 2 activation pieces + NG / 2 weight pieces) are issued
      // before the FMAs of the previous half -- left to itself the compiler waits for every pair of reads
      constexpr int HG = 6, SPT = NG / HG, NSTEP = 9 * SPT;          // six channel groups per step, SPT steps per tap
      float4 xa[2][HG], wa[2][HG];
      auto issue = [&](int step, float4 (&xd)[HG], float4 (&wd)[HG]) {
        const int tap = step / SPT, g0 = (step % SPT) * HG;
        const int off = (tap / 3 - 1) * FT_HW + (tap % 3 - 1);
        const T* xs = xns + (centre + off) * PX + 4 * hf + 8 * g0;
        const float* ws = vecs + tap * C + 4 * hf + 8 * g0;
#pragma unroll
        for (int g = 0; g < HG; ++g) { xd[g] = ld4(xs + 8 * g); wd[g] = ld4(ws + 8 * g); }
      };
      issue(0, xa[0], wa[0]);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        if (step + 1 < NSTEP) issue(step + 1, xa[(step + 1) & 1], wa[(step + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int g0 = (step % SPT) * HG;
#pragma unroll
        for (int g = 0; g < HG; ++g) {
          const float4 a = xa[step & 1][g], w = wa[step & 1][g];
          qin[g0 + g].x = fmaf(a.x, w.x, qin[g0 + g].x); qin[g0 + g].y = fmaf(a.y, w.y, qin[g0 + g].y);
          qin[g0 + g].z = fmaf(a.z, w.z, qin[g0 + g].z); qin[g0 + g].w = fmaf(a.w, w.w, qin[g0 + g].w);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) s += (qin[g].x + qin[g].y) + (qin[g].z + qin[g].w);
      s += lane_xor32(s);
      const float mean = s * (1.0f / C);
      float qq = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float a = qin[g].x - mean, b = qin[g].y - mean, c = qin[g].z - mean, d = qin[g].w - mean;
        qq += (a * a + b * b) + (c * c + d * d);
      }
      qq += lane_xor32(qq);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + p.epsq);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float4 gm = ld4(vecs + 9 * C + 8 * g + 4 * hf), bt = ld4(vecs + 10 * C + 8 * g + 4 * hf);
        qin[g].x = (qin[g].x - mean) * rstd * gm.x + bt.x; qin[g].y = (qin[g].y - mean) * rstd * gm.y + bt.y;
        qin[g].z = (qin[g].z - mean) * rstd * gm.z + bt.z; qin[g].w = (qin[g].w - mean) * rstd * gm.w + bt.w;
      }
    }
    // 16-bit storage: q_in leaves the fp32 registers here (it is rounded to the storage type exactly once, as the unfused path
    // does when it stores it): half the registers through phases C and D
    vec qb[F32 ? 1 : C / 16];
    if constexpr (!F32) {
#pragma unroll
      for (int s = 0; s < C / 16; ++s) {                               // k-step s = 2 u + j consumes channel groups 2 s, 2 s + 1
        f32x8 t8 = {qin[2 * s].x, qin[2 * s].y, qin[2 * s].z, qin[2 * s].w,
                    qin[2 * s + 1].x, qin[2 * s + 1].y, qin[2 * s + 1].z, qin[2 * s + 1].w};
        qb[s] = __builtin_convertvector(t8, vec);
      }
    }
    __syncthreads();                                                  // every wave is done with the halo tile
    stamp(2);

    // ---------------- phase C: K (rows = keys) and V^T (rows = channels) of this frame replace it; pads are zero
#pragma unroll
    for (int i = 0; i < NKP; ++i) {
      const int pc = tid + 256 * i;
      const int j = pc / NP, c4 = 4 * (pc - j * NP);
      const bool okj = j < p.Lk;
      const float4 kv4 = raw_to_f4(kraw[i]), vv4 = raw_to_f4(vraw[i]);
      // 16-bit storage: a 4-channel group keeps its order inside the permuted 32-block; groups of a 16-run go 0, 2, 1, 3
      const int kq = F32 ? c4 : (c4 & ~15) + 4 * (((c4 >> 2) & 1) * 2 + ((c4 >> 3) & 1));
      st4(Ks + j * PW + kq, okj ? kv4 : make_float4(0.f, 0.f, 0.f, 0.f));
      const int vq = F32 ? j : iperm32f(j);
      Vts[(c4 + 0) * PV + vq] = from_f<T>(okj ? vv4.x : 0.f);
      Vts[(c4 + 1) * PV + vq] = from_f<T>(okj ? vv4.y : 0.f);
      Vts[(c4 + 2) * PV + vq] = from_f<T>(okj ? vv4.z : 0.f);
      Vts[(c4 + 3) * PV + vq] = from_f<T>(okj ? vv4.w : 0.f);
    }
    __syncthreads();
    stamp(3);
    // the next tile's halo pieces: in flight during phase D
    if (AHEAD && vt + static_cast<int>(gridDim.x) < n_tiles) inside_mask = fetch_tile(tile_of(vt + gridDim.x));

    // ---------------- phase D (per wave, transposed): Q^T = Wq q_in^T + bq
    f32x16 qt[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {                                    // registers 4 g .. 4 g + 3 = channels 32 u + 8 g + 4 hf ..
        const float4 b4 = ld4(vecs + 11 * C + 32 * u + 8 * g + 4 * hf);
        qt[u][4 * g + 0] = b4.x; qt[u][4 * g + 1] = b4.y; qt[u][4 * g + 2] = b4.z; qt[u][4 * g + 3] = b4.w;
      }
    if constexpr (F32) {
      const float* wq0 = reinterpret_cast<const float*>(wqf);
      mfma_groups_f32<NG * NU>([&](int i) { return wq0 + 32 * (i % NU) * PW + 8 * (i / NU); },
                               [&](int i, float4 a) {
                                 const int g = i / NU, t = i % NU;
                                 DS_MFMA4(qt[t], a, qin[g].x, qin[g].y, qin[g].z, qin[g].w);
                               });
    } else {
      // k-step s = 2 u + j consumes channel groups 2 s, 2 s + 1; the NU weight fragments of step s + 1 are read before the
      // MFMAs of step s
      if constexpr (C <= 96) {
        vec af[2][NU];
#pragma unroll
        for (int t = 0; t < NU; ++t) af[0][t] = *reinterpret_cast<const vec*>(wqf + 32 * t * PW);
#pragma unroll
        for (int s = 0; s < C / 16; ++s) {
          if (s + 1 < C / 16) {
#pragma unroll
            for (int t = 0; t < NU; ++t) af[(s + 1) & 1][t] = *reinterpret_cast<const vec*>(wqf + 32 * t * PW + 16 * (s + 1));
          }
          const vec b = qb[s];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NU; ++t) qt[t] = Mma16<T>::mma(af[s & 1][t], b, qt[t]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // six row tiles: the fragments of a k-step are read half a step ahead (three at a time) -- the register file also
        // holds the next tile's halo pieces here
        constexpr int HT = NU / 2;
        vec af[2][HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) af[0][t] = *reinterpret_cast<const vec*>(wqf + 32 * t * PW);
#pragma unroll
        for (int i = 0; i < 2 * (C / 16); ++i) {                       // i = 2 s + half
          const int s = i >> 1, h0 = (i & 1) * HT;
          if (i + 1 < 2 * (C / 16)) {
            const int s1 = (i + 1) >> 1, h1 = ((i + 1) & 1) * HT;
#pragma unroll
            for (int t = 0; t < HT; ++t) af[(i + 1) & 1][t] = *reinterpret_cast<const vec*>(wqf + 32 * (h1 + t) * PW + 16 * s1);
          }
          const vec b = qb[s];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < HT; ++t) qt[h0 + t] = Mma16<T>::mma(af[i & 1][t], b, qt[h0 + t]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    // fp32: the V^T fragments of the O^T product below are read now, so that their LDS latency runs under the softmax
    float4 vfr[NU][HEADS][4];
    if constexpr (F32) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int hd = 0; hd < HEADS; ++hd) {
          if (32 * u + 32 <= hd * D || 32 * u >= (hd + 1) * D) continue;
          const bool mine = (32 * u >= hd * D && 32 * u + 32 <= (hd + 1) * D) || (32 * u + ml >= hd * D && 32 * u + ml < (hd + 1) * D);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 a = ld4(reinterpret_cast<const float*>(vtf) + 32 * u * PV + 8 * g);
            vfr[u][hd][g] = mine ? a : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
    }

    // ---------------- S^T = K Q^T per head (rows = keys), softmax over the keys, P^T
    f32x16 pt[HEADS];
#pragma unroll
    for (int hd = 0; hd < HEADS; ++hd) {
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      if constexpr (F32) {
        const float* k0 = reinterpret_cast<const float*>(kf);
        mfma_groups_f32<D / 8>([&](int i) { return k0 + 8 * (hd * (D / 8) + i); },          // channel groups of this head
                               [&](int i, float4 a) {
                                 const int g = hd * (D / 8) + i, u = g >> 2, r0 = 4 * (g & 3);
                                 DS_MFMA4(st, a, qt[u][r0 + 0], qt[u][r0 + 1], qt[u][r0 + 2], qt[u][r0 + 3]);
                               });
      } else {
#pragma unroll
        for (int s = hd * (D / 16); s < (hd + 1) * (D / 16); ++s) {     // k-steps (16 channels) of this head
          const int u = s >> 1, j = s & 1;
          f32x8 t8;
#pragma unroll
          for (int e = 0; e < 8; ++e) t8[e] = qt[u][8 * j + e];
          st = Mma16<T>::mma(*reinterpret_cast<const vec*>(kf + 16 * s), __builtin_convertvector(t8, vec), st);
        }
      }
      // key of register r in this lane half: (r & 3) + 8 (r >> 2) + 4 hf
      float mx = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = (r & 3) + 8 * (r >> 2) + 4 * hf < p.Lk;
        mx = ok ? fmaxf(mx, st[r]) : mx;
      }
      mx = fmaxf(mx, lane_xor32(mx));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = (r & 3) + 8 * (r >> 2) + 4 * hf < p.Lk;
        const float e = ok ? __expf((st[r] - mx) * p.scale) : 0.f;
        st[r] = e;
        sum += e;
      }
      sum += lane_xor32(sum);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int r = 0; r < 16; ++r) pt[hd][r] = st[r] * inv;
    }

    // ---------------- O^T = V^T P^T (rows = channels; a row tile that straddles the head boundary runs once per head with
    // the other head's rows of V^T masked to zero)
    f32x16 ot[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[u][r] = 0.f;
#pragma unroll
      for (int hd = 0; hd < HEADS; ++hd) {
        const int lo = hd * D, hi = (hd + 1) * D;                       // channel range of the head
        if (32 * u + 32 <= lo || 32 * u >= hi) continue;                // no row of this tile belongs to it
        const bool whole = 32 * u >= lo && 32 * u + 32 <= hi;
        const bool mine = whole || (32 * u + ml >= lo && 32 * u + ml < hi);
        if constexpr (F32) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (8 * g < p.Lk) {
              const float4 a = vfr[u][hd][g];
              DS_MFMA4(ot[u], a, pt[hd][4 * g + 0], pt[hd][4 * g + 1], pt[hd][4 * g + 2], pt[hd][4 * g + 3]);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (16 * j < p.Lk) {
              f32x8 t8;
#pragma unroll
              for (int e = 0; e < 8; ++e) t8[e] = pt[hd][8 * j + e];
              uint4 araw = *reinterpret_cast<const uint4*>(vtf + 32 * u * PV + 16 * j);
              if (!mine) araw = make_uint4(0u, 0u, 0u, 0u);
              ot[u] = Mma16<T>::mma(__builtin_bit_cast(vec, araw), __builtin_convertvector(t8, vec), ot[u]);
            }
          }
        }
      }
    }

    // ---------------- epilogue
    stamp(4);
    const int gy = y0 + trow, gx = x0 + tcol;
    const bool live = gy < p.H && gx < p.W;
    const long tokoff = ((static_cast<long>(n) * p.H + (live ? gy : 0)) * p.W + (live ? gx : 0)) * C + 4 * hf;
    if constexpr (F32) {
      // X1^T = Wp O^T + bp + x^T; the residual pieces are fetched before the product (L2 hits: the halo tile came from there)
      float4 xres[NU * 4];
      {
        const float* xs = reinterpret_cast<const float*>(p.x) + tokoff;
#pragma unroll
        for (int i = 0; i < NU * 4; ++i) xres[i] = ld4(xs + 8 * i);
      }
      f32x16 x1[NU];
#pragma unroll
      for (int t = 0; t < NU; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = ld4(vecs + 12 * C + 32 * t + 8 * g + 4 * hf);
          x1[t][4 * g + 0] = b4.x; x1[t][4 * g + 1] = b4.y; x1[t][4 * g + 2] = b4.z; x1[t][4 * g + 3] = b4.w;
        }
      {
        const float* wp0 = reinterpret_cast<const float*>(wpf);
        mfma_groups_f32<NU * 4 * NU>([&](int i) { return wp0 + 32 * (i % NU) * PW + 8 * (i / NU); },   // i = (4 u + g) NU + t
                                     [&](int i, float4 a) {
                                       const int t = i % NU, ug = i / NU, u = ug >> 2, g = ug & 3;
                                       DS_MFMA4(x1[t], a, ot[u][4 * g + 0], ot[u][4 * g + 1], ot[u][4 * g + 2], ot[u][4 * g + 3]);
                                     });
      }
      if (live) {
        float* dst = reinterpret_cast<float*>(p.out) + tokoff;
#pragma unroll
        for (int t = 0; t < NU; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 r4 = xres[4 * t + g];
            st4(dst + 32 * t + 8 * g, make_float4(x1[t][4 * g + 0] + r4.x, x1[t][4 * g + 1] + r4.y, x1[t][4 * g + 2] + r4.z,
                                                  x1[t][4 * g + 3] + r4.w));
          }
      }
    } else {
      if (live) {
        T* dst = p.out + tokoff;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            st4(dst + 32 * u + 8 * g, make_float4(ot[u][4 * g + 0], ot[u][4 * g + 1], ot[u][4 * g + 2], ot[u][4 * g + 3]));
      }
    }
    __syncthreads();                                                  // K / V^T are dead: the next halo tile may land
    stamp(5);
#ifdef DIFFSAL_DEV_STAMPS
    ++stamp_it;
#endif
  }
}

template <typename T, int C>
static int launch_front(const FrontArgs<T>& a, hipStream_t s) {
  typedef FrontLds<T, C> L;
  DS_RAISE_DYNAMIC_LDS((block_front_kernel<T, C>), 160 * 1024);
  const int n_tiles = a.N * a.tiles_y * a.tiles_x;
  const int resident = 256 * (L::total * 2 <= 160 * 1024 ? 2 : 1);
  const int grid = n_tiles < resident ? n_tiles : resident;
#ifdef DIFFSAL_DEV_STAMPS
  FrontArgs<T> b = a;
  b.stamps = (g_front_stamps && static_cast<size_t>(grid) * 64 * 8 <= g_front_stamp_bytes) ? g_front_stamps : nullptr;
  hipLaunchKernelGGL((block_front_kernel<T, C>), dim3(grid), dim3(256), L::total, s, b);
  return check_launch("block_front");
#endif
  hipLaunchKernelGGL((block_front_kernel<T, C>), dim3(grid), dim3(256), L::total, s, a);
  return check_launch("block_front");
}

}  // namespace diffsal

using namespace diffsal;

#ifdef DIFFSAL_DEV_STAMPS
// Development builds only (tools/probe_front_stamps.py): 8 stamps x 8 tiles per workgroup of every following launch
extern "C" int diffsal_set_front_stamps(void* device_buffer, size_t bytes) {
  diffsal::g_front_stamps = static_cast<unsigned long long*>(device_buffer);
  diffsal::g_front_stamp_bytes = device_buffer ? bytes : 0;
  return DIFFSAL_OK;
}
#endif

extern "C" int diffsal_block_front(const void* x, const void* k, const void* v, const float* g1, const float* b1, float eps1,
                                   const float* w9, const float* gq, const float* bq, float epsq, const void* wq,
                                   const float* bias_q, const void* wp, const float* bias_p, void* out, int N, int H, int W,
                                   int C, int Lk, int heads, float scale, int dtype, diffsal_stream_t stream) {
  DS_REQUIRE(x && k && v && g1 && b1 && w9 && gq && bq && wq && bias_q && out, DIFFSAL_E_ARG, "block_front: null argument");
  DS_REQUIRE((C == 96 || (C == 192 && dtype != DIFFSAL_F32)) && heads == 2, DIFFSAL_E_SHAPE,
             "block_front: built for C = 96 (and C = 192 on 16-bit storage), 2 heads (got C = %d, %d heads, dtype %d)", C, heads, dtype);
  DS_REQUIRE(N > 0 && H > 0 && W > 0 && Lk > 0 && Lk <= 32, DIFFSAL_E_SHAPE, "block_front: N=%d H=%d W=%d Lk=%d (Lk <= 32)", N, H, W, Lk);
  DS_REQUIRE(static_cast<long>(N) * H * W * C < (1L << 31), DIFFSAL_E_SHAPE, "block_front: tensor too large for one launch");
  DS_REQUIRE(dtype == DIFFSAL_F32 || dtype == DIFFSAL_BF16 || dtype == DIFFSAL_F16, DIFFSAL_E_ARG, "block_front: dtype %d", dtype);
  DS_REQUIRE(dtype != DIFFSAL_F32 || (wp && bias_p), DIFFSAL_E_ARG, "block_front: fp32 storage needs the output projection");
  DS_REQUIRE(aligned16(x) && aligned16(out) && aligned16(g1) && aligned16(b1) && aligned16(k) && aligned16(v) && aligned16(w9) &&
                 aligned16(gq) && aligned16(bq) && aligned16(wq) && aligned16(bias_q) && (!wp || aligned16(wp)) &&
                 (!bias_p || aligned16(bias_p)),
             DIFFSAL_E_ALIGN, "block_front: misaligned pointer (every operand is read in 16-byte pieces)");
  DS_REQUIRE(out != x, DIFFSAL_E_ARG, "block_front: in-place operation is not supported (neighbouring tiles read the halo)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_y = (H + FT_TH - 1) / FT_TH, tiles_x = (W + FT_TW - 1) / FT_TW;
#define DS_FRONT(T, CC)                                                                                                       \
  {                                                                                                                         \
    FrontArgs<T> a{static_cast<const T*>(x), static_cast<const T*>(k), static_cast<const T*>(v), g1, b1, w9, gq, bq,        \
                   static_cast<const T*>(wq), bias_q, static_cast<const T*>(wp), bias_p, static_cast<T*>(out), N, H, W, Lk, \
                   eps1, epsq, scale, tiles_y, tiles_x};                                                                    \
    return launch_front<T, CC>(a, s);                                                                                       \
  }
  if (dtype == DIFFSAL_F32) DS_FRONT(float, 96)
  if (C == 96) {
    if (dtype == DIFFSAL_BF16) DS_FRONT(bf16_t, 96)
    DS_FRONT(f16_t, 96)
  }
  if (dtype == DIFFSAL_BF16) DS_FRONT(bf16_t, 192)
  DS_FRONT(f16_t, 192)
#undef DS_FRONT
}
