// Implicit-GEMM convolution / linear layer on bf16 / fp16 STORAGE (BASELINE configs[1] "bf16", configs[4] "fp16"):
// activations, weights, residual and output are 16-bit in HBM, products run on the native 16-bit matrix cores
// (v_mfma_f32_32x32x16_{bf16,f16}) and accumulate in fp32; bias / BN affine / per-image vector stay fp32 and are
// applied to the fp32 accumulator before the single rounding to the storage type.
//
// Same operator, same im2col k order (ci/32, tap, ci%32) and same packed weight layout [Cout][K] as igemm.hip (the fp32
// kernel); what changes is the geometry of a K stage.  A 32-channel run of one pixel is 64 bytes here, so ONE LDS stage
// row holds TWO consecutive 32-k sub-slices (128 bytes, the same bytes per row and the same 36-dword pitch as the fp32
// kernel: conflict-free ds_read_b128).  The 8 loader lanes of a row split 4 + 4 over the two sub-slices, each half
// with its own (tap, channel-chunk) displacement; the matrix-core loop then runs four 16-wide k-steps per stage, one
// ds_read_b128 per 32-row fragment and step.  Half the HBM bytes and 1/8 of the matrix-pipe time of the fp32 kernel.
//
// Replaces the same reference call sites as diffsal_conv_igemm (see include/diffsal.h); selected by
// diffsal_conv_desc.dtype.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mma16;
template <> struct Mma16<__bf16> {
  typedef bf16x8 vec;
  static __device__ __forceinline__ f32x16 run(vec a, vec b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma16<_Float16> {
  typedef f16x8 vec;
  static __device__ __forceinline__ f32x16 run(vec a, vec b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <typename T>
struct Igemm16Args {
  const T* in;
  const T* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const T* residual;
  T* out;
  int M, K;
  int H, W, Cin, Ho, Wo, Cout;
  int KW, taps, stride_h, stride_w, pad_t, pad_l, dil_h, dil_w;
  int act, rowvec_ld;
  int linear;                // 1: 1x1 / stride 1 / no padding (a plain [M, Cin] x [Cin, Cout] product)
  int n_tiles_n, n_tiles;
  int xcd_order;   // persistent linear kernel: XCD-aware tile walk
  unsigned in_bytes, w_bytes;
  int splits, st_per_split;  // split-K over 64-k stages
  int vec_epilogue;          // 1: Cout % 4 == 0 and every epilogue pointer is aligned for 8 / 16-byte pieces
  int persist_wgs;           // > 0: workgroups of the persistent linear kernel (256 CUs x residents of the tile shape)
  float* partial;            // [splits][M][Cout] fp32 partial sums when splits > 1
  // pair launch (diffsal_linear_pair): a second problem of the same shape in the same grid (blockIdx.z = 1)
  int pair;
  const T* in2;
  const T* w2;
  const float* bias2;
  T* out2;
  float* partial2;
};

template <typename T>
__device__ __forceinline__ void select_pair16(Igemm16Args<T>& p, int which) {
  if (p.pair && which) { p.in = p.in2; p.w = p.w2; p.bias = p.bias2; p.out = p.out2; p.partial = p.partial2; }
}

constexpr int SUBK = 32;            // elements per sub-slice (one 64-byte run of a pixel)
constexpr int STK = 2 * SUBK;       // elements per LDS stage row
constexpr int PITCH16 = 36;         // dwords per LDS row (32 data + 4 pad)

template <int WM, int WN, int TM, int TN, typename T>
__global__ __launch_bounds__(256) void igemm16_kernel(Igemm16Args<T> p) {
  select_pair16(p, blockIdx.z);
  typedef typename Mma16<T>::vec frag_t;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int A_PASSES = BM / 32;
  constexpr int B_PASSES = BN / 32;
  constexpr int STAGE = (BM + BN) * PITCH16;  // dwords per LDS stage
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  int tile;
  const int split = blockIdx.y;
  {  // XCD-aware order: each XCD owns a contiguous run of tiles (see igemm.hip)
    const int nwg = p.n_tiles;
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tile_m = tile / p.n_tiles_n;
  const int tile_n = tile - tile_m * p.n_tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // loader: 8 lanes per row; lanes 0-3 fetch sub-slice 0 (4 x 16 B = 32 elements), lanes 4-7 sub-slice 1
  const int lrow = tid >> 3;
  const int l8 = tid & 7;
  const int lsub = l8 >> 2;
  const int lcol_dw = l8 * 4;  // dword column inside the LDS row
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(p.in), 0, static_cast<int>(p.in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(p.w), 0, static_cast<int>(p.w_bytes), 0x00020000);

  unsigned a_voff[A_PASSES];
  unsigned a_valid[A_PASSES];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < A_PASSES; ++j) {
    int m = m0 + lrow + 32 * j;
    m = m < p.M ? m : p.M - 1;
    if (p.linear) {   // 1x1, stride 1, no padding: row m of a [M, Cin] matrix -- skips three integer divisions per row
      a_voff[j] = static_cast<unsigned>(m * p.Cin + (l8 & 3) * 8) * 2u;
      a_valid[j] = 1u;
      continue;
    }
    const int n = m / HoWo;
    const int rem = m - n * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride_h - p.pad_t;
    const int ix0 = ox * p.stride_w - p.pad_l;
    a_voff[j] = static_cast<unsigned>(((n * p.H + iy0) * p.W + ix0) * p.Cin + (l8 & 3) * 8) * 2u;
    unsigned bits = 0;
    for (int t = 0; t < p.taps; ++t) {
      const int ky = t / p.KW, kx = t - ky * p.KW;
      const int iy = iy0 + ky * p.dil_h, ix = ix0 + kx * p.dil_w;
      bits |= ((iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W)) ? (1u << t) : 0u;
    }
    a_valid[j] = bits;
  }
  unsigned b_voff[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    int n = n0 + lrow + 32 * j;
    n = n < p.Cout ? n : p.Cout - 1;
    b_voff[j] = static_cast<unsigned>(n * p.K + l8 * 8) * 2u;
  }

  // Global loads run THREE stages ahead of the matrix cores (two register sets in flight + the stage being parked):
  // a stage's MFMA phase is ~8x shorter than in the fp32 kernel, so two stages ahead no longer covers an L2 round trip.
  float4 ra[2][A_PASSES], rb[2][B_PASSES];
  const int KT32 = p.K / SUBK;                 // sub-slices in K
  const int n_stages = (KT32 + 1) / 2;
  const int st_begin = split * p.st_per_split;
  const int st_end = min(n_stages, st_begin + p.st_per_split);
  const int nst = st_end - st_begin;

  auto issue_loads = [&](int st, bool live, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    // the two sub-slices of this stage: wave-uniform scalars, then a per-lane pick
    const int k0 = 2 * st, k1 = 2 * st + 1;
    const int c0 = k0 / p.taps, t0 = k0 - c0 * p.taps;
    const int c1 = k1 / p.taps, t1 = k1 - c1 * p.taps;
    const int ky0 = t0 / p.KW, kx0 = t0 - ky0 * p.KW;
    const int ky1 = t1 / p.KW, kx1 = t1 - ky1 * p.KW;
    const unsigned d0 = static_cast<unsigned>((ky0 * p.dil_h * p.W + kx0 * p.dil_w) * p.Cin + c0 * SUBK) * 2u;
    const unsigned d1 = static_cast<unsigned>((ky1 * p.dil_h * p.W + kx1 * p.dil_w) * p.Cin + c1 * SUBK) * 2u;
    const unsigned dead0 = (live && k0 < KT32) ? 0u : 0xFFFFFFFFu;
    const unsigned dead1 = (live && k1 < KT32) ? 0u : 0xFFFFFFFFu;
    const unsigned delta = lsub ? d1 : d0;
    const unsigned dead = lsub ? dead1 : dead0;
    const int tap = lsub ? t1 : t0;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) {
      const unsigned oob = ((a_valid[j] >> tap) & 1u) - 1u;
      ra[SET][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (a_voff[j] + delta) | oob | dead, 0, 0));
    }
    const unsigned kofs = static_cast<unsigned>(st * STK) * 2u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[SET][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (b_voff[j] + kofs) | dead, 0, 0));
  };
  auto store_tile = [&](float* stage, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) st4(&stage[(lrow + 32 * j) * PITCH16 + lcol_dw], ra[SET][j]);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) st4(&stage[(BM + lrow + 32 * j) * PITCH16 + lcol_dw], rb[SET][j]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment of k-step kk (16 elements = 8 dwords of the row): lane half h supplies elements 8h .. 8h+7 of the step
  const int frow = lane & 31;
  const int fk = (lane >> 5) * 4;
  const int a_frag = (wm * TM * 32 + frow) * PITCH16 + fk;
  const int b_frag = (BM + wn * TN * 32 + frow) * PITCH16 + fk;
  float4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](const float* stage, int kk, int set) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[set][i] = ld4(stage + a_frag + i * 32 * PITCH16 + kk * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[set][j] = ld4(stage + b_frag + j * 32 * PITCH16 + kk * 8);
  };
  auto do_mfmas = [&](int set) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = Mma16<T>::run(__builtin_bit_cast(frag_t, fb[set][j]), __builtin_bit_cast(frag_t, fa[set][i]), acc[i][j]);  // D^T
  };

  // prologue: stages 0, 1, 2 in flight together; stage 0 is parked, 1 and 2 stay in the two register sets
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  issue_loads(st_begin, true, S0{});
  issue_loads(st_begin + 1, nst > 1, S1{});
  store_tile(smem, S0{});                 // waits for stage 0 only (vmcnt is in issue order)
  issue_loads(st_begin + 2, nst > 2, S0{});
  __syncthreads();
  load_frags(smem, 0, 0);
  // iteration `it` computes stage it from LDS, parks stage it+1 (register set (it+1) & 1) and re-uses that set for
  // stage it+3; the set index must be a compile-time constant (register arrays), hence the two-way unrolled loop
  auto step = [&](int it, auto park_c) {
    float* cur = smem + (it & 1) * STAGE;
    float* nxt = smem + ((it & 1) ^ 1) * STAGE;
    load_frags(cur, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(0);
    store_tile(nxt, park_c);
    issue_loads(st_begin + it + 3, it + 3 < nst, park_c);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(cur, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(cur, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // every read of `cur` and every write of `nxt` has been issued
    load_frags(nxt, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int it = 0;
    for (; it + 1 < nst; it += 2) {
      step(it, S1{});       // stage it+1 sits in set 1 (odd stages), then set 1 fetches stage it+3
      step(it + 1, S0{});   // stage it+2 sits in set 0
    }
    if (it < nst) step(it, S1{});
  }

  // ---- epilogue.  The weight fragment is the MFMA "A" operand, so D is the transposed tile: a lane holds, for output
  // row (pixel) lane & 31 of its 32-row block, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- register quads of four
  // consecutive channels.  Two-byte stores straight from that layout (what round 2 started with) cost more than the
  // whole K loop on the M ~ 2e5 layers; a quad leaves as one 8-byte (16-bit output) / 16-byte (split-K slab) store.
  const int col_l = lane & 31;
  const int hq = (lane >> 5) * 4;
  const T* __restrict__ resid = p.residual;
  const float* __restrict__ rowv = p.rowvec;
  T* __restrict__ outp = p.out;
  float* part = p.splits > 1 ? p.partial + static_cast<long>(split) * p.M * p.Cout : nullptr;
  if (!p.vec_epilogue) {   // Cout % 4 != 0 or pointers not 8 / 16-byte aligned: scalar stores from the register layout
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + (wm * TM + i) * 32 + col_l;
      if (m >= p.M) continue;
      const int img = rowv ? m / HoWo : 0;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + (wn * TN + j) * 32 + (r & 3) + 8 * (r >> 2) + hq;
          if (n >= p.Cout) continue;
          const long o = static_cast<long>(m) * p.Cout + n;
          float v = acc[i][j][r];
          if (part) { part[o] = v; continue; }
          if (p.bias) v += p.bias[n];
          if (p.scale) v = v * p.scale[n] + (p.shift ? p.shift[n] : 0.f);
          if (rowv) v += rowv[static_cast<long>(img) * p.rowvec_ld + n];
          if (p.act == DIFFSAL_ACT_RELU) v = fmaxf(v, 0.f);
          else if (p.act == DIFFSAL_ACT_GELU_ERF) v = gelu_erf(v);
          else if (p.act == DIFFSAL_ACT_SIGMOID) v = sigmoidf_(v);
          if (resid) v += static_cast<float>(resid[o]);
          outp[o] = static_cast<T>(v);
        }
      }
    }
    return;
  }
  // 8-byte stores (16-byte for a split-K slab) straight from the register quads: no LDS staging, no barrier, and measured
  // 5-15 % faster than a coalescing pass through LDS -- what made the first version of this epilogue slow was the TWO-byte
  // store per lane of the untransposed layout, not the 16-byte segments per row.
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wm * TM + i) * 32 + col_l;
    if (m >= p.M) continue;
    const float* rv_row = rowv ? rowv + static_cast<long>(m / HoWo) * p.rowvec_ld : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (wn * TN + j) * 32 + g * 8 + hq;
        if (n >= p.Cout) continue;
        const long o = static_cast<long>(m) * p.Cout + n;
        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (part) { st4(part + o, make_float4(v[0], v[1], v[2], v[3])); continue; }
        if (p.bias) { const float4 t = ld4(p.bias + n); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
        if (p.scale) {
          const float4 sc = ld4(p.scale + n);
          const float4 sh = p.shift ? ld4(p.shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
        }
        if (rv_row) { const float4 t = ld4(rv_row + n); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
        if (p.act == DIFFSAL_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == DIFFSAL_ACT_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if (p.act == DIFFSAL_ACT_SIGMOID) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = sigmoidf_(v[e]);
        }
        if (resid) { const float4 t = ld4(resid + o); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
        st4(outp + o, make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  }
}

// =================================================================================================================
// Persistent form for plain [M, K] x [K, N] products (1x1 convolutions / Linear layers, no split-K).  The short-K token
// GEMMs of the transformer stages spend ~0.7 us of an ~8 us workgroup lifetime in MFMAs: the rest is index arithmetic, the
// first-load latency and the store tail, once per 128-row tile (in-kernel stamps).  Here a workgroup
// walks a strided list of tiles and the three-stage-ahead prefetch simply runs on into the next tile -- a tile's origin is
// two byte offsets, so there is no per-tile setup to speak of -- while the register-direct epilogue needs neither LDS nor a
// barrier.  Rows past M and columns past Cout read zeros through the buffer descriptors' range check.
// =================================================================================================================
template <int WM, int WN, int TM, int TN, typename T>
__global__ __launch_bounds__(256, (TM * TN <= 4 ? 2 : 1)) void igemm16_linear_kernel(Igemm16Args<T> p) {
  typedef typename Mma16<T>::vec frag_t;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int A_PASSES = BM / 32;
  constexpr int B_PASSES = BN / 32;
  constexpr int STAGE = (BM + BN) * PITCH16;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = tid >> 3, l8 = tid & 7, lsub = l8 >> 2, lcol_dw = l8 * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.in), 0, static_cast<int>(p.in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.w), 0, static_cast<int>(p.w_bytes), 0x00020000);
  unsigned a_rel[A_PASSES], b_rel[B_PASSES];   // byte offsets of this thread's 16-byte piece relative to the tile origin
#pragma unroll
  for (int j = 0; j < A_PASSES; ++j) a_rel[j] = static_cast<unsigned>((lrow + 32 * j) * p.K + l8 * 8) * 2u;
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) b_rel[j] = static_cast<unsigned>((lrow + 32 * j) * p.K + l8 * 8) * 2u;

  const int KT32 = p.K / SUBK;
  const int nst = (KT32 + 1) / 2;
  // tile walk: plain (v = tile, N fastest) or XCD order (one XCD owns whole M-tile rows) -- see igemm_linear_kernel in igemm.hip
  int n_tiles = p.n_tiles;
  if (p.xcd_order) {
    const int x = blockIdx.x & 7, ntm = p.n_tiles / p.n_tiles_n;
    n_tiles = 8 * p.n_tiles_n * (ntm > x ? (ntm - x + 7) >> 3 : 0);
  }
  auto tile_mn = [&](int v, int& tmi, int& tni) __attribute__((always_inline)) {
    if (p.xcd_order) {
      const int q = v >> 3, ml = q / p.n_tiles_n;
      tni = q - ml * p.n_tiles_n;
      tmi = ml * 8 + (v & 7);
    } else {
      tmi = v / p.n_tiles_n;
      tni = v - tmi * p.n_tiles_n;
    }
  };
  const int my_tiles = n_tiles > static_cast<int>(blockIdx.x)
                           ? (n_tiles - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x) : 0;
  const int total = my_tiles * nst;                 // global stage count of this workgroup
  // issue side: the next stage to fetch, as (tile, stage) with the tile's two origins
  int iss_w = blockIdx.x, iss_st = 0;
  unsigned iss_a = 0, iss_b = 0;
  auto origins = [&](int w, unsigned& oa, unsigned& ob) __attribute__((always_inline)) {
    int tmi, tni;
    tile_mn(w, tmi, tni);
    oa = static_cast<unsigned>(tmi) * static_cast<unsigned>(BM * p.K * 2);
    ob = static_cast<unsigned>(tni) * static_cast<unsigned>(BN * p.K * 2);
  };
  origins(iss_w, iss_a, iss_b);
  float4 ra[2][A_PASSES], rb[2][B_PASSES];
  auto issue_next = [&](auto set_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const bool live = iss_w < n_tiles;
    const int ks = 2 * iss_st + lsub;
    const unsigned dead = (live && ks < KT32) ? 0u : 0xFFFFFFFFu;
    const unsigned kofs = static_cast<unsigned>(iss_st * STK) * 2u;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j)
      ra[SET][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (iss_a + a_rel[j] + kofs) | dead, 0, 0));
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[SET][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (iss_b + b_rel[j] + kofs) | dead, 0, 0));
    if (++iss_st == nst) {
      iss_st = 0;
      iss_w += gridDim.x;
      if (iss_w < n_tiles) origins(iss_w, iss_a, iss_b);
    }
  };
  auto store_tile = [&](float* stage, auto set_c) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) st4(&stage[(lrow + 32 * j) * PITCH16 + lcol_dw], ra[SET][j]);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) st4(&stage[(BM + lrow + 32 * j) * PITCH16 + lcol_dw], rb[SET][j]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const int a_frag = (wm * TM * 32 + frow) * PITCH16 + fk;
  const int b_frag = (BM + wn * TN * 32 + frow) * PITCH16 + fk;
  float4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](const float* stage, int kk, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[set][i] = ld4(stage + a_frag + i * 32 * PITCH16 + kk * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[set][j] = ld4(stage + b_frag + j * 32 * PITCH16 + kk * 8);
  };
  auto do_mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = Mma16<T>::run(__builtin_bit_cast(frag_t, fb[set][j]), __builtin_bit_cast(frag_t, fa[set][i]), acc[i][j]);  // D^T
  };

  // compute side: which tile the stages being multiplied belong to
  int cmp_w = blockIdx.x, cmp_st = 0;
  const int col_l = lane & 31, hq = (lane >> 5) * 4;
  const T* __restrict__ resid = p.residual;
  T* __restrict__ outp = p.out;
  // the residual quads of the tile being multiplied are fetched at its first k-step and wait in registers: in the epilogue
  // they would be a dependent round trip with the matrix pipe idle
  uint2 rres[TM][TN][4];
  auto fetch_residual = [&]() __attribute__((always_inline)) {
    int tmi, tni;
    tile_mn(cmp_w, tmi, tni);
    const int m0 = tmi * BM, n0 = tni * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + (wm * TM + i) * 32 + col_l;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + (wn * TN + j) * 32 + g * 8 + hq;
          const bool ok = m < p.M && n < p.Cout;
          const long o = ok ? static_cast<long>(m) * p.Cout + n : 0;
          rres[i][j][g] = *reinterpret_cast<const uint2*>(resid + o);
        }
    }
  };
  auto finish_tile = [&]() __attribute__((always_inline)) {
    int tmi, tni;
    tile_mn(cmp_w, tmi, tni);
    const int m0 = tmi * BM, n0 = tni * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + (wm * TM + i) * 32 + col_l;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + (wn * TN + j) * 32 + g * 8 + hq;
          float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          acc[i][j][4 * g] = 0.f; acc[i][j][4 * g + 1] = 0.f; acc[i][j][4 * g + 2] = 0.f; acc[i][j][4 * g + 3] = 0.f;
          if (m >= p.M || n >= p.Cout) continue;
          const long o = static_cast<long>(m) * p.Cout + n;
          if (p.bias) { const float4 t = ld4(p.bias + n); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
          if (p.scale) {
            const float4 sc = ld4(p.scale + n);
            const float4 sh = p.shift ? ld4(p.shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
          }
          if (p.rowvec) {
            const float4 t = ld4(p.rowvec + static_cast<long>(m / (p.Ho * p.Wo)) * p.rowvec_ld + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
          }
          if (p.act == DIFFSAL_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (p.act == DIFFSAL_ACT_GELU_ERF) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          } else if (p.act == DIFFSAL_ACT_SIGMOID) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = sigmoidf_(v[e]);
          }
          if (resid) {
            T rt[4];
            __builtin_memcpy(rt, &rres[i][j][g], 8);
            v[0] += static_cast<float>(rt[0]); v[1] += static_cast<float>(rt[1]); v[2] += static_cast<float>(rt[2]); v[3] += static_cast<float>(rt[3]);
          }
          st4(outp + o, make_float4(v[0], v[1], v[2], v[3]));
        }
      }
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  issue_next(S0{});                       // global stage 0
  issue_next(S1{});                       // 1
  store_tile(smem, S0{});
  issue_next(S0{});                       // 2
  __syncthreads();
  load_frags(smem, 0, 0);
  // global step g multiplies stage g from LDS, parks stage g+1 (register set (g+1) & 1) and re-uses that set for g+3
  auto step = [&](int g, auto park_c) __attribute__((always_inline)) {
    float* cur = smem + (g & 1) * STAGE;
    float* nxt = smem + ((g & 1) ^ 1) * STAGE;
    if (resid && cmp_st == 0) fetch_residual();
    load_frags(cur, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(0);
    store_tile(nxt, park_c);
    issue_next(park_c);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(cur, 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(cur, 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    load_frags(nxt, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
    if (++cmp_st == nst) {                // that was the tile's last k-step: store it, start the next one from zero
      finish_tile();
      cmp_st = 0;
      cmp_w += gridDim.x;
    }
  };
  int g = 0;
  for (; g + 1 < total; g += 2) {
    step(g, S1{});
    step(g + 1, S0{});
  }
  if (g < total) step(g, S1{});
}

// Sum the fp32 split-K slabs in a fixed order, apply the epilogue, round once to the storage type.
template <typename T>
__global__ __launch_bounds__(256) void splitk16_reduce_kernel(Igemm16Args<T> p) {
  select_pair16(p, blockIdx.y);
  const int n4 = p.Cout >> 2;
  const long total = static_cast<long>(p.M) * n4;
  const long slab = static_cast<long>(p.M) * p.Cout;
  const int HoWo = p.Ho * p.Wo;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int n = static_cast<int>(i % n4) * 4;
    const long m = i / n4;
    const long o = m * p.Cout + n;
    float4 a = ld4(p.partial + o);
    for (int s = 1; s < p.splits; ++s) {
      const float4 b = ld4(p.partial + s * slab + o);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    float4 rs = make_float4(0, 0, 0, 0);
    if (p.residual) rs = ld4(p.residual + o);
    const float rr[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j];
      if (p.bias) x += p.bias[n + j];
      if (p.scale) x = x * p.scale[n + j] + p.shift[n + j];
      if (p.rowvec) x += p.rowvec[(m / HoWo) * p.rowvec_ld + n + j];
      if (p.act == DIFFSAL_ACT_RELU) x = fmaxf(x, 0.f);
      else if (p.act == DIFFSAL_ACT_GELU_ERF) x = gelu_erf(x);
      else if (p.act == DIFFSAL_ACT_SIGMOID) x = sigmoidf_(x);
      v[j] = x + rr[j];
    }
    st4(p.out + o, make_float4(v[0], v[1], v[2], v[3]));
  }
}

struct TileCfg16 { int bm, bn, occ; float eff; };
// order must match the dispatch switch below
// eff: measured on the network's shapes (tools/tune_igemm16.py, profiles/r02_tune_igemm16_bf16.log): with operands
// delivered through L1 at ~27 B/clk/CU the two-resident 128x96 / 128x128 tiles win almost everywhere; the 256-row
// tiles (one resident workgroup, two stages of prefetch) do not cover the load latency yet and lose 30-40 %.
static const TileCfg16 kCfgs16[] = {{128, 192, 1, 0.62f}, {128, 128, 2, 0.70f}, {128, 96, 2, 0.75f}, {64, 128, 2, 0.55f},
                                    {128, 64, 2, 0.55f},  {64, 64, 4, 0.40f},   {256, 96, 1, 0.50f},  {256, 128, 1, 0.50f}};
constexpr int kNumCfgs16 = 8;
constexpr int kCUs16 = 256;

struct Plan16 { int cfg, splits; };

// Same analytic model as igemm.hip's planner with the 16-bit matrix rate (2.5 PFLOP/s dense); `eff` folds in the
// LDS-bandwidth limit of the narrower wave tiles (reads per MFMA), fixed latencies weigh 16x more than in fp32.
static Plan16 choose_plan16(long M, int Cout, int K) {
  const double mac_per_s_cu = 1.0e15 / 2.0 / kCUs16;   // what the loop sustains at eff = 1 (not the 2.5 PF/s pipe peak)
  const double t_fixed = 6e-6;
  const int NST = (K / SUBK + 1) / 2;
  Plan16 best{5, 1};
  double best_t = 1e30;
  for (int c = 0; c < kNumCfgs16; ++c) {
    const TileCfg16& t = kCfgs16[c];
    if (t.bn > Cout && t.bn - Cout >= 32 && c != 5) continue;
    const long tiles = ((M + t.bm - 1) / t.bm) * ((Cout + t.bn - 1) / t.bn);
    for (int S = 1; S <= 16; S *= 2) {
      if (S > 1 && (NST / S < 4 || Cout % 4 != 0)) break;
      const long wgs = tiles * S;
      const long slots = static_cast<long>(kCUs16) * t.occ;
      const double rounds = static_cast<double>((wgs + slots - 1) / slots);
      const int st_per = (NST + S - 1) / S;
      const double t_mfma = static_cast<double>(t.bm) * t.bn * (st_per * STK) / (mac_per_s_cu * t.eff);
      const double resident = static_cast<double>(wgs < slots ? (wgs + kCUs16 - 1) / kCUs16 : t.occ);
      // a lone workgroup on a CU is bound by the latency of a stage (load -> LDS -> MFMA: 0.35 us + 0.1 us per 64 x 64 of tile), not
      // by the matrix rate: long-K products with few tiles (res2's convolutions, UpEmbed-1's extended-grid convolution, K = 6912)
      // take the K split.  Constants fitted to the sweep of every (tile, split) on the step's 23 convolution shapes
      // (tools/tune_igemm16.py, TUNE_SPLITS=0,1,2): regret 69 -> 28 us per step; 53 -> 39 us, 75 -> 52 us on the two named above
      const double t_stage = 0.35e-6 + 0.1e-6 * (t.bm * t.bn / 4096.0);
      const double t_alone = (t_mfma > st_per * t_stage ? t_mfma : st_per * t_stage) + t_fixed;
      const double round = resident * t_mfma > t_alone ? resident * t_mfma : t_alone;
      double tt = rounds * round;
      if (S > 1) tt += (S + 1.0) * M * Cout * 4.0 / 3.0e12 + 4.0e-6 + 0.5e-6 * S;
      if (tt < best_t) { best_t = tt; best = Plan16{c, S}; }
    }
  }
  return best;
}

template <int WM, int WN, int TM, int TN, typename T>
static int launch16(Igemm16Args<T>& a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  a.n_tiles_n = (a.Cout + BN - 1) / BN;
  const int tiles_m = (a.M + BM - 1) / BM;
  a.n_tiles = a.n_tiles_n * tiles_m;
  const unsigned nz = a.pair ? 2u : 1u;
  if (a.linear && a.splits == 1 && a.vec_epilogue && a.persist_wgs > 0 && !a.pair) {   // plain product: persistent workgroups
    const int grid = a.n_tiles < a.persist_wgs ? a.n_tiles : a.persist_wgs;
    a.xcd_order = (tune(TUNE_NO_XCD_ORDER) != 1 && grid % 8 == 0 && a.n_tiles >= a.persist_wgs && a.n_tiles_n > 1 && tiles_m >= 16) ? 1 : 0;
    hipLaunchKernelGGL((igemm16_linear_kernel<WM, WN, TM, TN, T>), dim3(grid), dim3(256), 0, s, a);
    note_kernel("igemm16_linear_kernel<%d, %d, %d, %d> [%dx%d tile]", WM, WN, TM, TN, BM, BN);
    return check_launch("diffsal_conv_igemm(16-bit linear)");
  }
  hipLaunchKernelGGL((igemm16_kernel<WM, WN, TM, TN, T>), dim3(a.n_tiles, a.splits, nz), dim3(256), 0, s, a);
  note_kernel("igemm16_kernel<%d, %d, %d, %d> [%dx%d tile, split-K %d%s]", WM, WN, TM, TN, BM, BN, a.splits, a.pair ? ", pair" : "");
  int rc = check_launch("diffsal_conv_igemm(16-bit)");
  if (rc || a.splits == 1) return rc;
  const long total4 = static_cast<long>(a.M) * (a.Cout / 4);
  long g = (total4 + 255) / 256;
  g = g > 2048 ? 2048 : g;
  hipLaunchKernelGGL((splitk16_reduce_kernel<T>), dim3(static_cast<int>(g), nz), dim3(256), 0, s, a);
  return check_launch("diffsal_conv_igemm(16-bit split-K reduce)");
}

// conv16_halo.hip: LDS-halo kernel for 3x3 stride-1 "same" convolutions
int conv16_halo_applies(const diffsal_conv_desc* d);
int conv16_halo_pointers_ok(const diffsal_conv_desc* d, const float* bias, const float* scale, const float* shift,
                            const float* rowvec, const void* residual, const void* out);
int conv16_halo_launch(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                       const float* shift, const float* rowvec, const void* residual, void* out, hipStream_t s);

// conv16_dma.hip: the same convolutions (any padding) with an LDS-DMA halo patch, two workgroups per CU
int conv16_dma_applies(const diffsal_conv_desc* d, const float* bias, const float* scale, const float* shift, const float* rowvec,
                       const void* residual, const void* out);
int conv16_dma_launch(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                      const float* shift, const float* rowvec, const void* residual, void* out, hipStream_t s);

static Plan16 plan_for(const diffsal_conv_desc* d) {
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  Plan16 pl = choose_plan16(M, d->Cout, d->KH * d->KW * d->Cin);
  if (tune(TUNE_IGEMM16_CFG) >= 0) {  // tuning aid: force a tile shape (value % 8) and 2^(value / 8) K splits
    const int v = tune(TUNE_IGEMM16_CFG);
    pl.cfg = v % kNumCfgs16;
    pl.splits = 1;
    const int NST = (d->KH * d->KW * d->Cin / SUBK + 1) / 2;
    for (int e = 0; e < v / kNumCfgs16 && e < 4; ++e)
      if (NST / (pl.splits * 2) >= 4 && d->Cout % 4 == 0) pl.splits *= 2;
  }
  if (conv16_halo_applies(d)) pl.splits = 1;   // halo-eligible shapes report no workspace; keep the pointer-alignment fallback valid
  return pl;
}

size_t igemm16_ws_bytes(const diffsal_conv_desc* d) {
  if (conv16_halo_applies(d)) return 0;
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const Plan16 pl = plan_for(d);
  return pl.splits > 1 ? static_cast<size_t>(pl.splits) * M * d->Cout * sizeof(float) : 0;
}

template <typename T>
static int run16(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                 const float* shift, const float* rowvec, const void* residual, void* out, void* ws, size_t ws_bytes,
                 hipStream_t s, const void* in2, const void* w2, const float* bias2, void* out2) {
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  Igemm16Args<T> a;
  a.xcd_order = 0;
  a.pair = in2 ? 1 : 0;
  a.in2 = static_cast<const T*>(in2); a.w2 = static_cast<const T*>(w2); a.bias2 = bias2; a.out2 = static_cast<T*>(out2);
  a.partial2 = nullptr;
  a.in = static_cast<const T*>(in); a.w = static_cast<const T*>(w); a.bias = bias; a.scale = scale; a.shift = shift;
  a.rowvec = rowvec; a.residual = static_cast<const T*>(residual); a.out = static_cast<T*>(out);
  a.M = static_cast<int>(M);
  a.K = d->KH * d->KW * d->Cin;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KW = d->KW; a.taps = d->KH * d->KW; a.stride_h = d->stride_h; a.stride_w = d->stride_w;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.dil_h = d->dil_h; a.dil_w = d->dil_w; a.act = d->act;
  a.rowvec_ld = d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout;
  a.linear = d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_t == 0 && d->pad_l == 0 && d->Ho == d->H && d->Wo == d->W;
  a.in_bytes = static_cast<unsigned>(static_cast<long>(d->N) * d->H * d->W * d->Cin * 2);
  a.w_bytes = static_cast<unsigned>(static_cast<long>(d->Cout) * a.K * 2);
  const Plan16 pl = plan_for(d);
  a.splits = pl.splits;
  const int n_stages = (a.K / SUBK + 1) / 2;
  a.st_per_split = (n_stages + pl.splits - 1) / pl.splits;
  a.partial = nullptr;
  if (pl.splits > 1) {
    const size_t one = static_cast<size_t>(pl.splits) * M * d->Cout * sizeof(float);
    const size_t need = in2 ? 2 * one : one;
    DS_REQUIRE(ws && ws_bytes >= need && aligned16(ws) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 &&
                   (!residual || (reinterpret_cast<uintptr_t>(residual) & 7u) == 0),
               DIFFSAL_E_ARG,
               "conv_igemm(16-bit): split-K needs %zu bytes of 16-byte aligned workspace (diffsal_conv_igemm_ws_bytes), got %zu",
               need, ws_bytes);
    a.partial = static_cast<float*>(ws);
    a.partial2 = in2 ? a.partial + one / sizeof(float) : nullptr;
  }
  {
    auto al = [](const void* q, uintptr_t m) { return (reinterpret_cast<uintptr_t>(q) & m) == 0; };
    a.vec_epilogue = d->Cout % 4 == 0 && al(out, 7) && al(residual, 7) && al(bias, 15) && al(scale, 15) && al(shift, 15) &&
                     al(rowvec, 15) && (!rowvec || a.rowvec_ld % 4 == 0) && al(a.partial, 15) && al(out2, 7) && al(bias2, 15);
  }
  a.persist_wgs = kCUs16 * kCfgs16[pl.cfg].occ;
  if (tune(TUNE_NO_PERSIST) == 1) a.persist_wgs = 0;
  switch (pl.cfg) {
    case 0: return launch16<2, 2, 2, 3, T>(a, s);
    case 1: return launch16<2, 2, 2, 2, T>(a, s);
    case 2: return launch16<4, 1, 1, 3, T>(a, s);
    case 3: return launch16<2, 2, 1, 2, T>(a, s);
    case 4: return launch16<2, 2, 2, 1, T>(a, s);
    case 6: return launch16<4, 1, 2, 3, T>(a, s);
    case 7: return launch16<4, 1, 2, 4, T>(a, s);
    default: return launch16<2, 2, 1, 1, T>(a, s);
  }
}

int igemm16_launch(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                   const float* shift, const float* rowvec, const void* residual, void* out, void* ws, size_t ws_bytes,
                   hipStream_t s, const void* in2, const void* w2, const float* bias2, void* out2) {
  if (!in2 && conv16_dma_applies(d, bias, scale, shift, rowvec, residual, out))
    return conv16_dma_launch(d, in, w, bias, scale, shift, rowvec, residual, out, s);
  if (!in2 && conv16_halo_applies(d) && conv16_halo_pointers_ok(d, bias, scale, shift, rowvec, residual, out))
    return conv16_halo_launch(d, in, w, bias, scale, shift, rowvec, residual, out, s);
  if (d->dtype == DIFFSAL_BF16)
    return run16<__bf16>(d, in, w, bias, scale, shift, rowvec, residual, out, ws, ws_bytes, s, in2, w2, bias2, out2);
  return run16<_Float16>(d, in, w, bias, scale, shift, rowvec, residual, out, ws, ws_bytes, s, in2, w2, bias2, out2);
}

}  // namespace diffsal
