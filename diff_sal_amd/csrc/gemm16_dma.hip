// Plain [M, K] x [N, K]^T products on bf16 / fp16 storage (token GEMMs, tap products, ReduceTemp's row products), two workgroups
// per CU: the plain-product sibling of conv16_dma.hip.
//
// gemm_dma.hip's 16-bit instantiation keeps ONE wavefront per SIMD busy with everything -- DMA issue, fragment reads, barrier --
// and the matrix pipe idles through each of them: 0.17 busy, 0.4-0.6 PF/s at 64 clips whatever the tile (96 x 96 and 192 x 192
// measured equal).  Here, as in conv16_dma.hip:
//   * a workgroup is 4 wavefronts and a 256 x 96 output tile (a wavefront: 64 rows x 96 columns = 2 x 3 accumulators of
//     v_mfma_f32_32x32x16); its LDS -- a three-slot ring of (256 + 96) rows x 64 bytes, 67 KB -- leaves room for a second,
//     independent workgroup per CU, whose matrix work covers this one's DMA issue, barriers, prologue and epilogue;
//   * both operands go memory -> LDS by LDS-DMA, 64-byte rows (one 32-element K chunk) with the four 16-byte slots XOR-swizzled by
//     (row >> 2) & 3 on the source side; rows past M / N are fetched out of range (zeros);
//   * a step is one K chunk: counted wait for the chunk issued two steps ago (every wavefront issues six DMA instructions per step),
//     one barrier, issue the chunk two ahead, 12 MFMAs per wavefront;
//   * epilogue through LDS, 16-byte (or, fp32 output, 32-byte) pieces per lane: bias, affine, per-image vector, activation,
//     residual in the order of the other 16-bit kernels.  Accumulation: K in order, fp32 -- the same sums as igemm16 / gemm_dma.
//
// R/models/saliency_decoder/attention.py:97-111 (proj_q / k / v, proj), common_block.py:125-147 (Mlp), transformer.py:150-157.
#include "common.h"

namespace diffsal {

typedef float gd_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 gd_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gd_f16x8 __attribute__((ext_vector_type(8)));
typedef int gd_i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* gd_lds_ptr_t;

template <typename T> struct GdMma;
template <> struct GdMma<__bf16> {
  typedef gd_bf16x8 vec;
  static __device__ __forceinline__ gd_f32x16 run(vec a, vec b, gd_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct GdMma<_Float16> {
  typedef gd_f16x8 vec;
  static __device__ __forceinline__ gd_f32x16 run(vec a, vec b, gd_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T>
struct GdArgs {
  const T* a;
  const T* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const T* residual;
  void* out;
  int M, N, K;
  int act, rowvec_ld, rows_per_img, out_f32;
  int tiles_m, tiles_n;
  // row form (ReduceTemp, R/models/saliency_decoder/sal_unet.py:300-318: a (kt, 1) kernel with stride (kt, 1) over [B, T, HW, C] that
  // leaves ONE frame): taps > 1: output row m = (image m / Wrow, position m % Wrow) reads input rows (image, t, position), t < taps;
  // K is ordered (32-channel chunk, tap, channel) as pack_conv_weight leaves it.  taps == 1: the plain product (Hrow = 1)
  int taps, Wrow, Hrow, Cin;
};

// inline assembly on purpose (see gemm_dma.hip): the compiler must know neither the LDS write nor the vmcnt event
__device__ __forceinline__ void gd_dma(unsigned lds_addr, unsigned voff, gd_i32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void gd_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// Two tile shapes, four wavefronts each, a wavefront = TM x 3 accumulators of 32 x 32:
//   <2, 1>  256 x 96: the wavefronts stacked along M (64 rows x 96 columns each);
//   <3, 2>  192 x 192: the wavefronts 2 x 2 (96 x 96 each).  Per MFMA a wavefront reads (1 / TM + 1 / 3) KiB of fragments from LDS:
//           0.83 against 0.67 -- at one MFMA per 8 cycles and CU that is 81 % of the LDS's 128 bytes per cycle against 65 %, before the
//           DMA's own LDS writes -- and a tile's L2 -> LDS bytes per flop are 1 / 256 + 1 / 96 against 2 / 192 (-27 %).
// A ring slot holds BM rows of A and BN rows of W, one 32-element K chunk (64 bytes) each; (BM + BN) / 16 <= 24 DMA instructions per step,
// six per wavefront in both shapes.
template <typename T, int TM, int WN>
__global__ __launch_bounds__(256, 2) void gemm16_dma2_kernel(GdArgs<T> p) {
  typedef typename GdMma<T>::vec vec;
  constexpr int TN = 3, WM = 4 / WN, BM = WM * TM * 32, BN = WN * 96;
  constexpr int kGdSlot = (BM + BN) * 64;
  constexpr int NA = BM / 64;                     // DMA instruction q of a wavefront: q < NA rows of A, else rows of W
  static_assert(BM % 64 == 0 && (BM + BN) / 16 <= 24, "six DMA instructions per wavefront and step");
  constexpr unsigned DEAD = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char gd_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b;
  {  // XCD-aware order: an XCD takes a contiguous run of tiles; N tiles of one row block are neighbours (they share the A rows)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tn = b % p.tiles_n, tm = b / p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = wave / WN, wn = wave - wm * WN;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((gd_lds_ptr_t)gd_smem));
  const unsigned lds_scratch = lds0 + 3 * kGdSlot;
  const int G = p.K >> 5;

  // ---- issue side: A instruction q (q < 4) of a wavefront covers rows (q * 4 + wave) * 16 + (lane >> 2), W instruction q (q < 2) rows
  // (q * 4 + wave) * 16 + (lane >> 2) of the N tile; physical slot lane & 3 receives logical slot (lane & 3) ^ ((row >> 2) & 3)
  // byte offset of output row m's first input row
  auto row_off = [&](int m) __attribute__((always_inline)) -> long {
    const int n = m / p.Wrow, pos = m - n * p.Wrow;
    return (static_cast<long>(n) * p.Hrow * p.Wrow + pos) * p.Cin * 2;
  };
  const long base_off = row_off(m0);
  const long total_b = static_cast<long>((p.M + p.Wrow - 1) / p.Wrow) * p.Hrow * p.Wrow * p.Cin * 2;
  const unsigned long pa = reinterpret_cast<unsigned long>(p.a) + static_cast<unsigned long>(base_off);
  const int rows_a = min(BM, p.M - m0), rows_w = min(BN, p.N - n0);
  const long rec_a = total_b - base_off;
  const gd_i32x4 rs_a = gd_i32x4{static_cast<int>(pa), static_cast<int>(pa >> 32) & 0xFFFF, static_cast<int>(rec_a < 0x7FFFFFFFL ? rec_a : 0x7FFFFFFFL), 0x00020000};
  const unsigned long pw = reinterpret_cast<unsigned long>(p.w + static_cast<long>(n0) * p.K);
  const gd_i32x4 rs_w = gd_i32x4{static_cast<int>(pw), static_cast<int>(pw >> 32) & 0xFFFF, rows_w * p.K * 2, 0x00020000};
  unsigned a_voff[NA], w_voff[6 - NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int row = (q * 4 + wave) * 16 + (lane >> 2), ls = (lane & 3) ^ ((row >> 2) & 3);
    a_voff[q] = row < rows_a ? static_cast<unsigned>(row_off(m0 + row) - base_off + ls * 16) : DEAD;
  }
#pragma unroll
  for (int q = 0; q < 6 - NA; ++q) {
    const int row = (q * 4 + wave) * 16 + (lane >> 2), ls = (lane & 3) ^ ((row >> 2) & 3);
    w_voff[q] = row < rows_w ? static_cast<unsigned>((row * p.K + ls * 8) * 2) : DEAD;     // rows past BN (256 x 96: q = 1, waves 2, 3): past rows_w
  }
  const unsigned tap_stride = static_cast<unsigned>(p.Wrow) * p.Cin * 2u;
  auto issue = [&](int g) __attribute__((always_inline)) {             // K chunk g -> slot g % 3
    const bool live = g < G;
    const unsigned dst = lds0 + (g % 3) * kGdSlot, soff = static_cast<unsigned>(g) * 64u;
    const int ch = g / p.taps, tap = g - ch * p.taps;
    const unsigned soff_a = static_cast<unsigned>(tap) * tap_stride + static_cast<unsigned>(ch) * 64u;
#pragma unroll
    for (int q = 0; q < NA; ++q) gd_dma(live ? dst + (q * 4 + wave) * 1024 : lds_scratch, live ? a_voff[q] : DEAD, rs_a, soff_a);
#pragma unroll
    for (int q = 0; q < 6 - NA; ++q) {
      const bool in_tile = (q * 4 + wave) * 16 < BN;      // 256 x 96: the second W instruction of wavefronts 2, 3 has no rows
      gd_dma(live && in_tile ? dst + BM * 64 + (q * 4 + wave) * 1024 : lds_scratch, live && in_tile ? w_voff[q] : DEAD, rs_w, soff);
    }
  };

  // ---- fragment addressing: lane -> row lp of its 32-row MFMA tile, k half kh; logical slot of (kk, kh) = 2 kk + kh
  const int lp = lane & 31, kh = lane >> 5;
  int a_off[TM][2], b_off[TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * (TM * 32) + i * 32 + lp;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_off[i][kk] = row * 64 + (((kk * 2 + kh) ^ ((row >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * 96 + j * 32 + lp;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_off[j][kk] = BM * 64 + row * 64 + (((kk * 2 + kh) ^ ((row >> 2) & 3)) << 4);
  }

  issue(0);
  issue(1);
  gd_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int g = 0; g < G; ++g) {
    // this wavefront's pieces of chunk g have landed (the six of chunk g + 1 may be in flight) and its LDS reads of step g - 1 have
    // returned (see conv16_dma.hip); after the barrier everybody's have, and slot (g + 2) % 3 is free
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(g + 2);
    const unsigned char* S = gd_smem + (g % 3) * kGdSlot;
    vec fa[2][TM], fb[2][TN];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kk][i] = *reinterpret_cast<const vec*>(S + a_off[i][kk]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kk][j] = *reinterpret_cast<const vec*>(S + b_off[j][kk]);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = GdMma<T>::run(fb[kk][j], fa[kk][i], acc[i][j]);   // D^T: rows = columns of out, cols = rows of out
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue through LDS: per 32-row tile a wavefront parks [32 rows][96 + 4] fp32 in its own 12.8 KB and reads back
  // (row, column octet) items
  float* stage = reinterpret_cast<float*>(gd_smem) + wave * (32 * 100);
  const T* __restrict__ resid = p.residual;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // the residual octets of this row tile are requested before the sums are parked (a round trip under the LDS traffic)
    uint4 rraw[6];
    if (resid) {
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int item = it * 64 + lane;
        const int rr = item / 12, oc = item - rr * 12;
        const int n = n0 + wn * 96 + oc * 8, m = m0 + wm * (TM * 32) + i * 32 + rr;
        rraw[it] = (m < p.M && n < p.N) ? *reinterpret_cast<const uint4*>(resid + static_cast<long>(m) * p.N + n) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        st4(stage + lp * 100 + j * 32 + q * 8 + kh * 4, make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int item = it * 64 + lane;            // 32 rows x 12 octets
      const int rr = item / 12, oc = item - rr * 12;
      const int n = n0 + wn * 96 + oc * 8, m = m0 + wm * (TM * 32) + i * 32 + rr;
      const float4 s0 = ld4(stage + rr * 100 + oc * 8), s1 = ld4(stage + rr * 100 + oc * 8 + 4);
      if (m >= p.M || n >= p.N) continue;
      float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      if (p.bias) {
        const float4 t0 = ld4(p.bias + n), t1 = ld4(p.bias + n + 4);
        v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
      }
      if (p.scale) {
        const float4 c0 = ld4(p.scale + n), c1 = ld4(p.scale + n + 4), h0 = ld4(p.shift + n), h1 = ld4(p.shift + n + 4);
        v[0] = v[0] * c0.x + h0.x; v[1] = v[1] * c0.y + h0.y; v[2] = v[2] * c0.z + h0.z; v[3] = v[3] * c0.w + h0.w;
        v[4] = v[4] * c1.x + h1.x; v[5] = v[5] * c1.y + h1.y; v[6] = v[6] * c1.z + h1.z; v[7] = v[7] * c1.w + h1.w;
      }
      if (p.rowvec) {
        const float* rv = p.rowvec + static_cast<long>(m / p.rows_per_img) * p.rowvec_ld + n;
        const float4 t0 = ld4(rv), t1 = ld4(rv + 4);
        v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
      }
      if (p.act == DIFFSAL_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (p.act == DIFFSAL_ACT_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
      } else if (p.act == DIFFSAL_ACT_SIGMOID) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sigmoidf_(v[e]);
      }
      const long o = static_cast<long>(m) * p.N + n;
      if (resid) {
        const f8v t = ld8(reinterpret_cast<const T*>(&rraw[it]));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += t.v[e];
      }
      if (p.out_f32) {
        float* of = static_cast<float*>(p.out) + o;
        st4(of, make_float4(v[0], v[1], v[2], v[3]));
        st4(of + 4, make_float4(v[4], v[5], v[6], v[7]));
      } else {
        f8v ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov.v[e] = v[e];
        st8(static_cast<T*>(p.out) + o, ov);
      }
    }
  }
}

// 1 if launched, 0 if the shape is not handled here (the caller goes on to the other kernels), < 0 on error
int try_gemm16_dma2(const diffsal_conv_desc* d, const void* a, const void* w, const float* bias, const float* scale, const float* shift,
                    const float* rowvec, int rowvec_ld, const void* residual, void* out, hipStream_t s, bool out_f32) {
  if (tune(TUNE_NO_STREAM16) == 1 || tune(TUNE_IGEMM16_CFG) >= 0) return 0;
  const int forced = tune(TUNE_GEMM_DMA16);         // 0: off, 3: on every shape it can run (tests); 1 / 2 force gemm_dma.hip's tiles
  if (forced == 0 || forced == 1 || forced == 2) return 0;     // 3 / 4: this kernel's 256 x 96 / 192 x 192 tile on every shape it can run
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const int K = d->KH * d->KW * d->Cin, N = d->Cout;
  if (d->dtype == DIFFSAL_F32 || d->KW != 1 || d->stride_w != 1 || d->pad_t != 0 || d->pad_l != 0 || d->dil_h != 1 || d->Wo != d->W) return 0;
  const bool plain = d->KH == 1 && d->stride_h == 1 && d->Ho == d->H;
  const bool rowform = d->KH > 1 && d->Ho == 1 && d->KH <= d->H;           // one output frame: the stride does not matter
  if (!plain && !rowform) return 0;
  const long in_bytes = static_cast<long>(d->N) * d->H * d->W * d->Cin * 2;
  if (K % 32 != 0 || d->Cin % 32 != 0 || K < 64 || N % 8 != 0 || M <= 0 || M >= (1L << 31) || 256L * K * 2 >= (1L << 31) || 96L * K * 2 >= (1L << 31) ||
      (rowform && (static_cast<long>(d->H) * d->W * d->Cin * 2 * 2 >= (1L << 31) || in_bytes >= (1L << 40))))
    return 0;
  if (!aligned16(a) || !aligned16(w) || !aligned16(out) || !aligned16(bias) || !aligned16(scale) || !aligned16(shift) || !aligned16(rowvec) ||
      !aligned16(residual) || (rowvec && rowvec_ld % 4 != 0) || ((scale == nullptr) != (shift == nullptr)))
    return 0;
  if (out_f32 && residual) return 0;
  const long tiles = ((M + 255) / 256) * ((N + 95) / 96);
  if (tiles >= (1L << 31)) return 0;
  // 192 x 192 tiles (DIFFSAL_GEMM_DMA16 = 4 forces them, 3 the 256 x 96 ones): from 400 tiles on (measured at 4 .. 64 clips: faster than
  // both the 256 x 96 tiles and gemm_dma.hip's 96 x 96 ones from ~500 tiles, slower at ~250), unless a quarter of the last tile column
  // lies past N -- or any of it on a short K walk (N = 864, K = 192: 70 -> 78 us; K = 768: 1006 -> 885)
  const long tiles_sq = ((M + 191) / 192) * ((N + 191) / 192);
  const int n_waste = ((N + 191) / 192) * 192 - N;
  const bool square = forced == 4 || (forced != 3 && tiles_sq >= 400 && (n_waste == 0 || (n_waste <= N / 4 && K >= 384)));
  // 256 x 96: from a chip's worth of tiles (two workgroups per CU) on; below that gemm_dma.hip's 96 x 96 tiles (and their K split) fill
  // the chip better
  if (!square && forced != 3 && tiles < 512) return 0;
  const int bm = square ? 192 : 256, bn = square ? 192 : 96;
  const size_t lds = 3 * static_cast<size_t>(bm + bn) * 64 + 1024;
#define GD_LAUNCH(T, TM_, WN_)                                                                                              \
  do {                                                                                                                      \
    GdArgs<T> g{static_cast<const T*>(a), static_cast<const T*>(w), bias, scale, shift, rowvec, static_cast<const T*>(residual), out, \
                static_cast<int>(M), N, K, d->act, rowvec_ld, d->Ho * d->Wo, out_f32 ? 1 : 0, static_cast<int>((M + bm - 1) / bm), (N + bn - 1) / bn, \
                rowform ? d->KH : 1, rowform ? d->W : static_cast<int>(M), rowform ? d->H : 1, d->Cin};                        \
    DS_RAISE_DYNAMIC_LDS((gemm16_dma2_kernel<T, TM_, WN_>), 160 * 1024);                                                    \
    hipLaunchKernelGGL((gemm16_dma2_kernel<T, TM_, WN_>), dim3(static_cast<unsigned>(g.tiles_m) * g.tiles_n), dim3(256), lds, s, g); \
  } while (0)
  if (square) {
    if (d->dtype == DIFFSAL_BF16) GD_LAUNCH(__bf16, 3, 2); else GD_LAUNCH(_Float16, 3, 2);
  } else {
    if (d->dtype == DIFFSAL_BF16) GD_LAUNCH(__bf16, 2, 1); else GD_LAUNCH(_Float16, 2, 1);
  }
#undef GD_LAUNCH
  note_kernel("gemm16_dma2_kernel<%s> [%dx%d tile, LDS-DMA, 2 workgroups per CU]", d->dtype == DIFFSAL_BF16 ? "__bf16" : "_Float16", bm, bn);
  const int rc = check_launch("diffsal_conv_igemm(16-bit DMA, 2 per CU)");
  return rc == DIFFSAL_OK ? 1 : rc;
}

}  // namespace diffsal
