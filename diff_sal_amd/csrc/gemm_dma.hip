// Plain [M, K] x [N, K]^T products in exact fp32 (1x1 convolutions, Linear layers, the tap products of UpEmbed / mt_proj) with
// both operands staged by LDS-DMA -- the main loop owns no staging registers and issues no LDS writes.
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )        R/models/saliency_decoder/attention.py:97-111 (proj_q/k/v, proj),
//                                                          common_block.py:125-147 (Mlp fc1 / fc2), transformer.py:150-157
//
// Why another kernel beside igemm_linear_kernel: that one stages through registers (global -> VGPR -> ds_write), and its
// 64 x 64 tiles (the shape that fills the chip on the decoder's token counts) pull 16 B/clk/CU through the vector memory
// path for every 4096 matrix cycles -- the matrix pipe of the token GEMMs is 0.59 busy (profiles/r03_pmc_mfma_busy_fp32.md).
//
//  * Tiles are 96 wide in both directions (96 x 96, 96 x 192): every channel count of the network is a multiple of 96 and
//    the token counts are 3024 * 4^i, so the tile grid has no partly filled N tiles and M = 3024 x N = 768 is exactly 256
//    tiles -- one per CU -- where 64 x 64 tiles leave the CUs 2.25 tiles each (the busiest a third over the average) and
//    128 x 128 tiles fill 144 of 256 CUs.
//  * A wave owns 48 x 48 (or 48 x 96) outputs as 16 x 16 blocks of v_mfma_f32_16x16x4_f32 (bit-for-bit an fmaf chain, the
//    fp32 vector rate; 9 / 18 independent accumulators).
//  * A K slice is 32 floats: one 128-byte run per row.  An LDS stage is [BM + BN rows][128 B] with the eight 16-byte slots of
//    a row XOR-swizzled by (row >> 1) & 7, so that the sixteen lanes of a ds_read_b128 pass (16 consecutive rows, one slot)
//    cover all 64 banks.  A DMA instruction (buffer_load_dwordx4 ... lds) fills 8 rows = 1 KiB in lane order; the swizzle is
//    applied on the SOURCE address (the lane that fills physical slot p of row r fetches logical slot p ^ ((r >> 1) & 7)).
//    Rows past M / N are cut by the buffer descriptor's range check (the descriptor is rebuilt per tile: base = the tile's
//    first row, num_records = its valid rows), the DMA then writes zeros.
//  * STAGES-deep ring, the DMA of slice g + STAGES - 1 is issued while slice g is multiplied; ONE barrier per slice, placed
//    between the two halves of the slice's MFMAs (the fragments of the second half are already in registers), counted vmcnt.
//  * Persistent: a workgroup walks a strided list of tiles and the DMA ring runs on into the next tile; epilogue (bias, BN
//    affine, per-image vector, activation, residual) straight from the accumulators: a lane holds four consecutive channels
//    of one output row (weights are the MFMA "A" operand), 16-byte stores.
//
// The summation order inside a 32-wide slice differs from igemm_kernel's (k sets {j, j+4, j+8, j+12} per MFMA instead of pairs
// {j, j+4}), so results differ from that kernel by fp32 rounding (~1e-7 relative); both are exact-fp32 fmaf chains.
#include <type_traits>

#include "common.h"

namespace diffsal {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct DmaGemmArgs {
  const void* a;          // storage type T of the launch (float, bf16, f16): activations, weights, residual, output
  const void* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const void* residual;
  void* out;
  int M, N, K;
  int act, rowvec_ld, rows_per_img;
  int n_tiles_m, n_tiles_n, n_tiles;
  int xcd_order;   // tiles walked so that one XCD owns whole M-tile rows (large launches, grid % 8 == 0)
  // convolution mode (CONV): row m = (image, oy, ox) of a channels-last input; K slice kt = (32-channel chunk, tap), taps fastest
  int H, W, Cin, Ho, Wo, KW, taps, stride_h, stride_w, pad_t, pad_l, dil_h, dil_w;
  unsigned in_bytes;
  // split-K: unit u = (split, tile); a unit covers K slices [split * kt_per_unit, ...) and leaves raw sums in partial[split]
  int splits, kt_per_unit;
  float* partial;
  // batched plain products (one launch, `batch` independent problems of ONE shape: the 36 position products of the F(4x4, 3x3)
  // Winograd path): unit v = (batch index, tile), operands / output of problem b at a + b a_bs, w + b w_bs, out + b o_bs (elements)
  int batch;
  long a_bs, w_bs, o_bs;
  // a second run of `batch2` problems of the SAME shape appended to the batch: operands a2 + i a2_bs, ONE weight matrix w2, output
  // out2 + i o2_bs (a plain product whose rows are cut into batch2 pieces of M rows: the ResnetBlock's 1x1 shortcut rides in the
  // launch of its first convolution's position products, R/models/saliency_decoder/sal_unet.py:123-142)
  int out_f32;     // 16-bit storage launches: the output (and nothing else) is fp32 -- sums leave without a rounding to 16 bits
  int batch2;
  const void* a2;
  const void* w2;
  void* out2;
  long a2_bs, o2_bs;
#ifdef DIFFSAL_DEV_STAMPS
  unsigned long long* stamps;   // development build only: 32 time stamps per workgroup (diffsal_set_dma_stamps)
#endif
};

#ifdef DIFFSAL_DEV_STAMPS
static unsigned long long* g_dma_stamps = nullptr;
static size_t g_dma_stamp_bytes = 0;
#endif

// One LDS-DMA piece: 64 lanes x 16 bytes from the buffer `rsrc` at voff + soff into LDS at lds_addr + 16 lane.  Inline assembly on
// purpose: hipcc orders every later LDS read behind a DMA it knows about (s_waitcnt vmcnt(0) in front of the fragment reads at a
// loop header, whatever object they read), which serialises the ring.  Here the compiler sees neither the LDS write nor the
// vmcnt event; the kernel places its own counted waits.  (Its counted waits for ordinary loads stay safe: vmcnt retires in
// order, so DMAs it does not know about only make such a wait stricter.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_piece(unsigned lds_addr, unsigned voff, i32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}

typedef __bf16 dma_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dma_f16x8 __attribute__((ext_vector_type(8)));
// one 16 x 16 x 32 product of two 16-byte fragments (8 elements of k per lane), fp32 accumulation
__device__ __forceinline__ f32x4v mma16(const float4& a, const float4& b, f32x4v c, const bf16_t*) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dma_bf16x8, a), __builtin_bit_cast(dma_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v mma16(const float4& a, const float4& b, f32x4v c, const f16_t*) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dma_f16x8, a), __builtin_bit_cast(dma_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v mma16(const float4&, const float4&, f32x4v c, const float*) { return c; }   // never called

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Grouped launch: up to four independent problems (own operands, shapes, K splits) share ONE persistent grid; unit v of the
// launch belongs to problem idx = #(unit_end[j] <= v).  The host orders the problems by decreasing K slices per unit, so the
// long units start first (the four ReduceTemp products of a step: 120 / 60 / 30 / 15 slices per tile).
constexpr int kMaxGroup = 4;
struct DmaGroupArgs {
  int n, total;
  int unit_end[kMaxGroup];
  DmaGemmArgs prob[kMaxGroup];
};

// Field-by-field scalar selects (no address of the by-value kernel argument is taken: that would move it to scratch memory).
__device__ __forceinline__ DmaGemmArgs pick_problem(const DmaGroupArgs& g, int idx) {
  DmaGemmArgs r;
#define DS_PICK(f) r.f = idx == 0 ? g.prob[0].f : idx == 1 ? g.prob[1].f : idx == 2 ? g.prob[2].f : g.prob[3].f
  DS_PICK(a); DS_PICK(w); DS_PICK(bias); DS_PICK(scale); DS_PICK(shift); DS_PICK(rowvec); DS_PICK(residual); DS_PICK(out);
  DS_PICK(M); DS_PICK(N); DS_PICK(K); DS_PICK(act); DS_PICK(rowvec_ld); DS_PICK(rows_per_img);
  DS_PICK(n_tiles_m); DS_PICK(n_tiles_n); DS_PICK(n_tiles); DS_PICK(xcd_order);
  DS_PICK(H); DS_PICK(W); DS_PICK(Cin); DS_PICK(Ho); DS_PICK(Wo); DS_PICK(KW); DS_PICK(taps); DS_PICK(stride_h); DS_PICK(stride_w);
  DS_PICK(pad_t); DS_PICK(pad_l); DS_PICK(dil_h); DS_PICK(dil_w); DS_PICK(in_bytes); DS_PICK(splits); DS_PICK(kt_per_unit);
  DS_PICK(partial);
#undef DS_PICK
  return r;
}

// T: storage type.  fp32: a 128-byte K slice is 32 floats and every 16-byte fragment feeds four v_mfma_f32_16x16x4_f32; bf16 / f16:
// 64 elements per slice, a fragment is the 8-element operand of ONE v_mfma_f32_16x16x32 (plain products only: the packed
// convolution weights order K by 32-channel chunks).  Byte geometry -- rows, slots, swizzle, DMA pieces -- is the same.
template <typename T, int TMB, int TNB, int STAGES, int OCC, bool CONV, bool GROUPED>
__global__ __launch_bounds__(256, OCC) void gemm_dma_kernel(std::conditional_t<GROUPED, DmaGroupArgs, DmaGemmArgs> g) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int ESZ = sizeof(T);
  static_assert(F32 || !CONV, "16-bit storage: plain products only");
  constexpr int BM = 32 * TMB, BN = 32 * TNB;
  constexpr int APW = BM / 32, BPW = BN / 32;          // 1 KiB DMA pieces (8 rows) per wave and slice
  constexpr int PPW = APW + BPW;
  constexpr int STAGE_F = (BM + BN) * 32;              // floats per stage
  constexpr int P = STAGES - 1;                        // slices in flight ahead of the one being multiplied
  constexpr int WAIT_N = (P - 2 > 0 ? P - 2 : 0) * PPW;
  static_assert(STAGES >= 3 && WAIT_N < 64, "ring depth");
  __shared__ __attribute__((aligned(16))) float smem[STAGES * STAGE_F];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // pi: the problem of the unit being ISSUED (the DMA ring runs up to P slices ahead, into the next unit), pc: of the unit being
  // multiplied.  One problem per launch: both are the kernel argument.
  DmaGemmArgs pi, pc;
  if constexpr (GROUPED) { pi = pick_problem(g, 0); pc = pi; } else { pi = g; pc = g; }

  // ---- unit walk (same scheme as igemm_linear_kernel): v = blockIdx.x + i * gridDim.x
  int n_virtual;
  if constexpr (GROUPED) {
    n_virtual = g.total;
  } else {
    n_virtual = g.n_tiles * g.splits * (g.batch + g.batch2);         // xcd_order only without split-K
    if (g.xcd_order) {
      // rows = (problem of the batch, M tile): an XCD owns whole rows, so that a row's A block is fetched into ONE L2
      const int x = blockIdx.x & 7, rows = g.n_tiles_m * (g.batch + g.batch2);
      n_virtual = 8 * g.n_tiles_n * (rows > x ? (rows - x + 7) >> 3 : 0);
    }
  }
  // problem of unit v and the unit's index inside it
  auto locate = [&](int v, int& idx, int& lv) __attribute__((always_inline)) {
    if constexpr (GROUPED) {
      idx = (v >= g.unit_end[0]) + (v >= g.unit_end[1]) + (v >= g.unit_end[2]);
      lv = v - (idx == 0 ? 0 : idx == 1 ? g.unit_end[0] : idx == 2 ? g.unit_end[1] : g.unit_end[2]);
    } else {
      idx = 0;
      lv = v;
    }
  };
  auto tile_mn = [&](const DmaGemmArgs& p, int v, int& tmi, int& tni) __attribute__((always_inline)) {
    if (!GROUPED && p.xcd_order && p.batch + p.batch2 == 1) {
      const int q = v >> 3, ml = q / p.n_tiles_n;
      tni = q - ml * p.n_tiles_n;
      tmi = ml * 8 + (v & 7);
    } else {
      const int t = p.splits > 1 ? v % p.n_tiles : v;
      tmi = t / p.n_tiles_n;
      tni = t - tmi * p.n_tiles_n;
    }
  };
  auto split_of = [&](const DmaGemmArgs& p, int v) __attribute__((always_inline)) { return p.splits > 1 ? v / p.n_tiles : 0; };
  const int bid = blockIdx.x, gsz = gridDim.x;
  const int my_tiles = n_virtual > bid ? (n_virtual - bid + gsz - 1) / gsz : 0;
  if (my_tiles == 0) return;
#ifdef DIFFSAL_DEV_STAMPS
  auto stamp = [&](int k) {
    if constexpr (!GROUPED) { if (g.stamps && tid == 0 && k < 32) g.stamps[blockIdx.x * 32 + k] = wall_clock64(); }
  };
#else
  auto stamp = [](int) {};
#endif
  stamp(0);

  // ---- issue side.  Plain products: one per-lane byte offset serves every piece (pieces of a wave are 32 rows apart: the
  // swizzle term ((row >> 1) & 7) = (4 wave + (lane >> 4)) & 7 does not depend on the piece); the descriptor of A is rebuilt
  // per tile (base = first row, records = valid rows).  Convolutions: a row's pixel origin (a_pix) and the bit mask of its
  // taps that lie inside the image (a_valid) are rebuilt per tile, the tap displacement is added per slice, and a tap outside
  // the image -- or a row past M -- becomes offset 0xFFFFFFFF: out of range, the DMA writes zeros (the convolution's padding).
  constexpr int VQ = APW > BPW ? APW : BPW;
  unsigned voff[VQ];
  const int r0 = 8 * wave + (lane >> 3);
  const int slot = (lane & 7) ^ ((r0 >> 1) & 7);
  unsigned a_pix[APW], a_valid[APW];
  int iss_v = bid, iss_kt = 0, iss_lv = 0;
  int iss_tap = 0, iss_ky = 0, iss_kx = 0, iss_chunk = 0;   // CONV: position of the slice being issued
  unsigned iss_kofs = 0;                                    // byte offset of the slice inside a weight row
  i32x4 rs_a, rs_b;
  auto descriptors = [&](int v) __attribute__((always_inline)) {
    // past the last unit: zero records, every lane is out of range (the DMA then writes zeros into a free stage and touches
    // no memory) -- the ring keeps issuing so that the counted waits stay exact
    const bool live = v < n_virtual;
    if constexpr (GROUPED) {
      int idx;
      locate(live ? v : 0, idx, iss_lv);
      pi = pick_problem(g, idx);
    } else {
      iss_lv = v;
    }
    long ib_a = 0, ib_w = 0;       // batched: element offsets of the unit's problem
    const void* base_a = pi.a;
    const void* base_w = pi.w;
    if constexpr (!GROUPED) {
      if (g.batch + g.batch2 > 1) {
        const int per = g.n_tiles * g.splits, vv = live ? v : 0;
        int ib;
        if (g.xcd_order) {           // v = 8 (row-of-eight ml, N tile) + XCD: row = ml * 8 + XCD = ib * n_tiles_m + M tile
          const int q = vv >> 3, ml = q / g.n_tiles_n, row = ml * 8 + (vv & 7);
          ib = row / g.n_tiles_m;
          iss_lv = (row - ib * g.n_tiles_m) * g.n_tiles_n + (q - ml * g.n_tiles_n);
        } else {
          ib = vv / per;
          iss_lv = vv - ib * per;
        }
        if (ib < g.batch) {
          ib_a = ib * g.a_bs;
          ib_w = ib * g.w_bs;
        } else {
          ib_a = (ib - g.batch) * g.a2_bs;
          base_a = g.a2;
          base_w = g.w2;
        }
      }
    }
    const int K = pi.K;
#pragma unroll
    for (int q = 0; q < VQ; ++q) voff[q] = static_cast<unsigned>((r0 + 32 * q) * K * ESZ + slot * 16);
    int tmi, tni;
    tile_mn(pi, iss_lv, tmi, tni);
    const int m0 = tmi * BM, n0 = tni * BN;
    const int rows_b = live ? min(BN, pi.N - n0) : 0;
    const unsigned long pb = reinterpret_cast<unsigned long>(static_cast<const T*>(base_w) + ib_w + static_cast<long>(n0) * K);
    rs_b = i32x4{static_cast<int>(pb), static_cast<int>(pb >> 32) & 0xFFFF, rows_b * K * ESZ, 0x00020000};
    const int kt0 = split_of(pi, iss_lv) * pi.kt_per_unit;
    iss_kofs = static_cast<unsigned>(kt0) * 128u;
    if constexpr (!CONV) {
      const int rows_a = live ? min(BM, pi.M - m0) : 0;
      const unsigned long pa = reinterpret_cast<unsigned long>(static_cast<const T*>(base_a) + ib_a + static_cast<long>(m0) * K);
      rs_a = i32x4{static_cast<int>(pa), static_cast<int>(pa >> 32) & 0xFFFF, rows_a * K * ESZ, 0x00020000};
    } else {
      const unsigned long pa = reinterpret_cast<unsigned long>(pi.a);
      rs_a = i32x4{static_cast<int>(pa), static_cast<int>(pa >> 32) & 0xFFFF, static_cast<int>(live ? pi.in_bytes : 0u), 0x00020000};
      iss_chunk = kt0 / pi.taps;
      iss_tap = kt0 - iss_chunk * pi.taps;
      iss_ky = iss_tap / pi.KW;
      iss_kx = iss_tap - iss_ky * pi.KW;
      const int HoWo = pi.Ho * pi.Wo;
#pragma unroll
      for (int q = 0; q < APW; ++q) {
        const int m = m0 + r0 + 32 * q;
        const int mc = m < pi.M ? m : pi.M - 1;
        const int n = mc / HoWo, rem = mc - n * HoWo;
        const int oy = rem / pi.Wo, ox = rem - oy * pi.Wo;
        const int iy0 = oy * pi.stride_h - pi.pad_t, ix0 = ox * pi.stride_w - pi.pad_l;
        a_pix[q] = static_cast<unsigned>(((n * pi.H + iy0) * pi.W + ix0) * pi.Cin + slot * 4) * 4u;   // wraps for taps outside: masked
        unsigned bits = 0;
        if (m < pi.M)
          for (int t = 0; t < pi.taps; ++t) {
            const int ky = t / pi.KW, kx = t - ky * pi.KW;
            const int iy = iy0 + ky * pi.dil_h, ix = ix0 + kx * pi.dil_w;
            bits |= ((iy >= 0) & (iy < pi.H) & (ix >= 0) & (ix < pi.W)) ? (1u << t) : 0u;
          }
        a_valid[q] = bits;
      }
    }
  };
  descriptors(iss_v);
  const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)smem)) + wave * 1024u;
  // piece q of the slice being issued: q < APW rows of A, else rows of W
  auto issue_piece = [&](int stage, int q) __attribute__((always_inline)) {
    const unsigned st = lds_base + static_cast<unsigned>(stage * STAGE_F * 4);
    if (q < APW) {
      if constexpr (!CONV) {
        dma_piece(st + q * 4096u, voff[q], rs_a, iss_kofs);
      } else {
        const unsigned delta = static_cast<unsigned>((iss_ky * pi.dil_h * pi.W + iss_kx * pi.dil_w) * pi.Cin + iss_chunk * 32) * 4u;
        const unsigned oob = ((a_valid[q] >> iss_tap) & 1u) - 1u;      // 0 inside the image, else all ones
        dma_piece(st + q * 4096u, (a_pix[q] + delta) | oob, rs_a, 0u);
      }
    } else {
      dma_piece(st + (BM * 32 + (q - APW) * 1024) * 4u, voff[q - APW], rs_b, iss_kofs);
    }
  };
  // A unit is a multiple of STAGES slices and starts on ring stage 0, so the slice issued while stage S is multiplied (P ahead)
  // is the LAST of its unit only for S == 0: the unit-boundary test and the descriptor rebuild behind it exist once in the ring
  // (and not at all in the prologue: P < STAGES <= kt_per_unit), not once per stage -- a third of the code size for the
  // convolution form, whose descriptor rebuild is long.
  auto issue_advance = [&](auto may_end_unit) __attribute__((always_inline)) {
    iss_kofs += 128u;
    if constexpr (CONV) {
      if (++iss_kx == pi.KW) { iss_kx = 0; ++iss_ky; }
      if (++iss_tap == pi.taps) { iss_tap = 0; iss_ky = 0; iss_kx = 0; ++iss_chunk; }
    }
    ++iss_kt;
    if constexpr (decltype(may_end_unit)::value) {
      if (iss_kt == pi.kt_per_unit) {
        iss_kt = 0;
        iss_v += gsz;
        descriptors(iss_v);
      }
    }
  };
  auto issue_slice = [&](int stage) __attribute__((always_inline)) {      // prologue only
#pragma unroll
    for (int q = 0; q < PPW; ++q) issue_piece(stage, q);
    issue_advance(std::false_type{});
  };

  // ---- compute side: fragment addresses (floats inside a stage) for the two 16-wide halves of a slice
  const int q4 = lane >> 4, r16 = lane & 15;
  int a_off[TMB][2], b_off[TNB][2];
#pragma unroll
  for (int i = 0; i < TMB; ++i) {
    const int row = wm * TMB * 16 + i * 16 + r16;
#pragma unroll
    for (int h = 0; h < 2; ++h) a_off[i][h] = row * 32 + (((q4 + 4 * h) ^ ((row >> 1) & 7)) << 2);
  }
#pragma unroll
  for (int j = 0; j < TNB; ++j) {
    const int row = wn * TNB * 16 + j * 16 + r16;
#pragma unroll
    for (int h = 0; h < 2; ++h) b_off[j][h] = BM * 32 + row * 32 + (((q4 + 4 * h) ^ ((row >> 1) & 7)) << 2);
  }
  f32x4v acc[TMB][TNB];
#pragma unroll
  for (int i = 0; i < TMB; ++i)
#pragma unroll
    for (int j = 0; j < TNB; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  float4 fa[2][TMB], fb[2][TNB];
  auto load_frag = [&](const float* st, int h, int set, int f) __attribute__((always_inline)) {   // f < TMB: A block f, else W block
    if (f < TMB) fa[set][f] = ld4(st + a_off[f][h]);
    else fb[set][f - TMB] = ld4(st + b_off[f - TMB][h]);
  };
  auto load_frags = [&](const float* st, int h, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < TMB + TNB; ++f) load_frag(st, h, set, f);
  };
  // MFMA idx of a half slice.  fp32: idx = (c * TMB + i) * TNB + j, component c of the 16-byte fragments (four k per MFMA);
  // 16-bit: idx = i * TNB + j, the whole fragment (eight k per lane, 32 per MFMA)
  constexpr int NM = (F32 ? 4 : 1) * TMB * TNB;
  auto mfma_one = [&](int set, int idx) __attribute__((always_inline)) {
    const int c = idx / (TMB * TNB), i = (idx / TNB) % TMB, j = idx % TNB;
    if constexpr (F32) {
      const float av = c == 0 ? fa[set][i].x : c == 1 ? fa[set][i].y : c == 2 ? fa[set][i].z : fa[set][i].w;
      const float bv = c == 0 ? fb[set][j].x : c == 1 ? fb[set][j].y : c == 2 ? fb[set][j].z : fb[set][j].w;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);   // D^T: rows = channels, cols = output rows
    } else {
      acc[i][j] = mma16(fb[set][j], fa[set][i], acc[i][j], static_cast<const T*>(nullptr));
    }
  };
  auto do_mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int idx = 0; idx < NM; ++idx) mfma_one(set, idx);
  };

  int cmp_v = bid, cmp_lv = bid;
  long cmp_ob = 0;                 // batched: element offset of the output of the unit being multiplied
  void* cmp_out = pc.out;          // and its base (the appended run of a batched launch writes to out2)
  // Epilogue operands (per-channel vectors, residual quads) are requested at the start of the tile's LAST pass over the ring and
  // wait in registers: requested in the epilogue they would be a dependent round trip with the matrix pipe idle, and -- vmcnt
  // retires in order -- their wait would also drain the DMA ring.
  // (the residual quads only while they fit: 9 of them on the 96 x 96 tile; the 192 x 192 tile of 16-bit storage would hold 36 --
  // 144 registers -- and reads them in its epilogue instead)
  constexpr bool PRE_RES = TMB * TNB <= 12;
  float4 rres[PRE_RES ? TMB : 1][PRE_RES ? TNB : 1], rbias[PRE_RES ? TNB : 1], rscale[PRE_RES ? TNB : 1], rshift[PRE_RES ? TNB : 1];
  auto fetch_epilogue_operands = [&]() __attribute__((always_inline)) {
    if constexpr (!PRE_RES) return;
    int tmi, tni;
    tile_mn(pc, cmp_lv, tmi, tni);
    const int m0 = tmi * BM, n0 = tni * BN;
#pragma unroll
    for (int j = 0; j < TNB; ++j) {
      const int n = n0 + wn * TNB * 16 + j * 16 + 4 * q4;
      const int nc = n < pc.N ? n : 0;
      if constexpr (PRE_RES) {
        if (pc.bias) rbias[j] = ld4(pc.bias + nc);
        if (pc.scale) { rscale[j] = ld4(pc.scale + nc); rshift[j] = ld4(pc.shift + nc); }
        if (pc.residual) {
#pragma unroll
          for (int i = 0; i < TMB; ++i) {
            const int m = m0 + wm * TMB * 16 + i * 16 + r16;
            const bool ok = m < pc.M && n < pc.N;
            rres[i][j] = ld4(static_cast<const T*>(pc.residual) + (ok ? static_cast<long>(m) * pc.N + n : 0));
          }
        }
      }
    }
  };
  // act_c: the activation as a compile-time constant (one straight-line copy of the loop per activation, chosen by ONE scalar
  // branch per tile: with the choice inside the loop every 16 x 16 block re-tested it)
  auto epilogue = [&](auto act_c) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value;
    int tmi, tni;
    tile_mn(pc, cmp_lv, tmi, tni);
    const int mb = tmi * BM + wm * TMB * 16 + r16, nb = tni * BN + wn * TNB * 16 + 4 * q4;
    const long o0 = static_cast<long>(mb) * pc.N + nb;
    T* __restrict__ ob = static_cast<T*>(cmp_out) + cmp_ob + o0;
    const float* __restrict__ rvb = pc.rowvec;
#pragma unroll
    for (int j = 0; j < TNB; ++j) {
      const bool n_ok = nb + j * 16 < pc.N;
      float4 qb, qs, qh;
      if constexpr (PRE_RES) {
        qb = rbias[j]; qs = rscale[j]; qh = rshift[j];
      } else {
        const int nc = n_ok ? nb + j * 16 : 0;
        if (pc.bias) qb = ld4(pc.bias + nc);
        if (pc.scale) { qs = ld4(pc.scale + nc); qh = ld4(pc.shift + nc); }
      }
#pragma unroll
      for (int i = 0; i < TMB; ++i) {
        const int m = mb + i * 16;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (pc.bias) { v[0] += qb.x; v[1] += qb.y; v[2] += qb.z; v[3] += qb.w; }
        if (pc.scale) {
          v[0] = v[0] * qs.x + qh.x; v[1] = v[1] * qs.y + qh.y;
          v[2] = v[2] * qs.z + qh.z; v[3] = v[3] * qs.w + qh.w;
        }
        if (rvb) {
          const int mc = m < pc.M ? m : pc.M - 1;
          const float4 t = ld4(rvb + static_cast<long>(mc / pc.rows_per_img) * pc.rowvec_ld + (n_ok ? nb + j * 16 : 0));
          v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if constexpr (ACT == DIFFSAL_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if constexpr (ACT == DIFFSAL_ACT_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if constexpr (ACT == DIFFSAL_ACT_SIGMOID) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = sigmoidf_(v[e]);
        }
        float4 rq = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PRE_RES) {
          rq = rres[i][j];
        } else {
          if (pc.residual && m < pc.M && n_ok) rq = ld4(static_cast<const T*>(pc.residual) + o0 + (static_cast<long>(i) * 16 * pc.N + j * 16));
        }
        if constexpr (ACT == DIFFSAL_ACT_GELU_GRAD) {
          const float4 t = rq;
          v[0] *= gelu_erf_grad(t.x); v[1] *= gelu_erf_grad(t.y); v[2] *= gelu_erf_grad(t.z); v[3] *= gelu_erf_grad(t.w);
        } else {
          if (pc.residual) { const float4 t = rq; v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
        }
        if (m < pc.M && n_ok) {
          if (!F32 && pc.out_f32)
            st4(static_cast<float*>(cmp_out) + cmp_ob + o0 + (static_cast<long>(i) * 16 * pc.N + j * 16), make_float4(v[0], v[1], v[2], v[3]));
          else
            st4(ob + (static_cast<long>(i) * 16 * pc.N + j * 16), make_float4(v[0], v[1], v[2], v[3]));
        }
      }
    }
  };
  auto touch = [](const float4& x) __attribute__((always_inline)) { asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w)); };
  auto store_partial = [&]() __attribute__((always_inline)) {   // split-K: raw sums of this unit's K range
    int tmi, tni;
    tile_mn(pc, cmp_lv, tmi, tni);
    const int mb = tmi * BM + wm * TMB * 16 + r16, nb = tni * BN + wn * TNB * 16 + 4 * q4;
    float* __restrict__ ob = pc.partial + static_cast<long>(split_of(pc, cmp_lv)) * pc.M * pc.N + static_cast<long>(mb) * pc.N + nb;
#pragma unroll
    for (int j = 0; j < TNB; ++j)
#pragma unroll
      for (int i = 0; i < TMB; ++i) {
        const float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (mb + i * 16 < pc.M && nb + j * 16 < pc.N) st4(ob + (static_cast<long>(i) * 16 * pc.N + j * 16), v);
      }
  };
  // An unconditional use of every prefetched register.  hipcc puts s_waitcnt vmcnt(0) in front of it -- it cannot know about the
  // DMA ring, and vmcnt(0) drains the ring.  So the use sits right behind a mid-slice wait of the ring itself (the step AFTER the
  // one the operands were requested in): that wait is vmcnt(0) already, the compiler's is then free, and from there on hipcc knows
  // that nothing is pending -- the epilogue starts without a wait.  (In the epilogue itself the same wait cost 2-3 us per unit:
  // the first slices of the next unit, just requested, had to land before the first store could be issued.)
  auto touch_epilogue_operands = [&]() __attribute__((always_inline)) {
    if constexpr (PRE_RES) {
#pragma unroll
      for (int j = 0; j < TNB; ++j) {
        touch(rbias[j]); touch(rscale[j]); touch(rshift[j]);
#pragma unroll
        for (int i = 0; i < TMB; ++i) touch(rres[i][j]);
      }
    }
  };
  // no epilogue operand at all (the position products of the Winograd path, the tap products of UpEmbed / mt_proj): the sums go
  // out as they are -- ~60 instructions instead of the ~500 of the general epilogue, whose optional terms hipcc turns into selects
  auto store_plain = [&]() __attribute__((always_inline)) {
    int tmi, tni;
    tile_mn(pc, cmp_lv, tmi, tni);
    const int mb = tmi * BM + wm * TMB * 16 + r16, nb = tni * BN + wn * TNB * 16 + 4 * q4;
    float* __restrict__ ob = static_cast<float*>(cmp_out) + cmp_ob + static_cast<long>(mb) * pc.N + nb;
#pragma unroll
    for (int j = 0; j < TNB; ++j)
#pragma unroll
      for (int i = 0; i < TMB; ++i) {
        const float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (mb + i * 16 < pc.M && nb + j * 16 < pc.N) st4(ob + (static_cast<long>(i) * 16 * pc.N + j * 16), v);
      }
  };
  auto finish_tile = [&]() __attribute__((always_inline)) {
    if (pc.splits > 1) { store_partial(); return; }
    if constexpr (F32) {
      if (!pc.bias && !pc.scale && !pc.rowvec && !pc.residual && pc.act == DIFFSAL_ACT_NONE) { store_plain(); return; }
    }
    switch (pc.act) {
      case DIFFSAL_ACT_RELU: epilogue(std::integral_constant<int, DIFFSAL_ACT_RELU>{}); break;
      case DIFFSAL_ACT_GELU_ERF: epilogue(std::integral_constant<int, DIFFSAL_ACT_GELU_ERF>{}); break;
      case DIFFSAL_ACT_SIGMOID: epilogue(std::integral_constant<int, DIFFSAL_ACT_SIGMOID>{}); break;
      case DIFFSAL_ACT_GELU_GRAD: epilogue(std::integral_constant<int, DIFFSAL_ACT_GELU_GRAD>{}); break;
      default: epilogue(std::integral_constant<int, DIFFSAL_ACT_NONE>{}); break;
    }
  };

  // ---- prologue: P slices in flight; slice 0 must have landed before anyone reads it
#pragma unroll
  for (int s = 0; s < P; ++s) issue_slice(s);
  wait_vmcnt<(P - 1) * PPW>();
  __builtin_amdgcn_s_barrier();
  load_frags(smem, 0, 0);

  // one K slice: stage S is multiplied, stage S + 1 is read ahead, stage S - 1 (free after the barrier) takes the DMA of slice g + P
  // LAST: the unit's last pass over the ring -- stage 0 requests the epilogue operands behind its barrier, stage 1 uses them
  auto step = [&](auto idx, auto last_c) __attribute__((always_inline)) {
    constexpr int S = decltype(idx)::value;
    constexpr bool LAST = decltype(last_c)::value;
    const float* cur = smem + S * STAGE_F;
    const float* nxt = smem + ((S + 1) % STAGES) * STAGE_F;
    load_frags(cur, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
    // slice g + 1 (this wave's pieces) has landed; after the barrier every wave's pieces have, and nobody reads the stage of
    // slice g - 1 any more
    wait_vmcnt<WAIT_N>();
    __builtin_amdgcn_s_barrier();
    if constexpr (LAST && S == 0) {
      if (pc.splits == 1) fetch_epilogue_operands();
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LAST && S == 1) {
      touch_epilogue_operands();      // unconditional: on every path hipcc then knows that no request outlives the unit
      __builtin_amdgcn_sched_barrier(0);
    }
    // second half of the slice: its MFMAs (operands already in registers) with the DMA pieces of slice g + P and the fragment
    // reads of slice g + 1 slipped between them at even distances -- a wave alone on its SIMD must not stop issuing MFMAs
    // for the ~60 cycles a DMA instruction takes to issue
    {
      constexpr int NX = PPW + TMB + TNB;         // items to slip in: fragment reads first (the next step's first MFMAs wait for them)
#pragma unroll
      for (int idx = 0; idx < NM; ++idx) {
        mfma_one(1, idx);
#pragma unroll
        for (int x = idx * NX / NM; x < (idx + 1) * NX / NM; ++x) {
          __builtin_amdgcn_sched_barrier(0);
          if (x < TMB + TNB) load_frag(nxt, 0, 0, x);
          else issue_piece((S + P) % STAGES, x - TMB - TNB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      issue_advance(std::integral_constant<bool, S == 0>{});
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto ring = [&](auto self, auto idx, auto last_c) __attribute__((always_inline)) {
    constexpr int S = decltype(idx)::value;
    if constexpr (S < STAGES) {
      step(idx, last_c);
      self(self, std::integral_constant<int, S + 1>{}, last_c);
    }
  };
  for (int t = 0; t < my_tiles; ++t) {          // the host guarantees kt_per_unit % STAGES == 0: a unit starts on stage 0
    if constexpr (GROUPED) {
      int idx;
      locate(cmp_v, idx, cmp_lv);
      pc = pick_problem(g, idx);
      cmp_out = pc.out;
    } else {
      cmp_lv = cmp_v;
      if (g.batch + g.batch2 > 1) {
        const int per = g.n_tiles * g.splits;
        int ib;
        if (g.xcd_order) {
          const int q = cmp_v >> 3, ml = q / g.n_tiles_n, row = ml * 8 + (cmp_v & 7);
          ib = row / g.n_tiles_m;
          cmp_lv = (row - ib * g.n_tiles_m) * g.n_tiles_n + (q - ml * g.n_tiles_n);
        } else {
          ib = cmp_v / per;
          cmp_lv = cmp_v - ib * per;
        }
        if (ib < g.batch) {
          cmp_ob = ib * g.o_bs;
          cmp_out = g.out;
        } else {
          cmp_ob = (ib - g.batch) * g.o2_bs;
          cmp_out = g.out2;
        }
      }
    }
    const int nkt = pc.kt_per_unit;
    stamp(1 + 3 * t);
    for (int kt = STAGES; kt < nkt; kt += STAGES) ring(ring, std::integral_constant<int, 0>{}, std::false_type{});
    ring(ring, std::integral_constant<int, 0>{}, std::true_type{});
    stamp(2 + 3 * t);
    finish_tile();
    stamp(3 + 3 * t);
    cmp_v += gsz;
  }
}

// Sum of the split-K slabs in a fixed order (deterministic) + the epilogue; float4 over N.
template <typename T>
__global__ __launch_bounds__(256) void gemm_dma_reduce_kernel(DmaGemmArgs p) {
  const T* __restrict__ resid = static_cast<const T*>(p.residual);
  T* __restrict__ outp = static_cast<T*>(p.out);
  const int n4 = p.N >> 2;
  const long total = static_cast<long>(p.M) * n4, slab = static_cast<long>(p.M) * p.N;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int n = static_cast<int>(i % n4) * 4;
    const long m = i / n4, o = m * p.N + n;
    float4 a = ld4(p.partial + o);
    for (int s = 1; s < p.splits; ++s) {
      const float4 b = ld4(p.partial + s * slab + o);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = v[e];
      if (p.bias) x += p.bias[n + e];
      if (p.scale) x = x * p.scale[n + e] + p.shift[n + e];
      if (p.rowvec) x += p.rowvec[(m / p.rows_per_img) * p.rowvec_ld + n + e];
      if (p.act == DIFFSAL_ACT_RELU) x = fmaxf(x, 0.f);
      else if (p.act == DIFFSAL_ACT_GELU_ERF) x = gelu_erf(x);
      else if (p.act == DIFFSAL_ACT_SIGMOID) x = sigmoidf_(x);
      v[e] = x;
    }
    if (p.act == DIFFSAL_ACT_GELU_GRAD) {
      const float4 t = ld4(resid + o);
      v[0] *= gelu_erf_grad(t.x); v[1] *= gelu_erf_grad(t.y); v[2] *= gelu_erf_grad(t.z); v[3] *= gelu_erf_grad(t.w);
    } else if (resid) {
      const float4 t = ld4(resid + o);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    st4(outp + o, make_float4(v[0], v[1], v[2], v[3]));
  }
}

namespace {

// K split for a launch of `tiles` 96-wide tiles and kt K slices: units = tiles * S should fill 256 CUs (two workgroups each) without
// a long last round; every unit needs a multiple of `stages` slices.  Cost model in units of one K slice on one CU:
// rounds of 256 CUs x (slices per unit) + the slab round trip of a split (S + 1 passes over M x N at ~3 TB/s ~ slices).
int choose_split(long tiles, int kt, int stages, long mn, double* t_out = nullptr) {
  int best = 1;
  double best_t = 1e30;
  for (int S = 1; S <= 32; ++S) {
    if (kt % (S * stages) != 0) continue;
    const int per = kt / S;
    if (S > 1 && per < 2 * stages) break;
    const long units = tiles * S;
    const double rounds = static_cast<double>((units + 255) / 256);        // co-resident pairs share the matrix pipe
    double t = rounds * per * 1.13e-6;                                     // one 96 x 96 x 32 slice on one CU
    t += 3.0e-6 + 1.2e-6;                                                  // fill / drain
    if (S > 1) t += (S + 1.0) * mn * 4.0 / 3.0e12 + 4.0e-6;
    if (t < best_t) { best_t = t; best = S; }
  }
  if (t_out) *t_out = best_t;
  return best;
}

template <typename T, int TMB, int TNB, int STAGES, int OCC>
int launch_dma(DmaGemmArgs& a, bool conv, hipStream_t s) {
  constexpr int BM = 32 * TMB, BN = 32 * TNB;
  const int slots = 256 * OCC;
  const int units = a.n_tiles * a.splits * (a.batch + a.batch2);
  const int grid = units < slots ? units : slots;
#ifdef DIFFSAL_DEV_STAMPS
  a.stamps = (g_dma_stamps && static_cast<size_t>(grid) * 32 * 8 <= g_dma_stamp_bytes) ? g_dma_stamps : nullptr;
#endif
  // XCD-aware unit order (one XCD owns whole rows of N tiles) on plain products that fill the chip.  The same order over the
  // (position, M tile) rows of a BATCHED launch is implemented (DIFFSAL_BATCH_XCD=1) but off: measured on one box, alternating,
  // 3.58 against 3.50 ms per step -- an XCD then owns 4 or 5 of the 36 positions and the launch ends on the XCDs with 5.
  const long rows_total = static_cast<long>(a.n_tiles_m) * (a.batch + a.batch2);
  const bool batched = a.batch + a.batch2 > 1;
  a.xcd_order = (a.splits == 1 && tune(TUNE_NO_XCD_ORDER) != 1 && grid % 8 == 0 && a.n_tiles_n > 1 &&
                 (batched ? (tune(TUNE_BATCH_XCD) == 1 && units >= 128 && rows_total >= 16) : (a.n_tiles >= slots && a.n_tiles_m >= 16))) ? 1 : 0;
  if constexpr (sizeof(T) == 4) {
    if (conv) hipLaunchKernelGGL((gemm_dma_kernel<T, TMB, TNB, STAGES, OCC, true, false>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_dma_kernel<T, TMB, TNB, STAGES, OCC, false, false>), dim3(grid), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((gemm_dma_kernel<T, TMB, TNB, STAGES, OCC, false, false>), dim3(grid), dim3(256), 0, s, a);
  }
  note_kernel("gemm_dma_kernel<%s, %d, %d, %d, %d, %s, false> [%dx%d tile, %d stages, split-K %d, batch %d]",
              sizeof(T) == 4 ? "float" : (std::is_same<T, bf16_t>::value ? "__bf16" : "_Float16"), TMB, TNB, STAGES, OCC,
              conv ? "true" : "false", BM, BN, STAGES, a.splits, a.batch + a.batch2);
  int rc = check_launch("diffsal_conv_igemm(dma)");
  if (rc || a.splits == 1) return rc;
  long g = (static_cast<long>(a.M) * (a.N / 4) + 255) / 256;
  g = g > 2048 ? 2048 : g;
  hipLaunchKernelGGL(gemm_dma_reduce_kernel<T>, dim3(static_cast<unsigned>(g)), dim3(256), 0, s, a);
  return check_launch("diffsal_conv_igemm(dma split-K sum)");
}

struct DmaCfg { int bm, bn, stages; };
constexpr int kNumDmaCfgs = 2;                 // selectable through DIFFSAL_GEMM_DMA; entry 2: the wide tile of batched launches
const DmaCfg kDmaCfgs[kNumDmaCfgs + 2] = {{96, 96, 3}, {96, 96, 6}, {96, 128, 3}, {192, 192, 3}};   // entry 3: 16-bit storage, large products

// fills the problem-independent part of the arguments; false: shape not handled by tile configuration c.  esz: bytes per
// element of the storage type (a K slice is 128 bytes of a row)
bool dma_fill(DmaGemmArgs& g, const DmaCfg& c, const diffsal_conv_desc* d, bool as_conv, int esz, const void* a, const void* w,
              const float* bias, const float* scale, const float* shift, const float* rowvec, int rowvec_ld, const void* residual, void* out) {
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const int K = d->KH * d->KW * d->Cin, N = d->Cout;
  const int epk = 128 / esz;                                       // elements per K slice
  if (K % epk != 0 || (K / epk) % c.stages != 0) return false;     // a unit must start on ring stage 0
  if (esz != 4 && as_conv) return false;
  // 32-bit byte offsets: plain products address A rows relative to the tile (any M), the output / residual through 64-bit
  // pointers; the weight matrix and (convolutions) the whole input must stay below 4 GiB, which validate() has checked
  if (N % 4 != 0 || M <= 0 || M >= (1L << 31) || static_cast<long>(N) * K * esz >= (1L << 32) - 16 || 96L * K * esz >= (1L << 31)) return false;
  if (!aligned16(a) || !aligned16(w) || !aligned16(out) || (bias && !aligned16(bias)) || (scale && !(aligned16(scale) && aligned16(shift))) ||
      (rowvec && !(aligned16(rowvec) && rowvec_ld % 4 == 0)) || (residual && !aligned16(residual)) || d->KH * d->KW > 32)
    return false;
  g = DmaGemmArgs{};
  g.a = a; g.w = w; g.bias = bias; g.scale = scale; g.shift = shift; g.rowvec = rowvec; g.residual = residual; g.out = out;
  g.M = static_cast<int>(M); g.N = N; g.K = K; g.act = d->act; g.rowvec_ld = rowvec_ld; g.rows_per_img = d->Ho * d->Wo;
  g.n_tiles_m = static_cast<int>((M + c.bm - 1) / c.bm);
  g.n_tiles_n = (N + c.bn - 1) / c.bn;
  g.n_tiles = g.n_tiles_m * g.n_tiles_n;
  if (as_conv) {
    g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Ho = d->Ho; g.Wo = d->Wo; g.KW = d->KW; g.taps = d->KH * d->KW;
    g.stride_h = d->stride_h; g.stride_w = d->stride_w; g.pad_t = d->pad_t; g.pad_l = d->pad_l; g.dil_h = d->dil_h; g.dil_w = d->dil_w;
    g.in_bytes = static_cast<unsigned>(static_cast<long>(d->N) * d->H * d->W * d->Cin * 4);
  }
  g.splits = 1;
  g.kt_per_unit = K / epk;
  g.batch = 1;
  return true;
}

}  // namespace

// Tile configurations of this kernel, in the order of DIFFSAL_GEMM_DMA's value - 1.
//   0: 96 x 96, 3 stages (72 KB), two workgroups per CU      1: 96 x 96, 6 stages (144 KB), one per CU
// (96 x 192 tiles were built and measured in round 4: 5-15 % slower on every shape of the step, removed.)
// as_conv: the convolution form (taps, padding); false = d is a plain [M, K] x [N, K]^T product.  esz: 4 (fp32), 2 (dtype names
// bf16 or f16; plain products only).  Split-K is planned when the workspace allows (gemm_dma_ws_bytes).  Returns 1 if launched,
// 0 if the shape is not handled here, < 0 on error.
size_t gemm_dma_ws_bytes(int cfg, long M, int K, int N, int esz) {
  if (cfg < 0 || cfg >= kNumDmaCfgs || K % (128 / esz) != 0) return 0;
  const DmaCfg& c = kDmaCfgs[cfg];
  const long tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
  const int S = choose_split(tiles, K / (128 / esz), c.stages, M * N);
  return S > 1 ? static_cast<size_t>(S) * M * N * sizeof(float) : 0;
}

// the planner's time estimate (seconds) for this kernel on a plain fp32 M x K x N product, 1e30 if the shape is not handled
double gemm_dma_estimate(int cfg, long M, int K, int N) {
  if (cfg < 0 || cfg >= kNumDmaCfgs || K % 32 != 0 || (K / 32) % kDmaCfgs[cfg].stages != 0) return 1e30;
  const DmaCfg& c = kDmaCfgs[cfg];
  const long tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
  double t = 1e30;
  choose_split(tiles, K / 32, c.stages, M * N, &t);
  return t;
}

int try_gemm_dma(int cfg, const diffsal_conv_desc* d, bool as_conv, const void* a, const void* w, const float* bias, const float* scale,
                 const float* shift, const float* rowvec, int rowvec_ld, const void* residual, void* out, void* ws, size_t ws_bytes,
                 hipStream_t s, bool out_f32) {
  const int esz = d->dtype == DIFFSAL_F32 ? 4 : 2;
  const bool big16 = cfg == 3 && esz == 2;     // 192 x 192 tiles, one workgroup per CU: twice the flops per staged byte
  if (cfg < 0 || (cfg >= kNumDmaCfgs && !big16)) return 0;
  const DmaCfg& c = kDmaCfgs[cfg];
  DmaGemmArgs g;
  if (!dma_fill(g, c, d, as_conv, esz, a, w, bias, scale, shift, rowvec, rowvec_ld, residual, out)) return 0;
  const long MN = static_cast<long>(g.M) * g.N;
  const int kt = g.kt_per_unit;
  g.splits = big16 ? 1 : choose_split(g.n_tiles, kt, c.stages, MN);
  if (g.splits > 1 && (!ws || ws_bytes < static_cast<size_t>(g.splits) * MN * sizeof(float) || !aligned16(ws))) g.splits = 1;
  g.kt_per_unit = kt / g.splits;
  g.partial = g.splits > 1 ? static_cast<float*>(ws) : nullptr;
  if (out_f32) {
    if (esz == 4 || residual) return 0;         // 16-bit storage only; a residual would be read in the storage type
    g.out_f32 = 1;
    if (g.splits > 1) { g.splits = 1; g.kt_per_unit = kt; g.partial = nullptr; }     // the slab sum writes the storage type
  }
  int rc;
  if (d->dtype == DIFFSAL_F32) rc = cfg == 0 ? launch_dma<float, 3, 3, 3, 2>(g, as_conv, s) : launch_dma<float, 3, 3, 6, 1>(g, as_conv, s);
  else if (big16) rc = d->dtype == DIFFSAL_BF16 ? launch_dma<bf16_t, 6, 6, 3, 1>(g, false, s) : launch_dma<f16_t, 6, 6, 3, 1>(g, false, s);
  else if (cfg != 0) return 0;
  else if (d->dtype == DIFFSAL_BF16) rc = launch_dma<bf16_t, 3, 3, 3, 2>(g, false, s);
  else rc = launch_dma<f16_t, 3, 3, 3, 2>(g, false, s);
  return rc == DIFFSAL_OK ? 1 : rc;
}

// `batch` plain fp32 products of one shape in ONE launch of the 96 x 96 / 3-stage kernel, no epilogue: out_b [M, N] = a_b [M, K] x
// w_b [N, K]^T with the operands of problem b at a + b a_bs, w + b w_bs, out + b o_bs (elements).  Returns 1 if launched, 0 if the
// shape is not handled (K % 96 != 0, N % 4 != 0, misaligned), < 0 on error.
// side_rows > 0: the plain product side_out [side_rows, N] = side_a [side_rows, K] x side_w [N, K]^T shares the launch as
// side_rows / M further problems of the same shape (side_rows % M == 0, else 0 is returned and nothing is launched).
int gemm_dma_batched(const float* a, const float* w, float* out, long M, int K, int N, int batch, long a_bs, long w_bs, long o_bs,
                     hipStream_t s, const float* side_a, const float* side_w, float* side_out, long side_rows) {
  diffsal_conv_desc d{};
  d.N = 1; d.H = 1; d.W = static_cast<int>(M); d.Ho = 1; d.Wo = static_cast<int>(M); d.Cin = K; d.Cout = N; d.KH = 1; d.KW = 1;
  d.stride_h = d.stride_w = d.dil_h = d.dil_w = 1;
  d.dtype = DIFFSAL_F32;
  if (M <= 0 || M >= (1L << 31) || batch < 1 || a_bs % 4 != 0 || w_bs % 4 != 0 || o_bs % 4 != 0) return 0;
  const long nb2 = side_rows > 0 ? side_rows / M : 0;
  if (side_rows > 0 && (side_rows % M != 0 || !side_a || !side_w || !side_out || !aligned16(side_a) || !aligned16(side_w) ||
                        !aligned16(side_out) || nb2 > 4096))
    return 0;
  // Tile shape.  96 x 96 at two workgroups per CU is the default; 96 x 128 at one per CU (a third fewer, a third longer units)
  // where that shortens the busiest CU's share: the launch's units spread over 256 CUs in whole units, so 288 units of 96 x 96
  // (36 positions x 8 tiles of N = 768) put two on 32 CUs while 216 units of 96 x 128 are one round.  Cost in K slices of a
  // 96 x 96 unit on one CU; the wider tile is taken when it wins by 10 % (it has no co-resident workgroup to cover its
  // epilogues).  DIFFSAL_BATCH_TILE = 0 / 1 forces 96 x 96 / 96 x 128 (shape permitting).
  const long tm = (M + 95) / 96, nbt = batch + nb2;
  const long u96 = tm * ((N + 95) / 96) * nbt, u128 = tm * ((N + 127) / 128) * nbt;
  const double t96 = static_cast<double>((u96 + 255) / 256), t128 = static_cast<double>((u128 + 255) / 256) * (4.0 / 3.0);
  const int force = tune(TUNE_BATCH_TILE);
  const bool wide = N % 128 == 0 && (force == 1 || (force != 0 && t128 * 1.1 < t96));
  DmaGemmArgs g;
  if (!dma_fill(g, kDmaCfgs[wide ? 2 : 0], &d, false, 4, a, w, nullptr, nullptr, nullptr, nullptr, 0, nullptr, out)) return 0;
  g.batch = batch; g.a_bs = a_bs; g.w_bs = w_bs; g.o_bs = o_bs;
  if (side_rows > 0) {
    g.batch2 = static_cast<int>(nb2);
    g.a2 = side_a; g.w2 = side_w; g.out2 = side_out;
    g.a2_bs = M * K; g.o2_bs = M * N;
  }
  if (static_cast<long>(g.n_tiles) * (batch + g.batch2) >= (1L << 30)) return 0;
  const int rc = wide ? launch_dma<float, 3, 4, 3, 1>(g, false, s) : launch_dma<float, 3, 3, 3, 2>(g, false, s);
  return rc == DIFFSAL_OK ? 1 : rc;
}

// Up to four fp32 problems in ONE launch of the 96 x 96 / 3-stage kernel (convolution form: a plain product is its 1x1 case).
// No split-K: every problem's units are whole tiles; the host sorts the problems by K slices per unit (longest first) and the
// grid has one workgroup per unit, so the hardware dispatcher hands the next (shorter) unit to whichever CU frees up first.
// Returns 1 if launched, 0 if some problem does not fit this kernel (the caller then launches them one by one).
int try_gemm_dma_group(int n, const diffsal_conv_desc* const* d, const float* const* a, const float* const* w, const float* const* bias,
                       float* const* out, hipStream_t s) {
  if (n < 1 || n > kMaxGroup) return 0;
  const DmaCfg& c = kDmaCfgs[0];
  DmaGemmArgs pr[kMaxGroup];
  for (int i = 0; i < n; ++i)
    if (!dma_fill(pr[i], c, d[i], true, 4, a[i], w[i], bias ? bias[i] : nullptr, nullptr, nullptr, nullptr, 0, nullptr, out[i])) return 0;
  int order[kMaxGroup];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (pr[order[j]].kt_per_unit > pr[order[i]].kt_per_unit) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  DmaGroupArgs g{};
  g.n = n;
  int units = 0;
  for (int i = 0; i < kMaxGroup; ++i) {
    if (i < n) {
      g.prob[i] = pr[order[i]];
      units += g.prob[i].n_tiles;
    } else {
      g.prob[i] = g.prob[n - 1];
    }
    g.unit_end[i] = i < n ? units : 0x7FFFFFFF;
  }
  g.total = units;
  const int persist = tune(TUNE_GROUP_GRID);   // > 0: a persistent grid of that many workgroups walks the units with stride
  const int grid = persist > 0 && units > persist ? persist : units;
  hipLaunchKernelGGL((gemm_dma_kernel<float, 3, 3, 3, 2, true, true>), dim3(grid), dim3(256), 0, s, g);
  note_kernel("gemm_dma_kernel<float, 3, 3, 3, 2, true, true> [96x96 tile, 3 stages, %d problems, %d units]", n, units);
  const int rc = check_launch("diffsal_conv_igemm_group(dma)");
  return rc == DIFFSAL_OK ? 1 : rc;
}

}  // namespace diffsal

#ifdef DIFFSAL_DEV_STAMPS
// Development builds only (hipcc -DDIFFSAL_DEV_STAMPS; tools/probe_dma_stamps.py): a caller-owned device buffer receives 32 time
// stamps per workgroup of every following single-problem launch of gemm_dma_kernel that fits in `bytes` (slot 0: kernel entry;
// 1 + 3t, 2 + 3t, 3 + 3t: unit t starts, its K walk is done, its epilogue is done); (nullptr, 0) turns it off.
extern "C" int diffsal_set_dma_stamps(void* device_buffer, size_t bytes) {
  diffsal::g_dma_stamps = static_cast<unsigned long long*>(device_buffer);
  diffsal::g_dma_stamp_bytes = device_buffer ? bytes : 0;
  return DIFFSAL_OK;
}
#endif
