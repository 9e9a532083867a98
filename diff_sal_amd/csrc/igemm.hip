// Implicit-GEMM convolution / linear layer on the gfx950 fp32 matrix cores.
//
//   out[m, co] = epilogue( sum_k A[m, k] * Wt[co, k] )
//
// A is the im2col view of a channels-last image (never materialised): m = (n, oy, ox),
// k = (ci_chunk, ky, kx, ci % 32): a 32-wide K slice is one contiguous 128-byte run of one input
// pixel, and the KH*KW taps of one 32-channel chunk are consecutive K slices, so a workgroup
// sweeps all taps over the same few KB of input while they are still in L1/L2 (tap-major order
// re-streams the whole channel depth once per tap and overflows the 4 MiB per-XCD L2).
// Wt is the weight packed as [Cout][K] in the same k order: both operands are K-contiguous
// ("NT" GEMM), both are staged through LDS with a 36-dword row pitch (conflict-free for
// ds_read_b128 on the 64-bank LDS), and each lane reads 4 consecutive k of its row at once.
// v_mfma_f32_32x32x2_f32 consumes k pairs {j, j+4} of an 8-wide group (lane half h supplies
// k = 4h + j); the k order inside a tile is irrelevant as long as A and B agree.
//
// Numerics: exact fp32 FMA chains (the MFMA is bit-for-bit an fmaf chain), no reduced
// precision anywhere -- this is what the 1e-3 parity bar is measured on.
//
// Replaces the cuDNN/cuBLAS call sites behind torch.nn.Conv2d / Linear / Conv3d(k,1,1) in
// R/models/saliency_decoder/{sal_unet,common_block,attention,transformer}.py (see diffsal.h).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace diffsal {

// igemm16.hip: the same operator on bf16 / fp16 storage (native 16-bit MFMA)
size_t igemm16_ws_bytes(const diffsal_conv_desc* d);
int igemm16_launch(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                   const float* shift, const float* rowvec, const void* residual, void* out, void* ws, size_t ws_bytes,
                   hipStream_t s, const void* in2 = nullptr, const void* w2 = nullptr, const float* bias2 = nullptr,
                   void* out2 = nullptr);

// lin_stream.hip: barrier-free streaming kernel for short-K, huge-M linear layers
int try_linear_stream(const float* x, const float* w, const float* bias, const float* residual, float* out, long M,
                      int K, int N, int act, hipStream_t s);

// gemm_dma.hip: products / convolutions with LDS-DMA staging, 96-wide tiles (fp32; plain products also on 16-bit storage)
int try_gemm_dma(int cfg, const diffsal_conv_desc* d, bool as_conv, const void* a, const void* w, const float* bias, const float* scale,
                 const float* shift, const float* rowvec, int rowvec_ld, const void* residual, void* out, void* ws, size_t ws_bytes,
                 hipStream_t s, bool out_f32 = false);
int try_gemm_dma_group(int n, const diffsal_conv_desc* const* d, const float* const* a, const float* const* w, const float* const* bias,
                       float* const* out, hipStream_t s);
size_t gemm_dma_ws_bytes(int cfg, long M, int K, int N, int esz);
// gemm16_dma.hip: 16-bit plain products on 256 x 96 tiles, two workgroups per CU
int try_gemm16_dma2(const diffsal_conv_desc* d, const void* a, const void* w, const float* bias, const float* scale, const float* shift,
                    const float* rowvec, int rowvec_ld, const void* residual, void* out, hipStream_t s, bool out_f32);
double gemm_dma_estimate(int cfg, long M, int K, int N);

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmArgs {
  const float* in;
  const float* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const float* residual;
  float* out;
  int M, K;
  int H, W, Cin, Ho, Wo, Cout;
  int KW, taps, stride_h, stride_w, pad_t, pad_l, dil_h, dil_w;
  int act, rowvec_ld;
  int linear;                // 1: 1x1 / stride 1 / no padding (a plain [M, Cin] x [Cin, Cout] product)
  int persist_wgs;           // > 0: workgroups of the persistent linear kernel (256 CUs x residents of the tile shape)
  int w_split;               // bf16x3 mode: weights arrive pre-split (diffsal_split_weight), no conversion of the B operand
  int n_tiles_n, n_tiles;  // tiles along N, total tiles
  unsigned in_bytes, w_bytes;  // sizes for the buffer descriptors (< 4 GiB each)
  int splits, kt_per_split;  // split-K: workgroup (tile, s) covers K slices [s*kt_per_split, ...)
  float* partial;            // [splits][M][Cout] raw partial sums when splits > 1
  int vec_epilogue;          // 1: Cout % 4 == 0 and every epilogue pointer is 16-byte aligned
  int xcd_order;             // persistent linear kernel: tiles walked so that one XCD owns whole M-tile rows (see the kernel)
  // pair launch (diffsal_linear_pair): a second problem of the SAME shape rides in the same grid (blockIdx.z = 1)
  int pair;
  const float* in2;
  const float* w2;
  const float* bias2;
  float* out2;
  float* partial2;
};

// blockIdx.z = 1 of a pair launch works on the second pointer set (uniform selects on the kernel arguments)
__device__ __forceinline__ void select_pair(IgemmArgs& p, int which) {
  if (p.pair && which) { p.in = p.in2; p.w = p.w2; p.bias = p.bias2; p.out = p.out2; p.partial = p.partial2; }
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;
constexpr int PITCH = BK + 4;  // dwords; 36*r mod 64 hits 16 distinct 4-bank slots for 16 rows

// PREC 0: exact fp32 on v_mfma_f32_32x32x2_f32 (the default, what the parity numbers are quoted on).
// PREC 1: "bf16x3" -- every fp32 operand is split once, on its way into LDS, into hi = bf16(x) and lo = bf16(x - hi); the
//   product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate, fp32 accumulation).  Dropped
//   term lo*lo ~ 2^-16 relative per product; measured ~4e-6 of the output maximum on a K = 1728 convolution GEMM
//   (tools/spikes/bf16x3_gemm.hip).  Opt-in (diffsal_set_gemm_precision), reported separately from the headline.
template <int WM, int WN, int TM, int TN, int PREC>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmArgs p) {
  select_pair(p, blockIdx.z);
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  // A-operand loader: fp32 -> a thread owns 4 consecutive k of a row (one 16-byte load), 32 rows per pass;
  // bf16x3 -> 8 consecutive k (two loads), 64 rows per pass, so that its bf16 hi and lo halves are 16 bytes each and go
  // to LDS as conflict-free ds_write_b128 (the 8-byte stores of a 4-k owner were 4-way bank conflicted).
  constexpr int A_RPP = PREC == 1 ? 64 : 32;
  constexpr int A_LPT = PREC == 1 ? 2 : 1;
  constexpr int A_PASSES = BM / A_RPP;
  constexpr int A_REGS = A_PASSES * A_LPT;
  constexpr int B_PASSES = BN / 32;
  constexpr int STAGE = (BM + BN) * PITCH;  // floats per LDS stage
  static_assert(WM * WN == 4, "4 waves per workgroup");

  // Two LDS stages: while the MFMAs consume stage `cur`, the same wave parks the next K slice
  // (already in registers) into stage `cur ^ 1` and issues the global loads of the slice after
  // that.  One barrier per K slice; no phase in which the matrix pipe has nothing to do.
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous run of tiles (same-M tiles share their A rows in one L2).  Bijective for any count.
  int tile;
  const int split = blockIdx.y;
  {
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

  // ---- loader mapping: 8 lanes cover one 32-float K slice of a row; 32 rows per pass ----
  // Loads go through buffer descriptors: a 32-bit byte offset per row, the per-slice displacement
  // added as a scalar, and padding taps / rows handled by the hardware range check (an offset of
  // 0xFFFFFFFF is out of range => the load returns 0 and touches no memory).  No branches, no selects
  // on the data: ~4 VALU per load, so the loader hides completely under the MFMAs.
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 4;
  const int lrow_a = PREC == 1 ? tid >> 2 : lrow;
  const int lcol_a = PREC == 1 ? (tid & 3) * 8 : lcol;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in), 0, static_cast<int>(p.in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, static_cast<int>(p.w_bytes), 0x00020000);

  unsigned a_voff[A_PASSES];   // byte offset of (n, iy0, ix0, lcol); wraps for negative iy0/ix0, only used when valid
  unsigned a_valid[A_PASSES];  // bit t: tap t of this row lies inside the image
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < A_PASSES; ++j) {
    int m = m0 + lrow_a + A_RPP * j;
    m = m < p.M ? m : p.M - 1;  // rows past M compute garbage that is never stored
    if (p.linear) {   // 1x1, stride 1, no padding: row m of a [M, Cin] matrix -- skips three integer divisions per row
      a_voff[j] = static_cast<unsigned>(m * p.Cin + lcol_a) * 4u;
      a_valid[j] = 1u;
      continue;
    }
    const int n = m / HoWo;
    const int rem = m - n * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride_h - p.pad_t;
    const int ix0 = ox * p.stride_w - p.pad_l;
    a_voff[j] = static_cast<unsigned>(((n * p.H + iy0) * p.W + ix0) * p.Cin + lcol_a) * 4u;
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
    n = n < p.Cout ? n : p.Cout - 1;  // columns past Cout are never stored
    b_voff[j] = static_cast<unsigned>(n * p.K + lcol) * 4u;
  }

  float4 ra[A_REGS], rb[B_PASSES];
  const int kt_begin = split * p.kt_per_split;
  const int kt_end = min(p.K / BK, kt_begin + p.kt_per_split);
  const int nkt = kt_end - kt_begin;

  auto issue_loads = [&](int kt, bool live) {
    const int chunk = kt / p.taps;  // wave-uniform (scalar unit)
    const int tap = kt - chunk * p.taps;
    const int ky = tap / p.KW;
    const int kx = tap - ky * p.KW;
    const unsigned delta = static_cast<unsigned>((ky * p.dil_h * p.W + kx * p.dil_w) * p.Cin + chunk * BK) * 4u;
    const unsigned dead = live ? 0u : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) {
      const unsigned oob = ((a_valid[j] >> tap) & 1u) - 1u;  // 0 when the tap is inside, else all ones
#pragma unroll
      for (int l = 0; l < A_LPT; ++l) {
        const unsigned off = (a_voff[j] + delta + 16u * l) | oob | dead;
        ra[j * A_LPT + l] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, off, 0, 0));
      }
    }
    const unsigned kofs = static_cast<unsigned>(kt * BK) * 4u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (b_voff[j] + kofs) | dead, 0, 0));
  };
  // PREC 1 row layout (same 36-dword pitch): dwords 0..15 = the 32 hi halves (k order), 16..31 = the 32 lo halves
  // two elements at a time: cvt_pk, shift / mask back to fp32, packed subtract, cvt_pk = 2.5 VALU per element
  auto split_pair = [](float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 v = {x0, x1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xFFFF0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf16x2));
  };
  auto split_store = [&](float* rowp, const float4& v) {
    uint2 hi, lo;
    split_pair(v.x, v.y, hi.x, lo.x);
    split_pair(v.z, v.w, hi.y, lo.y);
    *reinterpret_cast<uint2*>(rowp + (lcol >> 1)) = hi;
    *reinterpret_cast<uint2*>(rowp + 16 + (lcol >> 1)) = lo;
  };
  auto split_store8 = [&](float* rowp, const float4& v0, const float4& v1) {   // 8 consecutive k: 16 B of hi, 16 B of lo
    uint4 hi, lo;
    split_pair(v0.x, v0.y, hi.x, lo.x);
    split_pair(v0.z, v0.w, hi.y, lo.y);
    split_pair(v1.x, v1.y, hi.z, lo.z);
    split_pair(v1.z, v1.w, hi.w, lo.w);
    *reinterpret_cast<uint4*>(rowp + (lcol_a >> 1)) = hi;
    *reinterpret_cast<uint4*>(rowp + 16 + (lcol_a >> 1)) = lo;
  };
  auto store_tile = [&](float* stage) {
    if constexpr (PREC == 0) {
#pragma unroll
      for (int j = 0; j < A_PASSES; ++j) st4(&stage[(lrow + 32 * j) * PITCH + lcol], ra[j]);
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j) st4(&stage[(BM + lrow + 32 * j) * PITCH + lcol], rb[j]);
    } else {
#pragma unroll
      for (int j = 0; j < A_PASSES; ++j) split_store8(&stage[(lrow_a + A_RPP * j) * PITCH], ra[j * A_LPT], ra[j * A_LPT + A_LPT - 1]);
      if (p.w_split) {
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) st4(&stage[(BM + lrow + 32 * j) * PITCH + lcol], rb[j]);
      } else {
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) split_store(&stage[(BM + lrow + 32 * j) * PITCH], rb[j]);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int fk = (lane >> 5) * 4;
  const int a_frag = (wm * TM * 32 + frow) * PITCH + fk;
  const int b_frag = (BM + wn * TN * 32 + frow) * PITCH + fk;

  // Fragment registers are double-buffered: the ds_reads of group kk+1 are issued before the MFMAs of
  // group kk, and the first group of the NEXT K slice is fetched behind the barrier while the last
  // group of this slice still runs, so the matrix pipe never waits for LDS latency.
  float4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](const float* stage, int kk, int set) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[set][i] = ld4(stage + a_frag + i * 32 * PITCH + kk * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[set][j] = ld4(stage + b_frag + j * 32 * PITCH + kk * 8);
  };
  auto do_mfmas = [&](int set) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float av = s == 0 ? fa[set][i].x : s == 1 ? fa[set][i].y : s == 2 ? fa[set][i].z : fa[set][i].w;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float bv = s == 0 ? fb[set][j].x : s == 1 ? fb[set][j].y : s == 2 ? fb[set][j].z : fb[set][j].w;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[i][j], 0, 0, 0);   // D^T: rows = channels, cols = output rows
        }
      }
    }
  };

  // prologue: the loads of slices 0 and 1 are in flight together (slice 1 borrows a second register set)
  issue_loads(kt_begin + 1, nkt > 1);
  float4 ta[A_REGS], tb[B_PASSES];
#pragma unroll
  for (int j = 0; j < A_REGS; ++j) ta[j] = ra[j];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) tb[j] = rb[j];
  issue_loads(kt_begin, true);
  store_tile(smem);
#pragma unroll
  for (int j = 0; j < A_REGS; ++j) ra[j] = ta[j];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) rb[j] = tb[j];
  __syncthreads();
  if constexpr (PREC == 0) {
    load_frags(smem, 0, 0);
    for (int it = 0; it < nkt; ++it) {
      float* cur = smem + (it & 1) * STAGE;
      float* nxt = smem + ((it & 1) ^ 1) * STAGE;
      // group 0: MFMAs on set 0; meanwhile fetch group 1, park slice it+1 in the other stage
      load_frags(cur, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      do_mfmas(0);
      store_tile(nxt);
      // the registers are free again: slice it+2 starts its trip now and has a whole K slice of MFMAs to land
      issue_loads(kt_begin + it + 2, it + 2 < nkt);
      {  // spread the LDS writes over the first half of this group's MFMAs and the buffer loads over the second
        constexpr int NM = 4 * TM * TN, NW = A_PASSES + B_PASSES, H1 = NM / 2, PER = (NW + H1 - 1) / H1;
  #pragma unroll
        for (int i = 0; i < H1; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x200, PER, 0);  // then LDS writes
        }
  #pragma unroll
        for (int i = H1; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, PER, 0);  // then buffer loads
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // group 1: meanwhile fetch group 2
      load_frags(cur, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      do_mfmas(1);
      __builtin_amdgcn_sched_barrier(0);
      // group 2: meanwhile fetch group 3
      load_frags(cur, 3, 1);
      __builtin_amdgcn_sched_barrier(0);
      do_mfmas(0);
      __builtin_amdgcn_sched_barrier(0);
      // every read of `cur` and every write of `nxt` has been issued: one barrier per K slice
      __syncthreads();
      // group 3: meanwhile fetch group 0 of the next slice
      load_frags(nxt, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      do_mfmas(1);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // bf16x3: a K slice is two 16-wide MFMA steps; lane half h owns k = g*16 + h*8 .. +7 of step g (A and B agree, so
    // the k order inside a step is free): one ds_read_b128 of hi halves and one of lo halves per 32-row tile and step.
    bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    const int a_frag3 = (wm * TM * 32 + frow) * PITCH + (lane >> 5) * 4;
    const int b_frag3 = (BM + wn * TN * 32 + frow) * PITCH + (lane >> 5) * 4;
    auto load_frags3 = [&](const float* stage, int g, int set) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[set][i] = __builtin_bit_cast(bf16x8, ld4(stage + a_frag3 + i * 32 * PITCH + g * 8));
        al[set][i] = __builtin_bit_cast(bf16x8, ld4(stage + a_frag3 + i * 32 * PITCH + 16 + g * 8));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[set][j] = __builtin_bit_cast(bf16x8, ld4(stage + b_frag3 + j * 32 * PITCH + g * 8));
        bl[set][j] = __builtin_bit_cast(bf16x8, ld4(stage + b_frag3 + j * 32 * PITCH + 16 + g * 8));
      }
    };
    auto do_mfmas3 = [&](int set) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {   // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[set][j], ah[set][i], acc[i][j], 0, 0, 0);   // D^T, as above
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[set][j], al[set][i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[set][j], ah[set][i], acc[i][j], 0, 0, 0);
        }
    };
    // The MFMA phase of a K slice is ~5x shorter than in fp32, so global loads are issued THREE slices ahead (two
    // register sets in flight) instead of two: at iteration `it` set it&1 holds slice it+1 (landed), the other set
    // holds slice it+2 (in flight); after parking set it&1 in LDS it is re-used for slice it+3.
    float4 qa[2][A_REGS], qb[2][B_PASSES];
    auto issue_loads3 = [&](int kt, bool live, int set) {
      const int chunk = kt / p.taps;
      const int tap = kt - chunk * p.taps;
      const int ky = tap / p.KW;
      const int kx = tap - ky * p.KW;
      const unsigned delta = static_cast<unsigned>((ky * p.dil_h * p.W + kx * p.dil_w) * p.Cin + chunk * BK) * 4u;
      const unsigned dead = live ? 0u : 0xFFFFFFFFu;
#pragma unroll
      for (int j = 0; j < A_PASSES; ++j) {
        const unsigned oob = ((a_valid[j] >> tap) & 1u) - 1u;
#pragma unroll
        for (int l = 0; l < A_LPT; ++l)
          qa[set][j * A_LPT + l] = __builtin_bit_cast(
              float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (a_voff[j] + delta + 16u * l) | oob | dead, 0, 0));
      }
      const unsigned kofs = static_cast<unsigned>(kt * BK) * 4u;
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j)
        qb[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (b_voff[j] + kofs) | dead, 0, 0));
    };
    auto store_tile3 = [&](float* stage, int set) {
#pragma unroll
      for (int j = 0; j < A_PASSES; ++j)
        split_store8(&stage[(lrow_a + A_RPP * j) * PITCH], qa[set][j * A_LPT], qa[set][j * A_LPT + A_LPT - 1]);
      if (p.w_split) {   // pre-split rows already have the LDS row layout: 16 dwords hi | 16 dwords lo
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) st4(&stage[(BM + lrow + 32 * j) * PITCH + lcol], qb[set][j]);
      } else {
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) split_store(&stage[(BM + lrow + 32 * j) * PITCH], qb[set][j]);
      }
    };
    // (the generic prologue above already parked slice 0 in stage 0 and holds slice 1 in ra/rb)
#pragma unroll
    for (int j = 0; j < A_REGS; ++j) qa[0][j] = ra[j];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) qb[0][j] = rb[j];
    issue_loads3(kt_begin + 2, nkt > 2, 1);
    load_frags3(smem, 0, 0);
    auto slice = [&](int it, auto set_c) {       // set index must be a compile-time constant (register arrays)
      constexpr int SET = decltype(set_c)::value;
      float* cur = smem + SET * STAGE;            // slice parity == SET by construction
      float* nxt = smem + (SET ^ 1) * STAGE;
      load_frags3(cur, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      do_mfmas3(0);
      store_tile3(nxt, SET);
      issue_loads3(kt_begin + it + 3, it + 3 < nkt, SET);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();          // all reads of `cur`, all writes of `nxt` issued
      load_frags3(nxt, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      do_mfmas3(1);
      __builtin_amdgcn_sched_barrier(0);
    };
    int it = 0;
    for (; it + 1 < nkt; it += 2) {
      slice(it, std::integral_constant<int, 0>{});
      slice(it + 1, std::integral_constant<int, 1>{});
    }
    if (it < nkt) slice(it, std::integral_constant<int, 0>{});
  }

  // ---- epilogue.  The weight fragment is the MFMA "A" operand, so D is the transposed tile: a lane holds, for output
  // row lane & 31 of its 32-row block, channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- register quads of four
  // consecutive channels.  The sums are staged through the idle LDS stages, NPASS row groups at a time, and leave as
  // fully coalesced 16-byte pieces (a 96-channel row is 384 contiguous bytes) with the epilogue applied on the way
  // (16-byte stores straight from the quads measured 2 % slower end to end here, 5-15 % FASTER in the 16-bit kernel);
  // scalar stores from the register layout (2 x 128 bytes per wave instruction) cost the short-K GEMMs a third of their time.
  const int col_l = lane & 31;
  const int hq = (lane >> 5) * 4;
  const float* __restrict__ resid = p.residual;
  const float* __restrict__ rowv = p.rowvec;
  float* __restrict__ outp = p.out;
  float* part = p.splits > 1 ? p.partial + static_cast<long>(split) * p.M * p.Cout : nullptr;
  if (!p.vec_epilogue) {   // Cout % 4 != 0 or a pointer not 16-byte aligned: scalar stores from the register layout
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
          if (p.act == DIFFSAL_ACT_GELU_GRAD) v *= gelu_erf_grad(resid[o]);
          else if (resid) v += resid[o];
          outp[o] = v;
        }
      }
    }
    return;
  }
  constexpr int CP = BN + 4;                                  // staging pitch (dwords): conflict-free 16-byte stores
  constexpr int NPASS = (BM * CP <= 2 * STAGE) ? 1 : ((BM / 2) * CP <= 2 * STAGE ? 2 : 4);
  static_assert(WM % NPASS == 0 && (BM / NPASS) * CP <= 2 * STAGE, "staging passes must split the wave rows");
  constexpr int RP = BM / NPASS;                              // rows per pass
  constexpr int C4 = BN / 4;
  __syncthreads();                                            // the last stage's fragment reads are done
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    if (wm / (WM / NPASS) == ps) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = ((wm % (WM / NPASS)) * TM + i) * 32 + col_l;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            st4(smem + row * CP + (wn * TN + j) * 32 + g * 8 + hq,
                make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < RP * C4 / 256; ++k) {
      const int idx = tid + k * 256;
      const int row = idx / C4, c4 = idx - row * C4;
      const int m = m0 + ps * RP + row, n = n0 + c4 * 4;
      if (m >= p.M || n >= p.Cout) continue;
      const float4 a4 = ld4(smem + row * CP + c4 * 4);
      const long o = static_cast<long>(m) * p.Cout + n;
      if (part) { st4(part + o, a4); continue; }
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
      if (p.bias) { const float4 t = ld4(p.bias + n); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
      if (p.scale) {
        const float4 sc = ld4(p.scale + n);
        const float4 sh = p.shift ? ld4(p.shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
      }
      if (rowv) {
        const float4 t = ld4(rowv + static_cast<long>(m / HoWo) * p.rowvec_ld + n);
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
      if (p.act == DIFFSAL_ACT_GELU_GRAD) {
        const float4 t = ld4(resid + o);
        v[0] *= gelu_erf_grad(t.x); v[1] *= gelu_erf_grad(t.y); v[2] *= gelu_erf_grad(t.z); v[3] *= gelu_erf_grad(t.w);
      } else if (resid) { const float4 t = ld4(resid + o); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
      st4(outp + o, make_float4(v[0], v[1], v[2], v[3]));
    }
    if (ps + 1 < NPASS) __syncthreads();
  }
}

// =================================================================================================================
// Persistent form for plain [M, K] x [K, N] products in exact fp32 (1x1 convolutions / Linear layers, no split-K): a
// workgroup walks a strided list of tiles and the two-slices-ahead prefetch runs on into the next tile, whose origin is
// just two byte offsets; the epilogue stores the register quads directly (no LDS, no barrier).  Per-tile index
// arithmetic, first-load latency and store tail (1.8 + 1.5 + 4.7 us per 128 x 96 tile by the in-kernel stamps, against
// ~9 us of MFMAs at K = 192) leave the critical path.  Same arithmetic and summation order as igemm_kernel<.., 0>.
// =================================================================================================================
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, (TM * TN <= 4 ? 2 : 1)) void igemm_linear_kernel(IgemmArgs p) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int A_PASSES = BM / 32;
  constexpr int B_PASSES = BN / 32;
  constexpr int STAGE = (BM + BN) * PITCH;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, static_cast<int>(p.in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, static_cast<int>(p.w_bytes), 0x00020000);
  unsigned a_rel[A_PASSES], b_rel[B_PASSES];
#pragma unroll
  for (int j = 0; j < A_PASSES; ++j) a_rel[j] = static_cast<unsigned>((lrow + 32 * j) * p.K + lcol) * 4u;
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) b_rel[j] = static_cast<unsigned>((lrow + 32 * j) * p.K + lcol) * 4u;

  const int nkt = p.K / BK;
  // Tile walk: virtual index v = blockIdx.x + i * gridDim.x.  Plain order: v = tile, N fastest.  XCD order (large launches,
  // gridDim.x % 8 == 0): workgroup b runs on XCD b % 8, and XCD x owns the M tiles x, x + 8, ... with all their N tiles, so the
  // N tiles that share an A row block meet in ONE L2 instead of eight.  v_end: first virtual index past this worker's tiles.
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
  const int total = my_tiles * nkt;
  int iss_w = blockIdx.x, iss_kt = 0;
  unsigned iss_a = 0, iss_b = 0;
  auto origins = [&](int w, unsigned& oa, unsigned& ob) __attribute__((always_inline)) {
    int tmi, tni;
    tile_mn(w, tmi, tni);
    oa = static_cast<unsigned>(tmi) * static_cast<unsigned>(BM * p.K * 4);
    ob = static_cast<unsigned>(tni) * static_cast<unsigned>(BN * p.K * 4);
  };
  origins(iss_w, iss_a, iss_b);
  float4 ra[A_PASSES], rb[B_PASSES];
  auto issue_next = [&]() __attribute__((always_inline)) {
    const unsigned dead = iss_w < n_tiles ? 0u : 0xFFFFFFFFu;
    const unsigned kofs = static_cast<unsigned>(iss_kt * BK) * 4u;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j)
      ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (iss_a + a_rel[j] + kofs) | dead, 0, 0));
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (iss_b + b_rel[j] + kofs) | dead, 0, 0));
    if (++iss_kt == nkt) {
      iss_kt = 0;
      iss_w += gridDim.x;
      if (iss_w < n_tiles) origins(iss_w, iss_a, iss_b);
    }
  };
  auto store_tile = [&](float* stage) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) st4(&stage[(lrow + 32 * j) * PITCH + lcol], ra[j]);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) st4(&stage[(BM + lrow + 32 * j) * PITCH + lcol], rb[j]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  const int a_frag = (wm * TM * 32 + frow) * PITCH + fk;
  const int b_frag = (BM + wn * TN * 32 + frow) * PITCH + fk;
  float4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](const float* stage, int kk, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[set][i] = ld4(stage + a_frag + i * 32 * PITCH + kk * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[set][j] = ld4(stage + b_frag + j * 32 * PITCH + kk * 8);
  };
  auto do_mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float av = s4 == 0 ? fa[set][i].x : s4 == 1 ? fa[set][i].y : s4 == 2 ? fa[set][i].z : fa[set][i].w;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float bv = s4 == 0 ? fb[set][j].x : s4 == 1 ? fb[set][j].y : s4 == 2 ? fb[set][j].z : fb[set][j].w;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[i][j], 0, 0, 0);   // D^T: rows = channels, cols = output rows
        }
      }
    }
  };

  int cmp_w = blockIdx.x, cmp_kt = 0;
  const int col_l = lane & 31, hq = (lane >> 5) * 4;
  const float* __restrict__ resid = p.residual;
  float* __restrict__ outp = p.out;
  // the residual quads of the tile being multiplied are fetched at its first K slice and wait in registers: in the epilogue
  // they would be a dependent round trip with the matrix pipe idle
  float4 rres[TM][TN][4];
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
          rres[i][j][g] = ld4(resid + (ok ? static_cast<long>(m) * p.Cout + n : 0));
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
          if (p.act == DIFFSAL_ACT_GELU_GRAD) {
            const float4 t = rres[i][j][g];
            v[0] *= gelu_erf_grad(t.x); v[1] *= gelu_erf_grad(t.y); v[2] *= gelu_erf_grad(t.z); v[3] *= gelu_erf_grad(t.w);
          } else if (resid) { const float4 t = rres[i][j][g]; v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
          st4(outp + o, make_float4(v[0], v[1], v[2], v[3]));
        }
      }
    }
  };

  // prologue: slices 0 and 1 in flight together (slice 1 borrows a second register set)
  issue_next();
  float4 ta[A_PASSES], tb[B_PASSES];
#pragma unroll
  for (int j = 0; j < A_PASSES; ++j) ta[j] = ra[j];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) tb[j] = rb[j];
  issue_next();
  {  // park slice 0 (in ta/tb); ra/rb keep slice 1
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) st4(&smem[(lrow + 32 * j) * PITCH + lcol], ta[j]);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) st4(&smem[(BM + lrow + 32 * j) * PITCH + lcol], tb[j]);
  }
  __syncthreads();
  load_frags(smem, 0, 0);
  for (int g = 0; g < total; ++g) {
    float* cur = smem + (g & 1) * STAGE;
    float* nxt = smem + ((g & 1) ^ 1) * STAGE;
    if (resid && cmp_kt == 0) fetch_residual();
    load_frags(cur, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    do_mfmas(0);
    store_tile(nxt);        // slice g+1
    issue_next();           // slice g+2 starts its trip
    {  // spread the LDS writes over the first half of this group's MFMAs and the buffer loads over the second
      constexpr int NM = 4 * TM * TN, NW = A_PASSES + B_PASSES, H1 = NM / 2, PER = (NW + H1 - 1) / H1;
#pragma unroll
      for (int i = 0; i < H1; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, PER, 0);
      }
#pragma unroll
      for (int i = H1; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, PER, 0);
      }
    }
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
    if (++cmp_kt == nkt) {   // the tile's last K slice: store it, start the next tile from zero
      finish_tile();
      cmp_kt = 0;
      cmp_w += gridDim.x;
    }
  }
}

// Sum the split-K slabs in a fixed order (deterministic) and apply the epilogue; float4 over Cout.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(IgemmArgs p) {
  select_pair(p, blockIdx.y);
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
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j];
      if (p.bias) x += p.bias[n + j];
      if (p.scale) x = x * p.scale[n + j] + p.shift[n + j];
      if (p.rowvec) x += p.rowvec[(m / HoWo) * p.rowvec_ld + n + j];
      if (p.act == DIFFSAL_ACT_RELU) x = fmaxf(x, 0.f);
      else if (p.act == DIFFSAL_ACT_GELU_ERF) x = gelu_erf(x);
      else if (p.act == DIFFSAL_ACT_SIGMOID) x = sigmoidf_(x);
      if (p.act == DIFFSAL_ACT_GELU_GRAD) x *= gelu_erf_grad(p.residual[o + j]);
      else if (p.residual) x += p.residual[o + j];
      v[j] = x;
    }
    st4(p.out + o, make_float4(v[0], v[1], v[2], v[3]));
  }
}

struct TileCfg { int bm, bn, occ; float eff; };  // occ = co-resident workgroups per CU (registers / LDS)
// order must match the dispatch switch in diffsal_conv_igemm
static const TileCfg kCfgs[] = {{128, 192, 1, 0.95f}, {128, 128, 2, 0.95f}, {128, 96, 2, 0.95f},
                                {64, 128, 2, 0.90f},  {128, 64, 2, 0.90f},  {64, 64, 4, 0.85f}};
constexpr int kNumCfgs = 6;  // (a 256x96 tile with 64x96 per wave and one wave per SIMD measured 5-20 % slower)
constexpr int kCUs = 256;

struct Plan { int cfg, splits; double t; };   // t: the model's time estimate (seconds)

// Pick tile shape and split-K factor with a small analytic model.  The 256 CUs each hold `occ`
// workgroups; co-resident workgroups share the CU's matrix pipes, and every workgroup also has a
// fixed, pipe-idle latency (prologue loads, store tail) that only the other residents can cover:
//   round = max(occ * t_mfma, t_mfma + t_fixed),  time = ceil(workgroups / (256 occ)) * round
// (+ the slab round trip for split-K).  Short-K GEMMs therefore prefer more, smaller residents;
// long-K convolutions prefer the widest tile that still fills the chip.
static bool is_linear(const diffsal_conv_desc* d) {
  return d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_t == 0 && d->pad_l == 0 && d->Ho == d->H &&
         d->Wo == d->W;
}

static Plan choose_plan(long M, int Cout, int K, int precision, bool linear = false) {
  // the bf16x3 loop sustains ~2x the fp32 one on large tiles: split-K slabs and fixed latencies weigh twice as much
  const double mac_per_s_cu = (precision == DIFFSAL_PREC_BF16X3 ? 2.0 : 1.0) * 157.3e12 / 2.0 / kCUs;
  const int KT = K / BK;
  Plan best{5, 1, 1e30};
  double best_t = 1e30;
  for (int c = 0; c < kNumCfgs; ++c) {
    const TileCfg& t = kCfgs[c];
    if (t.bn > Cout && t.bn - Cout >= 32 && c != 5) continue;  // mostly-empty N tile
    const long tiles = ((M + t.bm - 1) / t.bm) * ((Cout + t.bn - 1) / t.bn);
    // pipe-idle time per workgroup: first-load latency (~4 us) + the store tail, which scales with the tile (stamps: 1.5-3.3 us
    // prologue, 4.7 us epilogue for 128x96)
    const double t_fixed = 4e-6 + 6e-6 * (t.bm * t.bn) / (128.0 * 96.0);
    for (int S = 1; S <= 16; S *= 2) {
      if (S > 1 && (KT / S < 6 || Cout % 4 != 0)) break;
      const long wgs = tiles * S;
      const long slots = static_cast<long>(kCUs) * t.occ;
      const int kt_per = (KT + S - 1) / S;
      // plain products (1x1, the persistent kernel): measured per-CU rates of the large tiles with two residents are ~0.85 of
      // what the table says relative to four residents of 64x64 (tools/tune_igemm_thin.py: K <= 768 token and tap GEMMs are
      // 5-17 % faster on 64x64; K14's M = 28560, K = 768, N = 864 stays on 128x96 by 3 %)
      const double eff = (linear && S == 1 && precision != DIFFSAL_PREC_BF16X3 && c != 5) ? t.eff * 0.86 : t.eff;
      const double t_mfma = static_cast<double>(t.bm) * t.bn * (kt_per * BK) / (mac_per_s_cu * eff);
      // full rounds at full residency, then the remainder at ITS residency: the last workgroups of a launch have the CU
      // (almost) to themselves and finish sooner than a full round (in-kernel stamps)
      auto round_time = [&](double resident) {
        return resident * t_mfma > t_mfma + t_fixed ? resident * t_mfma : t_mfma + t_fixed;
      };
      const long full = wgs / slots, rest = wgs - full * slots;
      double tt = full * round_time(t.occ);
      if (rest > 0) tt += round_time(static_cast<double>((rest + kCUs - 1) / kCUs));
      if (S > 1) tt += (S + 1.0) * M * Cout * 4.0 / 3.0e12 + 4.0e-6;
      if (tt < best_t) { best_t = tt; best = Plan{c, S, tt}; }
    }
  }
  return best;
}

template <int WM, int WN, int TM, int TN>
static int launch(IgemmArgs& a, hipStream_t s, int precision) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  a.n_tiles_n = (a.Cout + BN - 1) / BN;
  const int tiles_m = (a.M + BM - 1) / BM;
  a.n_tiles = a.n_tiles_n * tiles_m;
  const unsigned nz = a.pair ? 2u : 1u;
  if (precision != DIFFSAL_PREC_BF16X3 && a.linear && a.splits == 1 && a.vec_epilogue && a.persist_wgs > 0 && !a.pair) {
    const int grid = a.n_tiles < a.persist_wgs ? a.n_tiles : a.persist_wgs;
    const bool no_xcd = tune(TUNE_NO_XCD_ORDER) == 1;
    a.xcd_order = (!no_xcd && grid % 8 == 0 && a.n_tiles >= a.persist_wgs && a.n_tiles_n > 1 && tiles_m >= 16) ? 1 : 0;
    hipLaunchKernelGGL((igemm_linear_kernel<WM, WN, TM, TN>), dim3(grid), dim3(256), 0, s, a);
    note_kernel("igemm_linear_kernel<%d, %d, %d, %d> [%dx%d tile]", WM, WN, TM, TN, BM, BN);
    return check_launch("diffsal_conv_igemm(linear)");
  }
  if (precision == DIFFSAL_PREC_BF16X3)
    hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, 1>), dim3(a.n_tiles, a.splits, nz), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, 0>), dim3(a.n_tiles, a.splits, nz), dim3(256), 0, s, a);
  note_kernel("igemm_kernel<%d, %d, %d, %d, %d> [%dx%d tile, split-K %d%s]", WM, WN, TM, TN, precision == DIFFSAL_PREC_BF16X3 ? 1 : 0, BM, BN,
              a.splits, a.pair ? ", pair" : "");
  int rc = check_launch("diffsal_conv_igemm");
  if (rc || a.splits == 1) return rc;
  const long total4 = static_cast<long>(a.M) * (a.Cout / 4);
  long g = (total4 + 255) / 256;
  g = g > 2048 ? 2048 : g;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(static_cast<int>(g), nz), dim3(256), 0, s, a);
  return check_launch("diffsal_conv_igemm(split-K reduce)");
}

}  // namespace diffsal

using namespace diffsal;

// Which LDS-DMA tile configuration (gemm_dma.hip) takes this fp32 launch, -1: none.
//  * plain products with K a multiple of 96 whose tile grid (x K split) fills the chip.  Measured (tools/bench_gemm_dma.py, B = 4
//    shapes): M = 3024: 1.35x, M = 12096: 1.1-1.2x, M = 48384: 1.0-1.05x against the tiled kernels; at one clip per step
//    (M = 756 .. 12096) 1.1-1.4x with its K split, but 0.8x on M = 756 x N = 3456 (288 tiles on 256 CUs).
//    DIFFSAL_GEMM_DMA=0 switches it off, 1..4 force one of its tile configurations on every shape it accepts;
//  * convolutions (taps displaced per K slice, padding by the DMA's range check).  Measured (tools/bench_conv_dma.py): the tiled
//    kernel already runs the large convolutions at 0.74-0.80 of the matrix peak (~0.9 of what the chip's clock under this load
//    allows), so the DMA form is within +-5 % of it; it wins where its tile grid fits better: many tiles with a short K
//    (UpEmbed-2 of stage 3: 275 -> 265 us, ReduceTemp of stage 3: 143 -> 136) and M <= ~400 rows with a long K, where its K
//    split fills the chip (Downsample 768: 61 -> 49 us, ReduceTemp of stage 0: 38 -> 35).  DIFFSAL_CONV_DMA=0 switches it
//    off, 1..4 force a configuration.
static int dma_route(const diffsal_conv_desc* d, bool linear, long M, int K, bool pair) {
  if (d->precision != DIFFSAL_PREC_FP32 || d->w_format != 0 || pair) return -1;
  if (d->dtype != DIFFSAL_F32) {
    // 16-bit storage: plain products with K a multiple of 192 (three 64-element slices per ring pass).  Measured on the token
    // GEMMs of a B = 4 step (tools/bench_gemm_dma.py, DMA_DTYPE=bf16): 12-19 us instead of 14-25 from ~192 tiles on (the
    // 16-bit products are bound by operand delivery and launch latency, not by the matrix pipe: the DMA ring has no
    // staging instructions to issue); below that the tiled kernel's small tiles win.  DIFFSAL_GEMM_DMA16=0 switches it off, 1
    // forces it on every shape it accepts.
    if (!linear || tune(TUNE_IGEMM16_CFG) >= 0 || K % 192 != 0 || d->Cout % 4 != 0) return -1;
    const int forced = tune(TUNE_GEMM_DMA16);
    if (forced == 0) return -1;
    if (forced > 0) return forced == 2 ? 3 : 0;
    return ((M + 95) / 96) * ((d->Cout + 95) / 96) >= 192 ? 0 : -1;
  }
  if (tune(TUNE_IGEMM_CFG) >= 0) return -1;
  const long tiles96 = ((M + 95) / 96) * ((d->Cout + 95) / 96);
  if (linear) {
    const int forced = tune(TUNE_GEMM_DMA);
    if (forced == 0) return -1;
    if (forced > 0) return forced - 1;
    if (K % 96 != 0 || d->Cout % 4 != 0) return -1;
    // both kernels priced by their models; the DMA kernel's (rounds of 256 CUs x K slices x 1.13 us + fill + slab trip) tracks
    // its measured times within ~10 %, the tiled planner's estimate is optimistic by 1.1-1.5x on these shapes (31 shapes of
    // the B = 1 and B = 4 steps, tools/bench_gemm_dma.py): the DMA kernel takes the launch when its estimate is within 1.25x
    const double t_dma = gemm_dma_estimate(0, M, K, d->Cout);
    const double t_tiled = choose_plan(M, d->Cout, K, d->precision, true).t;
    return t_dma <= 1.25 * t_tiled ? 0 : -1;
  }
  const int forced = tune(TUNE_CONV_DMA);
  if (forced == 0) return -1;
  if (forced > 0) return forced - 1;
  if (K % 96 != 0 || d->Cout % 4 != 0) return -1;
  return ((tiles96 >= 1024 && K <= 1024) || (M <= 400 && K >= 3072)) ? 0 : -1;
}

static int validate(const diffsal_conv_desc* d) {
  DS_REQUIRE(d, DIFFSAL_E_ARG, "conv_igemm: null descriptor");
  DS_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0, DIFFSAL_E_SHAPE, "conv_igemm: Cin=%d must be a multiple of 32", d->Cin);
  DS_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride_h > 0 &&
                 d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0,
             DIFFSAL_E_SHAPE, "conv_igemm: non-positive dimension");
  DS_REQUIRE(d->Ho > 0 && d->Wo > 0, DIFFSAL_E_SHAPE, "conv_igemm: empty output %dx%d", d->Ho, d->Wo);
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  DS_REQUIRE(M < (1L << 31) && M * d->Cout < (1L << 40), DIFFSAL_E_SHAPE, "conv_igemm: problem too large");
  DS_REQUIRE(d->KH * d->KW <= 32, DIFFSAL_E_SHAPE, "conv_igemm: at most 32 taps (KH*KW=%d)", d->KH * d->KW);
  DS_REQUIRE(d->precision == DIFFSAL_PREC_FP32 || d->precision == DIFFSAL_PREC_BF16X3, DIFFSAL_E_ARG,
             "conv_igemm: precision %d (DIFFSAL_PREC_FP32 or DIFFSAL_PREC_BF16X3)", d->precision);
  DS_REQUIRE(d->dtype == DIFFSAL_F32 || d->dtype == DIFFSAL_BF16 || d->dtype == DIFFSAL_F16, DIFFSAL_E_ARG,
             "conv_igemm: dtype %d (DIFFSAL_F32, DIFFSAL_BF16 or DIFFSAL_F16)", d->dtype);
  DS_REQUIRE(d->dtype == DIFFSAL_F32 || (d->precision == DIFFSAL_PREC_FP32 && d->w_format == 0), DIFFSAL_E_ARG,
             "conv_igemm: 16-bit storage uses the native MFMA (precision and w_format must be 0)");
  const long esz = d->dtype == DIFFSAL_F32 ? 4 : 2;
  const long in_bytes = static_cast<long>(d->N) * d->H * d->W * d->Cin * esz;
  const long w_bytes = static_cast<long>(d->Cout) * d->KH * d->KW * d->Cin * esz;
  DS_REQUIRE(in_bytes < (1L << 32) - 16 && w_bytes < (1L << 32) - 16, DIFFSAL_E_SHAPE,
             "conv_igemm: input (%ld B) and weight (%ld B) must each stay below 4 GiB (32-bit buffer offsets); "
             "split the batch", in_bytes, w_bytes);
  return DIFFSAL_OK;
}

// the planner's choice, or the tuning aid's: DIFFSAL_IGEMM_CFG = tile shape (value % 8) and 2^(value / 8) K splits
static Plan plan_or_forced(long M, int Cout, int K, int precision, bool linear) {
  Plan pl = choose_plan(M, Cout, K, precision, linear);
  const int v = tune(TUNE_IGEMM_CFG);
  if (v >= 0) {
    pl.cfg = (v % 8) % kNumCfgs;
    pl.splits = 1;
    for (int e = 0; e < v / 8 && e < 4; ++e)
      if (K / BK / (pl.splits * 2) >= 4 && Cout % 4 == 0) pl.splits *= 2;
  }
  return pl;
}

extern "C" size_t diffsal_conv_igemm_ws_bytes(const diffsal_conv_desc* d) {
  if (validate(d) != DIFFSAL_OK) return 0;
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  const int K = d->KH * d->KW * d->Cin;
  size_t need;
  if (d->dtype != DIFFSAL_F32) {
    need = igemm16_ws_bytes(d);
  } else {
    const Plan pl = plan_or_forced(M, d->Cout, K, d->precision, is_linear(d));
    need = pl.splits > 1 ? static_cast<size_t>(pl.splits) * M * d->Cout * sizeof(float) : 0;
  }
  const int r = dma_route(d, is_linear(d), M, K, false);
  if (r >= 0) {
    const size_t nd = gemm_dma_ws_bytes(r, M, K, d->Cout, d->dtype == DIFFSAL_F32 ? 4 : 2);
    need = nd > need ? nd : need;
  }
  return need;
}

// second problem of a pair launch (same descriptor): diffsal_linear_pair
struct PairExtra {
  const void* in2;
  const void* w2;
  const float* bias2;
  void* out2;
};

static int conv_igemm_impl(const diffsal_conv_desc* d, const void* in_v, const void* w_v, const float* bias, const float* scale,
                           const float* shift, const float* rowvec, const void* residual_v, void* out_v, void* ws,
                           size_t ws_bytes, diffsal_stream_t stream, const PairExtra* px) {
  int rc = validate(d);
  if (rc) return rc;
  DS_REQUIRE(in_v && w_v && out_v, DIFFSAL_E_ARG, "conv_igemm: null argument");
  DS_REQUIRE((scale == nullptr) == (shift == nullptr), DIFFSAL_E_ARG, "conv_igemm: scale and shift go together");
  DS_REQUIRE(aligned16(in_v) && aligned16(w_v), DIFFSAL_E_ALIGN, "conv_igemm: in/w must be 16-byte aligned");
  DS_REQUIRE((d->act >= DIFFSAL_ACT_NONE && d->act <= DIFFSAL_ACT_SIGMOID) ||
                 (d->act == DIFFSAL_ACT_GELU_GRAD && d->dtype == DIFFSAL_F32 && residual_v && !px),
             DIFFSAL_E_ARG, "conv_igemm: act=%d (DIFFSAL_ACT_GELU_GRAD: fp32 only, the pre-activation goes in `residual`)", d->act);
  if (d->dtype != DIFFSAL_F32 && !px && d->precision == DIFFSAL_PREC_FP32 && d->w_format == 0 && d->act != DIFFSAL_ACT_GELU_GRAD) {
    // plain products and ReduceTemp's row form
    const int rr = try_gemm16_dma2(d, in_v, w_v, bias, scale, shift, rowvec, d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout, residual_v, out_v,
                                   static_cast<hipStream_t>(stream), false);
    if (rr != 0) return rr < 0 ? rr : DIFFSAL_OK;
  }
  if (d->dtype != DIFFSAL_F32) {
    const long M16 = static_cast<long>(d->N) * d->Ho * d->Wo;
    const int r16 = dma_route(d, is_linear(d), M16, d->KH * d->KW * d->Cin, px != nullptr);
    if (r16 >= 0 && (!rowvec || d->rowvec_ld % 4 == 0)) {
      const int rr = try_gemm_dma(r16, d, false, in_v, w_v, bias, scale, shift, rowvec, d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout, residual_v,
                                  out_v, ws, ws_bytes, static_cast<hipStream_t>(stream));
      if (rr != 0) return rr < 0 ? rr : DIFFSAL_OK;
    }
  }
  if (d->dtype != DIFFSAL_F32)
    return igemm16_launch(d, in_v, w_v, bias, scale, shift, rowvec, residual_v, out_v, ws, ws_bytes,
                          static_cast<hipStream_t>(stream), px ? px->in2 : nullptr, px ? px->w2 : nullptr,
                          px ? px->bias2 : nullptr, px ? px->out2 : nullptr);
  const float* in = static_cast<const float*>(in_v);
  const float* w = static_cast<const float*>(w_v);
  const float* residual = static_cast<const float*>(residual_v);
  float* out = static_cast<float*>(out_v);
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;

  IgemmArgs a;
  a.xcd_order = 0;
  a.pair = px ? 1 : 0;
  a.in2 = px ? static_cast<const float*>(px->in2) : nullptr;
  a.w2 = px ? static_cast<const float*>(px->w2) : nullptr;
  a.bias2 = px ? px->bias2 : nullptr;
  a.out2 = px ? static_cast<float*>(px->out2) : nullptr;
  a.partial2 = nullptr;
  a.in = in; a.w = w; a.bias = bias; a.scale = scale; a.shift = shift; a.rowvec = rowvec;
  a.residual = residual; a.out = out;
  a.M = static_cast<int>(M);
  a.K = d->KH * d->KW * d->Cin;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KW = d->KW; a.taps = d->KH * d->KW; a.stride_h = d->stride_h; a.stride_w = d->stride_w;
  a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.dil_h = d->dil_h; a.dil_w = d->dil_w; a.act = d->act;
  a.rowvec_ld = d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout;
  a.linear = d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_t == 0 && d->pad_l == 0 && d->Ho == d->H && d->Wo == d->W;
  DS_REQUIRE(d->w_format == 0 || (d->w_format == 1 && d->precision == DIFFSAL_PREC_BF16X3), DIFFSAL_E_ARG,
             "conv_igemm: w_format=%d needs precision = DIFFSAL_PREC_BF16X3 in the descriptor", d->w_format);
  a.w_split = d->w_format;
  a.in_bytes = static_cast<unsigned>(static_cast<long>(d->N) * d->H * d->W * d->Cin * 4);
  a.w_bytes = static_cast<unsigned>(static_cast<long>(d->Cout) * a.K * 4);
  hipStream_t s = static_cast<hipStream_t>(stream);

  if (d->w_format == 0 && d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_t == 0 && d->pad_l == 0 &&
      d->Ho == d->H && d->Wo == d->W && !scale && !rowvec && aligned16(out) && (!residual || aligned16(residual)) && !px &&
      d->act != DIFFSAL_ACT_GELU_GRAD) {
    const int r = try_linear_stream(in, w, bias, residual, out, M, d->Cin, d->Cout, d->act, s);
    if (r > 0) note_kernel("lin_stream_kernel [K=%d, N=%d]", d->Cin, d->Cout);
    if (r != 0) return r < 0 ? r : DIFFSAL_OK;
  }
  {
    const int r = dma_route(d, a.linear != 0, M, a.K, px != nullptr);
    if (r >= 0) {
      const int rr = try_gemm_dma(r, d, !a.linear, in, w, bias, scale, shift, rowvec, a.rowvec_ld, residual, out, ws, ws_bytes, s);
      if (rr != 0) return rr < 0 ? rr : DIFFSAL_OK;
    }
  }
  const Plan pl = plan_or_forced(M, d->Cout, a.K, d->precision, a.linear != 0);
  if (tune(TUNE_PLAN_DEBUG) == 1) {   // tuning aid: which tile shape / split the planner chose
    fprintf(stderr, "[diffsal plan] M=%ld K=%d N=%d linear=%d -> %dx%d splits=%d\n", M, a.K, d->Cout, a.linear,
                             kCfgs[pl.cfg].bm, kCfgs[pl.cfg].bn, pl.splits);
  }
  a.splits = pl.splits;
  a.kt_per_split = (a.K / BK + pl.splits - 1) / pl.splits;
  a.partial = nullptr;
  if (pl.splits > 1) {
    const size_t one = static_cast<size_t>(pl.splits) * M * d->Cout * sizeof(float);
    const size_t need = px ? 2 * one : one;
    DS_REQUIRE(ws && ws_bytes >= need && aligned16(ws) && aligned16(out), DIFFSAL_E_ARG,
               "conv_igemm: split-K needs %zu bytes of 16-byte aligned workspace (diffsal_conv_igemm_ws_bytes), got %zu",
               need, ws_bytes);
    a.partial = static_cast<float*>(ws);
    a.partial2 = px ? a.partial + one / sizeof(float) : nullptr;
  }
  a.vec_epilogue = d->Cout % 4 == 0 && aligned16(out) && (!residual || aligned16(residual)) && (!bias || aligned16(bias)) &&
                   (!scale || (aligned16(scale) && aligned16(shift))) && (!rowvec || (aligned16(rowvec) && a.rowvec_ld % 4 == 0));
  if (px) a.vec_epilogue = a.vec_epilogue && aligned16(px->out2) && (!px->bias2 || aligned16(px->bias2));
  a.persist_wgs = kCUs * kCfgs[pl.cfg].occ;
  if (tune(TUNE_NO_PERSIST) == 1) a.persist_wgs = 0;
  switch (pl.cfg) {
    case 0: return launch<2, 2, 2, 3>(a, s, d->precision);
    case 1: return launch<2, 2, 2, 2>(a, s, d->precision);
    case 2: return launch<4, 1, 1, 3>(a, s, d->precision);
    case 3: return launch<2, 2, 1, 2>(a, s, d->precision);
    case 4: return launch<2, 2, 2, 1>(a, s, d->precision);
    default: return launch<2, 2, 1, 1>(a, s, d->precision);
  }
}

extern "C" int diffsal_conv_igemm(const diffsal_conv_desc* d, const void* in_v, const void* w_v,
                                  const float* bias, const float* scale, const float* shift,
                                  const float* rowvec, const void* residual_v, void* out_v, void* ws,
                                  size_t ws_bytes, diffsal_stream_t stream) {
  return conv_igemm_impl(d, in_v, w_v, bias, scale, shift, rowvec, residual_v, out_v, ws, ws_bytes, stream, nullptr);
}

extern "C" int diffsal_conv_igemm_group(int n, const diffsal_conv_desc* const* descs, const void* const* in, const void* const* w,
                                        const float* const* bias, void* const* out, void* ws, size_t ws_bytes, diffsal_stream_t stream) {
  DS_REQUIRE(n >= 1 && n <= 4 && descs && in && w && out, DIFFSAL_E_ARG, "conv_igemm_group: 1..4 problems, non-null tables");
  bool dma_ok = tune(TUNE_GEMM_DMA) != 0 && tune(TUNE_IGEMM_CFG) < 0;
  for (int i = 0; i < n; ++i) {
    const int rc = validate(descs[i]);
    if (rc) return rc;
    DS_REQUIRE(in[i] && w[i] && out[i], DIFFSAL_E_ARG, "conv_igemm_group: null operand of problem %d", i);
    DS_REQUIRE(descs[i]->act >= DIFFSAL_ACT_NONE && descs[i]->act <= DIFFSAL_ACT_SIGMOID, DIFFSAL_E_ARG, "conv_igemm_group: act=%d", descs[i]->act);
    dma_ok = dma_ok && descs[i]->dtype == DIFFSAL_F32 && descs[i]->precision == DIFFSAL_PREC_FP32 && descs[i]->w_format == 0;
  }
  if (dma_ok) {
    const float* a_[4]; const float* w_[4]; const float* b_[4]; float* o_[4];
    for (int i = 0; i < n; ++i) {
      a_[i] = static_cast<const float*>(in[i]); w_[i] = static_cast<const float*>(w[i]);
      b_[i] = bias ? bias[i] : nullptr; o_[i] = static_cast<float*>(out[i]);
    }
    const int r = try_gemm_dma_group(n, descs, a_, w_, b_, o_, static_cast<hipStream_t>(stream));
    if (r != 0) return r < 0 ? r : DIFFSAL_OK;
  }
  for (int i = 0; i < n; ++i) {     // some problem does not fit the grouped kernel: one launch each (same results)
    const int rc = conv_igemm_impl(descs[i], in[i], w[i], bias ? bias[i] : nullptr, nullptr, nullptr, nullptr, nullptr, out[i], ws, ws_bytes,
                                   stream, nullptr);
    if (rc) return rc;
  }
  return DIFFSAL_OK;
}

/* 16-bit storage in, fp32 out: out [M, N] (fp32) = act(in [M, K] x w [N, K]^T + bias) for a plain product described by d (KH = KW = 1,
 * dtype bf16 / f16): the sums leave the matrix cores without a rounding to the storage type -- for consumers that add many of them
 * (the nine tap products of mt_proj under the 4-scale interpolation, R/models/saliency_decoder/sal_unet.py:480-489).  Runs on
 * gemm_dma_kernel only: K % 192 == 0, N % 4 == 0; DIFFSAL_E_SHAPE otherwise. */
extern "C" int diffsal_linear_f32out(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, float* out,
                                     diffsal_stream_t stream) {
  DS_REQUIRE(d && in && w && out, DIFFSAL_E_ARG, "linear_f32out: null argument");
  int rc = validate(d);
  if (rc) return rc;
  DS_REQUIRE(is_linear(d) && d->dtype != DIFFSAL_F32 && d->precision == DIFFSAL_PREC_FP32 && d->w_format == 0 &&
                 d->act >= DIFFSAL_ACT_NONE && d->act <= DIFFSAL_ACT_SIGMOID,
             DIFFSAL_E_ARG, "linear_f32out: a plain product on bf16 / f16 storage");
  DS_REQUIRE(aligned16(in) && aligned16(w) && aligned16(out) && (!bias || aligned16(bias)), DIFFSAL_E_ALIGN, "linear_f32out: misaligned pointer");
  int rr = try_gemm16_dma2(d, in, w, bias, nullptr, nullptr, nullptr, d->Cout, nullptr, out, static_cast<hipStream_t>(stream), true);
  if (rr == 0)
    rr = try_gemm_dma(0, d, false, in, w, bias, nullptr, nullptr, nullptr, d->Cout, nullptr, out, nullptr, 0,
                      static_cast<hipStream_t>(stream), true);
  if (rr < 0) return rr;
  DS_REQUIRE(rr == 1, DIFFSAL_E_SHAPE, "linear_f32out: K=%d must be a multiple of 192 and N=%d of 4", d->KH * d->KW * d->Cin, d->Cout);
  return DIFFSAL_OK;
}

extern "C" int diffsal_linear_pair(const diffsal_conv_desc* d, const void* in0, const void* in1, const void* w0, const void* w1,
                                   const float* bias0, const float* bias1, void* out0, void* out1, void* ws, size_t ws_bytes,
                                   diffsal_stream_t stream) {
  DS_REQUIRE(d && in0 && in1 && w0 && w1 && out0 && out1, DIFFSAL_E_ARG, "linear_pair: null argument");
  DS_REQUIRE(d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_t == 0 && d->pad_l == 0 &&
                 d->Ho == d->H && d->Wo == d->W && d->act == DIFFSAL_ACT_NONE && d->w_format == 0 && d->precision == DIFFSAL_PREC_FP32,
             DIFFSAL_E_SHAPE, "linear_pair: two plain [M,K] x [K,N] products of one shape (1x1, no activation, exact arithmetic)");
  DS_REQUIRE((bias0 == nullptr) == (bias1 == nullptr), DIFFSAL_E_ARG, "linear_pair: both products carry a bias or neither does");
  DS_REQUIRE(aligned16(in1) && aligned16(w1), DIFFSAL_E_ALIGN, "linear_pair: in/w must be 16-byte aligned");
  const PairExtra px{in1, w1, bias1, out1};
  return conv_igemm_impl(d, in0, w0, bias0, nullptr, nullptr, nullptr, nullptr, out0, ws, ws_bytes, stream, &px);
}
