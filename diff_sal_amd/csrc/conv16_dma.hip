// 3x3 stride-1 convolutions (dilation 1 or 2, any padding) on bf16 / fp16 storage: LDS halo patch staged by LDS-DMA, two
// workgroups per CU.
//
// conv16_halo.hip holds one 8-wavefront workgroup per CU (161 KB of LDS at dilation 2): its prologue (the first patch), its
// register-staged refills and its store tail are exposed -- at 64 clips the matrix pipe is 0.29-0.33 busy and the UpEmbed /
// ResnetBlock convolutions (R/models/saliency_decoder/common_block.py:196-216, sal_unet.py:104-142) run at 0.53-0.74 PF/s, although
// the stage-3 convolutions move so few bytes per pixel that they are bound by HBM, not by the matrix pipe, once those phases
// overlap.  This kernel is built so that they do:
//
//  * a workgroup is 4 wavefronts and 256 output pixels (8 x 32 or 16 x 16) x 96 output channels; its LDS -- two patch buffers of
//    one 32-channel chunk, a four-slot ring of single-tap weight slices, 80 KB -- leaves room for a SECOND workgroup on the CU:
//    one's prologue, barriers and epilogue run under the other's matrix work;
//  * patch and weights go memory -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write); a pixel / weight
//    row is 64 bytes with its four 16-byte slots XOR-swizzled by (index >> 2) & 3 -- applied on the SOURCE address -- so that the
//    sixteen rows of a ds_read_b128 pass cover all 64 banks without padding; padding pixels and rows past Cout are fetched with an
//    out-of-range offset (the DMA writes zeros);
//  * a step is ONE tap of one chunk (12 MFMAs 32x32x16 per wavefront): wait for the pieces issued two steps ago, one barrier,
//    issue the weight slice two taps ahead and a share of the next chunk's patch (every wavefront issues exactly three DMA
//    instructions per step, dead ones into a scratch KiB, so one counted vmcnt serves every step), multiply;
//  * the epilogue goes through LDS: a wavefront parks its 32 pixels x 96 channels of fp32 sums, reads them back a pixel row at a
//    time and stores 16 bytes per lane -- whole 192-byte pixels, a wavefront's 32 pixels contiguous -- with the per-channel
//    affine, per-image vector, activation and residual applied in the same order as the other 16-bit kernels.
//
// Accumulation order (chunk, tap, two 16-channel halves, fp32 MFMA accumulation) is that of igemm16.hip / conv16_halo.hip:
// results are bit-identical to them.  Weight layout: pack_conv_weight's [Cout][Cin / 32][9][32].
#include "common.h"

namespace diffsal {

typedef float cd_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cd_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cd_f16x8 __attribute__((ext_vector_type(8)));
typedef int cd_i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* cd_lds_ptr_t;

template <typename T> struct CdMma;
template <> struct CdMma<__bf16> {
  typedef cd_bf16x8 vec;
  static __device__ __forceinline__ cd_f32x16 run(vec a, vec b, cd_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct CdMma<_Float16> {
  typedef cd_f16x8 vec;
  static __device__ __forceinline__ cd_f32x16 run(vec a, vec b, cd_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T>
struct CdArgs {
  const T* in;
  const T* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const T* residual;
  T* out;
  int N, H, W, Ho, Wo, Cin, Cout, K;   // K = 9 * Cin
  int dil, pad, act, rowvec_ld;
  int tiles_x, tiles_y, tiles_n;
};

// inline assembly on purpose (see gemm_dma.hip): the compiler must know neither the LDS write nor the vmcnt event
__device__ __forceinline__ void cd_dma(unsigned lds_addr, unsigned voff, cd_i32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void cd_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int kCdWaves = 4;
constexpr int kCdRowB = 64;                     // bytes per staged pixel / weight row (one 32-channel chunk)

// Two tile shapes (the LDS of both is exactly 80 KB), four wavefronts, a wavefront = TM x 3 accumulators of 32 x 32:
//   WN = 1: 256 pixels (8 x 32 or 16 x 16) x 96 channels, the wavefronts stacked over the pixels (64 pixels x 96 channels each); two
//           28 KB patch buffers, a FOUR-slot ring of 6 KB weight slices;
//   WN = 2: 192 pixels (8 x 24) x 192 channels, the wavefronts 2 x 2 (96 pixels x 96 channels each): per MFMA a wavefront reads
//           (1 / 3 + 1 / 3) KiB of fragments from LDS instead of (1 / 2 + 1 / 3) -- with the DMA's own writes the LDS was as busy as the
//           matrix pipe (49 KB per 384 matrix cycles and workgroup at 128 B per cycle), which is what held these convolutions at 0.55 of
//           the pipe; now 62 KB per 576 -- and the 8 x 24 tile wastes 12.5 % of a 28 x 48 or 14 x 24 map where 8 x 32 wasted 34 %.
//           Two 22 KB patch buffers, a THREE-slot ring of 12 KB weight slices.
template <int WN> struct CdShape {
  static constexpr int TM = WN == 1 ? 2 : 3;
  static constexpr int PIX = (kCdWaves / WN) * TM * 32;              // 256 / 192
  static constexpr int BN = WN * 96;
  static constexpr int PATCH_ISS = WN == 1 ? 7 : 6;                  // patch DMA instructions per wavefront and chunk
  static constexpr int PATCH_CAP = WN == 1 ? 28672 : 22528;          // bytes of a patch buffer; its last KiB takes the dead DMAs
  static constexpr int SLOTS = WN == 1 ? 4 : 3;
  static constexpr int SLOT_B = BN * kCdRowB;
  static constexpr int W_ISS = (BN / 16 + kCdWaves - 1) / kCdWaves;  // weight DMA instructions per wavefront and step: 2 / 3
  static constexpr int LDS = 2 * PATCH_CAP + SLOTS * SLOT_B;         // 81920 both
};

// TW: tile width in pixels (WN = 1: 32 -- 8 x 32 tile -- or 16 -- 16 x 16; WN = 2: 24 -- 8 x 24).  TW = 0 (WN = 1): maps of at most 128
// output pixels (the 9 x 14 extended grid of a 7 x 12 map: a 256-pixel tile of one image would be half padding) -- a tile is TWO WHOLE
// IMAGES, 128 tile rows each, the patch buffer holds both padded images
// HALF (WN = 1 only): a 128-pixel tile (8 x 16 or 4 x 32; a wavefront = 32 pixels x 96 channels) for launches whose 256-pixel tiles would
// leave most CUs without a workgroup (B <= 4 maps: 96-170 workgroups).  A CU that has a workgroup is already 0.6 busy on the matrix pipe
// there (a step of 12 MFMAs per SIMD takes 0.29 us; eight wavefronts per workgroup or a deeper DMA queue change nothing: measured) --
// what is idle is the other CUs, so the tiles are halved and twice as many CUs work.
template <int TW, int WN, typename T, bool HALF = false>
__global__ __launch_bounds__(256, 2) void conv16_dma_kernel(CdArgs<T> p) {
  typedef typename CdMma<T>::vec vec;
  typedef CdShape<WN> S;
  static_assert(!HALF || (WN == 1 && TW != 0), "half tiles: 128 pixels x 96 channels");
  constexpr int TM = HALF ? 1 : S::TM, TN = 3, BN = S::BN;
  constexpr bool MI = TW == 0;
  constexpr int TWX = MI ? 1 : TW;
  constexpr int TH = (HALF ? S::PIX / 2 : S::PIX) / TWX;
  constexpr int kCdPatchIss = S::PATCH_ISS;
  static_assert(S::PIX % TWX == 0 && S::LDS <= 81920 && (!MI || WN == 1), "tile");
  constexpr unsigned DEAD = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char cd_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b;
  {  // XCD-aware order: an XCD takes a contiguous run of tiles (neighbouring patches share halo rows, N tiles share the patch)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tn = b % p.tiles_n; b /= p.tiles_n;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int img = MI ? (b / p.tiles_y) * 2 : b / p.tiles_y;        // MI: the first of the tile's two images (tiles_x = tiles_y = 1)
  const int y0 = MI ? 0 : ty * TH, x0 = MI ? 0 : tx * TW, n0 = tn * BN;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int d = p.dil;
  const int PH = (MI ? p.Ho : TH) + 2 * d, PW = (MI ? p.Wo : TW) + 2 * d;
  const int PP = PH * PW, OP = p.Ho * p.Wo;        // MI: pixels of a padded image / of an output map
  const int patch_bytes = (MI ? 2 : 1) * PP * kCdRowB;       // <= 27648
  // LDS: [patch 0][patch 1][weight slot 0..2][scratch KiB]
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((cd_lds_ptr_t)cd_smem));
  constexpr int PATCH_CAP = S::PATCH_CAP;
  constexpr int SLOT_B = S::SLOT_B;
  // dead DMA instructions write zeros into the last KiB of patch buffer 0, which no patch reaches (host-checked)
  const unsigned lds_w = lds0 + 2 * PATCH_CAP, lds_scratch = lds0 + PATCH_CAP - 1024;
  const int n_chunks = p.Cin >> 5;
  const int G = 9 * n_chunks;

  // ---- issue side.  Patch instruction q of a wavefront (q < 7) covers pieces idx = (q * 4 + wave) * 64 + lane: pixel idx >> 2,
  // physical slot idx & 3, which receives logical slot (idx & 3) ^ ((pixel >> 2) & 3).  Weight instruction q (q < 2): rows
  // (q * 4 + wave) * 16 + (lane >> 2).
  const T* in_img = p.in + static_cast<long>(img) * p.H * p.W * p.Cin;
  const unsigned long pa = reinterpret_cast<unsigned long>(in_img);
  const int a_records = __builtin_amdgcn_readfirstlane((MI && img + 1 < p.N ? 2 : 1) * p.H * p.W * p.Cin * 2);
  const cd_i32x4 rs_a = cd_i32x4{static_cast<int>(pa), static_cast<int>(pa >> 32) & 0xFFFF, a_records, 0x00020000};
  const unsigned long pw = reinterpret_cast<unsigned long>(p.w);
  const cd_i32x4 rs_w = cd_i32x4{static_cast<int>(pw), static_cast<int>(pw >> 32) & 0xFFFF, p.Cout * p.K * 2, 0x00020000};
  unsigned a_voff[kCdPatchIss], w_voff[S::W_ISS];
#pragma unroll
  for (int q = 0; q < kCdPatchIss; ++q) {
    const int idx = (q * kCdWaves + wave) * 64 + lane;
    const int pix = idx >> 2, ls = (idx & 3) ^ ((pix >> 2) & 3);
    const int pj = MI ? pix / PP : 0, pl = pix - pj * PP;             // MI: image of the tile, pixel of its padded map
    const int pr = pl / PW, pc = pl - pr * PW;
    const int gy = y0 - p.pad + pr, gx = x0 - p.pad + pc;
    const bool ok = pix < (MI ? 2 : 1) * PP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;        // (an image past N: past the descriptor's range)
    a_voff[q] = ok ? static_cast<unsigned>((((pj * p.H + gy) * p.W + gx) * p.Cin + ls * 8) * 2) : DEAD;
  }
#pragma unroll
  for (int q = 0; q < S::W_ISS; ++q) {
    const int row = (q * kCdWaves + wave) * 16 + (lane >> 2), ls = (lane & 3) ^ ((row >> 2) & 3);
    const bool ok = row < BN && n0 + row < p.Cout;
    w_voff[q] = ok ? static_cast<unsigned>(((n0 + row) * p.K + ls * 8) * 2) : DEAD;
  }
  // patch instruction q of chunk c -> buffer c & 1 (instructions past the patch write zeros into their own KiB of the 28 KiB buffer)
  auto issue_patch = [&](int c, int q) __attribute__((always_inline)) {
    const unsigned dst = lds0 + (c & 1) * PATCH_CAP + (q * kCdWaves + wave) * 1024;
    const bool live = c < n_chunks && (q * kCdWaves + wave) * 1024 < patch_bytes + 1024 && (q * kCdWaves + wave) * 1024 < PATCH_CAP;
    cd_dma(live ? dst : lds_scratch, live ? a_voff[q] : DEAD, rs_a, static_cast<unsigned>(c) * 64u);
  };
  auto issue_weights = [&](int g) __attribute__((always_inline)) {       // the slice of step g (chunk g / 9, tap g % 9) -> slot g % SLOTS
    const bool live = g < G;
    const unsigned dst = lds_w + (g % S::SLOTS) * SLOT_B;
    const unsigned soff = static_cast<unsigned>(g) * 64u;      // [Cin / 32][9][32] inside a weight row: step g is 64 bytes further
#pragma unroll
    for (int q = 0; q < S::W_ISS; ++q) {
      const bool in_tile = (q * kCdWaves + wave) * 16 < BN;    // 96 rows: the second instruction of wavefronts 2, 3 has none (scratch)
      cd_dma(live && in_tile ? dst + (q * kCdWaves + wave) * 1024 : lds_scratch, live && in_tile ? w_voff[q] : DEAD, rs_w, soff);
    }
  };

  // ---- fragment addressing: lane -> pixel lp of its MFMA row tile, k half kh; logical slot of (kk, kh) = 2 kk + kh
  const int lp = lane & 31, kh = lane >> 5;
  int a_pix[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int t = (wm * TM + i) * 32 + lp;         // pixel of the workgroup's tile, row-major over TH x TW
    if constexpr (MI) {                            // tile row = (image t / 128, pixel t % 128 of its map; rows past the map multiply pixel 0)
      const int tj = t >> 7, tl = (t & 127) < OP ? (t & 127) : 0;
      a_pix[i] = tj * PP + (tl / p.Wo) * PW + tl % p.Wo;
    } else {
      a_pix[i] = (t / TWX) * PW + t % TWX;
    }
  }
  int b_off[TN][2];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * 96 + j * 32 + lp;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_off[j][kk] = row * kCdRowB + (((kk * 2 + kh) ^ ((row >> 2) & 3)) << 4);
  }

  // ---- prologue: patch 0, weight slices 0 and 1
#pragma unroll
  for (int q = 0; q < kCdPatchIss; ++q) issue_patch(0, q);
  issue_weights(0);
  issue_weights(1);

  cd_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int g = 0;
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const unsigned char* Ab = cd_smem + (chunk & 1) * PATCH_CAP;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++g) {
      // everything this wavefront issued up to step g - 2 has landed (first step: all but weight slice 1); after the barrier
      // everybody's has.  What is issued now lands in buffers last read TWO steps ago -- weight slot (g + 2) % 4, and the other patch
      // buffer from tap 1 on -- so that a fragment read of step g - 1 that is still queued in a busy LDS (hipcc sinks that step's
      // last MFMAs and their lgkmcnt waits below this barrier) cannot meet a DMA: with a three-slot ring and lookahead 2 that
      // happened -- rare wrong tiles with every CU loaded -- and the cure there, lgkmcnt(0) in front of the barrier, cost 5 %
      // lgkmcnt(10): at most the ten fragment reads of step g - 1 are still on their way -- those of step g - 2, whose buffers the DMAs
      // below overwrite, have returned whatever the compiler did with that step's MFMAs (LDS operations return in order)
      if constexpr (WN == 1) {
        if (g == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(10)" ::: "memory");
      } else {
        // three weight slots: what is issued below lands in the slot read in step g - 1, so every fragment read of that step has to
        // have returned before the barrier (lgkmcnt(0)); four DMA instructions per wavefront and step (three weight rows, one patch)
        if (g == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      issue_weights(g + 2);
      constexpr int P0 = WN == 1 ? 1 : 0;          // first tap that refills the other patch buffer (WN = 1: reuse distance of two steps)
      if (tap >= P0 && tap < P0 + kCdPatchIss) issue_patch(chunk + 1, tap - P0);
      else cd_dma(lds_scratch, DEAD, rs_a, 0u);
      const unsigned char* Bb = cd_smem + 2 * PATCH_CAP + (g % S::SLOTS) * SLOT_B;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int toff = ky * d * PW + kx * d;
      if constexpr (WN == 2) {
        // keep the per-tap fragment addresses from being hoisted out of the chunk loop (9 taps x 3 row tiles x 2 halves of them: the
        // kernel spilled); recomputed per step they cost a few VALU instructions under the MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(a_pix[i]));
      }
      vec fa[2][TM], fb[2][TN];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int px = a_pix[i] + toff;
          fa[kk][i] = *reinterpret_cast<const vec*>(Ab + px * kCdRowB + (((kk * 2 + kh) ^ ((px >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[kk][j] = *reinterpret_cast<const vec*>(Bb + b_off[j][kk]);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = CdMma<T>::run(fb[kk][j], fa[kk][i], acc[i][j]);   // D^T: rows = channels, cols = pixels
    }
  }
  // every DMA (the dead ones of the last steps included) has landed before the LDS is reused
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (and this wavefront's last fragment reads have returned: the staging below reuses the patch buffers)
  __builtin_amdgcn_s_barrier();

  // ---- epilogue through LDS.  A lane holds, for pixel lp of row tile i, channels (r & 3) + 8 (r >> 2) + 4 kh of every 32-channel
  // block j.  Per row tile: park [32 pixels][96 + 4] fp32 in the wavefront's own 12.8 KB, read back (pixel, channel octet) items.
  float* stage = reinterpret_cast<float*>(cd_smem) + wave * (32 * 100);
  const T* __restrict__ resid = p.residual;
  T* __restrict__ outp = p.out;
  const float* rv_row = p.rowvec ? p.rowvec + static_cast<long>(img) * p.rowvec_ld : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // the residual octets of this row tile are requested before the sums are parked (a round trip under the LDS traffic)
    uint4 rraw[6];
    if (resid) {
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int item = it * 64 + lane;
        const int px = item / 12, oc = item - px * 12;
        const int n = n0 + wn * 96 + oc * 8;
        const int t = (wm * TM + i) * 32 + px;
        const int ti = MI ? img + (t >> 7) : img, tl = MI ? (t & 127) : 0;
        const int gy = MI ? tl / p.Wo : y0 + t / TWX, gx = MI ? tl % p.Wo : x0 + t % TWX;
        rraw[it] = (gy < p.Ho && gx < p.Wo && n < p.Cout && ti < p.N)
                       ? *reinterpret_cast<const uint4*>(resid + ((static_cast<long>(ti) * p.Ho + gy) * p.Wo + gx) * p.Cout + n)
                       : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        st4(stage + lp * 100 + j * 32 + q * 8 + kh * 4, make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
    // (same wavefront reads what it wrote: LDS operations of a wavefront complete in order)
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int item = it * 64 + lane;            // 32 pixels x 12 octets
      const int px = item / 12, oc = item - px * 12;
      const int n = n0 + wn * 96 + oc * 8;
      const int t = (wm * TM + i) * 32 + px;
      const int ti = MI ? img + (t >> 7) : img, tl = MI ? (t & 127) : 0;
      const int gy = MI ? tl / p.Wo : y0 + t / TWX, gx = MI ? tl % p.Wo : x0 + t % TWX;
      const float4 s0 = ld4(stage + px * 100 + oc * 8), s1 = ld4(stage + px * 100 + oc * 8 + 4);
      if (gy >= p.Ho || gx >= p.Wo || n >= p.Cout || ti >= p.N) continue;
      float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      if (p.bias) {
        const float4 t0 = ld4(p.bias + n), t1 = ld4(p.bias + n + 4);
        v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
      }
      if (p.scale) {
        const float4 c0 = ld4(p.scale + n), c1 = ld4(p.scale + n + 4);
        const float4 h0 = p.shift ? ld4(p.shift + n) : make_float4(0.f, 0.f, 0.f, 0.f), h1 = p.shift ? ld4(p.shift + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[0] = v[0] * c0.x + h0.x; v[1] = v[1] * c0.y + h0.y; v[2] = v[2] * c0.z + h0.z; v[3] = v[3] * c0.w + h0.w;
        v[4] = v[4] * c1.x + h1.x; v[5] = v[5] * c1.y + h1.y; v[6] = v[6] * c1.z + h1.z; v[7] = v[7] * c1.w + h1.w;
      }
      if (rv_row) {
        const float* rv = MI ? rv_row + static_cast<long>(ti - img) * p.rowvec_ld : rv_row;
        const float4 t0 = ld4(rv + n), t1 = ld4(rv + n + 4);
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
      const long o = ((static_cast<long>(ti) * p.Ho + gy) * p.Wo + gx) * p.Cout + n;
      if (resid) {
        const f8v t = ld8(reinterpret_cast<const T*>(&rraw[it]));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += t.v[e];
      }
      f8v ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov.v[e] = v[e];
      st8(outp + o, ov);
    }
  }
}

template <int TW, int WN, typename T, bool HALF = false>
static int launch_cd(CdArgs<T>& a, hipStream_t s) {
  typedef CdShape<WN> S;
  constexpr bool MI = TW == 0;
  constexpr int TWX = MI ? 1 : TW;
  constexpr int TH = (HALF ? S::PIX / 2 : S::PIX) / TWX;
  a.tiles_x = MI ? 1 : (a.Wo + TWX - 1) / TWX;
  a.tiles_y = MI ? 1 : (a.Ho + TH - 1) / TH;
  a.tiles_n = (a.Cout + S::BN - 1) / S::BN;
  const size_t lds = S::LDS;                       // 81920: exactly two per CU
  DS_RAISE_DYNAMIC_LDS((conv16_dma_kernel<TW, WN, T, HALF>), 160 * 1024);
  const long blocks = static_cast<long>(MI ? (a.N + 1) / 2 : a.N) * a.tiles_y * a.tiles_x * a.tiles_n;
  hipLaunchKernelGGL((conv16_dma_kernel<TW, WN, T, HALF>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, s, a);
  if (MI) note_kernel("conv16_dma_kernel<0> [2 images of %dx%d pixels x %d channels, LDS-DMA halo, 2 workgroups per CU]", a.Ho, a.Wo, S::BN);
  else note_kernel("conv16_dma_kernel<%d> [%dx%d pixels x %d channels, LDS-DMA halo, 2 workgroups per CU]", TW, TH, TW, S::BN);
  return check_launch("diffsal_conv_igemm(16-bit DMA halo)");
}

// the tile of a descriptor: 0 = none, 1 = 8 x 32 x 96, 2 = 16 x 16 x 96, 3 = 8 x 24 x 192, 4 = two whole images x 96
struct CdPlan { int tile; long blocks; double used; };

static CdPlan cd_plan(const diffsal_conv_desc* d) {
  auto fill = [&](int th, int tw, int bn, int cap) {
    CdPlan r{0, 0, 0.0};
    if ((th + 2 * d->dil_h) * (tw + 2 * d->dil_w) * kCdRowB > cap - 1024) return r;     // the patch + the scratch KiB
    const long ty = (d->Ho + th - 1) / th, tx = (d->Wo + tw - 1) / tw;
    r.tile = 1;
    r.blocks = static_cast<long>(d->N) * ty * tx * ((d->Cout + bn - 1) / bn);
    r.used = static_cast<double>(d->Ho) * d->Wo / static_cast<double>(ty * th * tx * tw);
    return r;
  };
  // 256-pixel tiles: 8 x 32 where the width fills them; 16 x 16 where 32 would be mostly padding (14 x 24, 28 x 48 maps)
  const CdPlan w32 = fill(8, 32, 96, CdShape<1>::PATCH_CAP), w16 = fill(16, 16, 96, CdShape<1>::PATCH_CAP);
  CdPlan best{0, 0, 0.0};
  if (w32.tile && (!w16.tile || w32.used >= w16.used)) { best = w32; best.tile = 1; }
  else if (w16.tile) { best = w16; best.tile = 2; }
  // small maps: two whole images per 256-pixel tile, when that covers them better and both padded images fit the patch buffer
  if (d->Ho * d->Wo <= 128 && 2 * (d->Ho + 2 * d->dil_h) * (d->Wo + 2 * d->dil_w) * kCdRowB <= CdShape<1>::PATCH_CAP - 1024 &&
      static_cast<long>(d->H) * d->W * d->Cin * 2 * 2 < (1L << 31)) {
    const double used = d->Ho * d->Wo / 128.0;
    if (used > best.used + 0.05) {
      best.tile = 4;
      best.blocks = static_cast<long>((d->N + 1) / 2) * ((d->Cout + 95) / 96);
      best.used = used;
    }
  }
  // 192 pixels x 192 channels (DIFFSAL_CONV16_TILE: 0 never, 1 wherever it can run): where its 8 x 24 tiles cover the map better (28 x 48
  // at dilation 2, 14 x 24: 0.875 against 0.66 -- measured +22-24 %; on maps both cover fully the two shapes are within 3 % of each
  // other either way: these convolutions run at ~1 PF/s with the chip's clock held down, and fewer LDS reads per MFMA do not change
  // that), at most a third of the last channel tile empty, the chip still filled twice
  const int forced = tune(TUNE_CONV16_TILE);
  CdPlan sq = fill(8, 24, 192, CdShape<2>::PATCH_CAP);
  if (sq.tile && forced != 0) {
    const int waste = (d->Cout + 191) / 192 * 192 - d->Cout;
    const bool ok = forced == 1 || !best.tile || (waste * 3 <= d->Cout && sq.blocks >= 512 && sq.used >= best.used + 0.05);
    if (ok) { sq.tile = 3; return sq; }
  }
  return best;
}

// 1 if this kernel handles the descriptor (16-bit storage assumed)
int conv16_dma_applies(const diffsal_conv_desc* d, const float* bias, const float* scale, const float* shift, const float* rowvec,
                       const void* residual, const void* out) {
  // DIFFSAL_FORCE_HALO = 2 takes this kernel on every shape it can run (tests); = 1 forces conv16_halo.hip's
  const bool force = tune(TUNE_FORCE_HALO) == 2;
  if (tune(TUNE_NO_STREAM16) == 1 || tune(TUNE_NO_HALO) == 1 || tune(TUNE_FORCE_HALO) == 1 || tune(TUNE_IGEMM16_CFG) >= 0) return 0;
  const bool shape_ok = d->KH == 3 && d->KW == 3 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == d->dil_w &&
                        (d->dil_h == 1 || d->dil_h == 2) && d->pad_t == d->pad_l && d->pad_t >= 0 &&
                        d->Ho == d->H + 2 * d->pad_t - 2 * d->dil_h && d->Wo == d->W + 2 * d->pad_l - 2 * d->dil_w &&
                        d->Cin % 32 == 0 && d->Cout % 8 == 0 && d->Ho >= 4 && d->Wo >= 8 && d->Cout >= 64 &&
                        static_cast<long>(d->H) * d->W * d->Cin * 2 < (1L << 31) && static_cast<long>(d->Cout) * 9 * d->Cin * 2 < (1L << 31);
  if (!shape_ok) return 0;
  const int ld = d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout;
  if (!aligned16(out) || !aligned16(residual) || !aligned16(bias) || !aligned16(scale) || !aligned16(shift) || !aligned16(rowvec) ||
      (rowvec && ld % 4 != 0))
    return 0;
  const CdPlan pl = cd_plan(d);
  if (!pl.tile) return 0;
  // Measured against what the planner took before (tools/bench_conv16.py, 4 .. 64 clips): ahead wherever the tiles are not mostly
  // padding (the 9 x 14 extended grid of a 7 x 12 map fills 0.49 of a tile: the generic kernel's flattened rows win) and the
  // launch is not a handful of workgroups with a long K walk (14 x 24 maps of the noise encoder at 4 clips, K = 6912: the generic
  // kernel splits K over the idle CUs)
  return force || (pl.used >= 0.6 && pl.blocks >= 96 && (pl.blocks >= 512 || 9 * d->Cin <= 3456));
}

template <typename T>
static int run_cd(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale, const float* shift,
                  const float* rowvec, const void* residual, void* out, hipStream_t s) {
  CdArgs<T> a;
  a.in = static_cast<const T*>(in); a.w = static_cast<const T*>(w); a.bias = bias; a.scale = scale; a.shift = shift;
  a.rowvec = rowvec; a.residual = static_cast<const T*>(residual); a.out = static_cast<T*>(out);
  a.N = d->N; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.Cin = d->Cin; a.Cout = d->Cout; a.K = 9 * d->Cin;
  a.dil = d->dil_h; a.pad = d->pad_t; a.act = d->act;
  a.rowvec_ld = d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout;
  const CdPlan pl = cd_plan(d);
  // fewer workgroups than CUs on the 256-pixel tiles: 128-pixel tiles (DIFFSAL_CONV16_HALF = 0 / 1: never / on every 8 x 32, 16 x 16 shape)
  const int fh = tune(TUNE_CONV16_HALF);
  const bool half = fh == 1 || (fh != 0 && pl.blocks < 224);
  switch (pl.tile) {
    case 1: return half ? launch_cd<32, 1, T, true>(a, s) : launch_cd<32, 1, T>(a, s);
    case 2: return half ? launch_cd<16, 1, T, true>(a, s) : launch_cd<16, 1, T>(a, s);
    case 4: return launch_cd<0, 1, T>(a, s);
    default: return launch_cd<24, 2, T>(a, s);
  }
}

int conv16_dma_launch(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                      const float* shift, const float* rowvec, const void* residual, void* out, hipStream_t s) {
  if (d->dtype == DIFFSAL_BF16) return run_cd<__bf16>(d, in, w, bias, scale, shift, rowvec, residual, out, s);
  return run_cd<_Float16>(d, in, w, bias, scale, shift, rowvec, residual, out, s);
}

}  // namespace diffsal
