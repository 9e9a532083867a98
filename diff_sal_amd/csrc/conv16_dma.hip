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
constexpr int kCdPatchIss = 7;                  // patch DMA instructions per wavefront and chunk (28 KiB >= the largest patch)
constexpr int kCdSlotRows = 96;                 // weight rows of a ring slot (the N tile)

// TW: tile width in pixels (32: 8 x 32 tile, a wavefront owns two image rows; 16: 16 x 16 tile, four image rows)
template <int TW, typename T>
__global__ __launch_bounds__(256, 2) void conv16_dma_kernel(CdArgs<T> p) {
  typedef typename CdMma<T>::vec vec;
  constexpr int TH = 256 / TW;
  constexpr int RPM = 32 / TW;                     // image rows per 32-pixel MFMA tile
  constexpr int TM = 2, TN = 3, BN = 96;
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
  const int img = b / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const int d = p.dil;
  const int PH = TH + 2 * d, PW = TW + 2 * d;
  const int patch_bytes = PH * PW * kCdRowB;       // <= 27648
  // LDS: [patch 0][patch 1][weight slot 0..2][scratch KiB]
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((cd_lds_ptr_t)cd_smem));
  constexpr int PATCH_CAP = kCdPatchIss * kCdWaves * 1024;           // 28672
  constexpr int SLOT_B = kCdSlotRows * kCdRowB;                      // 8192
  // dead DMA instructions write zeros into the last KiB of patch buffer 0's 28 KiB, which no patch reaches (host-checked)
  const unsigned lds_w = lds0 + 2 * PATCH_CAP, lds_scratch = lds0 + PATCH_CAP - 1024;
  const int n_chunks = p.Cin >> 5;
  const int G = 9 * n_chunks;

  // ---- issue side.  Patch instruction q of a wavefront (q < 7) covers pieces idx = (q * 4 + wave) * 64 + lane: pixel idx >> 2,
  // physical slot idx & 3, which receives logical slot (idx & 3) ^ ((pixel >> 2) & 3).  Weight instruction q (q < 2): rows
  // (q * 4 + wave) * 16 + (lane >> 2).
  const T* in_img = p.in + static_cast<long>(img) * p.H * p.W * p.Cin;
  const unsigned long pa = reinterpret_cast<unsigned long>(in_img);
  const cd_i32x4 rs_a = cd_i32x4{static_cast<int>(pa), static_cast<int>(pa >> 32) & 0xFFFF, p.H * p.W * p.Cin * 2, 0x00020000};
  const unsigned long pw = reinterpret_cast<unsigned long>(p.w);
  const cd_i32x4 rs_w = cd_i32x4{static_cast<int>(pw), static_cast<int>(pw >> 32) & 0xFFFF, p.Cout * p.K * 2, 0x00020000};
  unsigned a_voff[kCdPatchIss], w_voff[2];
#pragma unroll
  for (int q = 0; q < kCdPatchIss; ++q) {
    const int idx = (q * kCdWaves + wave) * 64 + lane;
    const int pix = idx >> 2, ls = (idx & 3) ^ ((pix >> 2) & 3);
    const int pr = pix / PW, pc = pix - pr * PW;
    const int gy = y0 - p.pad + pr, gx = x0 - p.pad + pc;
    const bool ok = pix < PH * PW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_voff[q] = ok ? static_cast<unsigned>(((gy * p.W + gx) * p.Cin + ls * 8) * 2) : DEAD;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (q * kCdWaves + wave) * 16 + (lane >> 2), ls = (lane & 3) ^ ((row >> 2) & 3);
    const bool ok = row < BN && n0 + row < p.Cout;
    w_voff[q] = ok ? static_cast<unsigned>(((n0 + row) * p.K + ls * 8) * 2) : DEAD;
  }
  const bool w_live1 = (kCdWaves + wave) * 16 < BN;          // the second weight instruction of waves 2 and 3 covers rows >= 96: scratch
  // patch instruction q of chunk c -> buffer c & 1 (instructions past the patch write zeros into their own KiB of the 28 KiB buffer)
  auto issue_patch = [&](int c, int q) __attribute__((always_inline)) {
    const unsigned dst = lds0 + (c & 1) * PATCH_CAP + (q * kCdWaves + wave) * 1024;
    const bool live = c < n_chunks && (q * kCdWaves + wave) * 1024 < patch_bytes + 1024;
    cd_dma(live ? dst : lds_scratch, live ? a_voff[q] : DEAD, rs_a, static_cast<unsigned>(c) * 64u);
  };
  auto issue_weights = [&](int g) __attribute__((always_inline)) {       // the slice of step g (chunk g / 9, tap g % 9) -> slot g % 4
    const bool live = g < G;
    const unsigned dst = lds_w + (g & 3) * SLOT_B;
    const unsigned soff = static_cast<unsigned>(g) * 64u;      // [Cin / 32][9][32] inside a weight row: step g is 64 bytes further
    cd_dma(live ? dst + wave * 1024 : lds_scratch, live ? w_voff[0] : DEAD, rs_w, soff);
    cd_dma(live && w_live1 ? dst + (kCdWaves + wave) * 1024 : lds_scratch, live && w_live1 ? w_voff[1] : DEAD, rs_w, soff);
  };

  // ---- fragment addressing: lane -> pixel lp of its MFMA row tile, k half kh; logical slot of (kk, kh) = 2 kk + kh
  const int lp = lane & 31, kh = lane >> 5;
  int a_pix[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_pix[i] = ((wave * TM + i) * RPM + lp / TW) * PW + lp % TW;
  int b_off[TN][2];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = j * 32 + lp;
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
      if (g == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(10)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue_weights(g + 2);
      if (tap >= 1 && tap <= kCdPatchIss) issue_patch(chunk + 1, tap - 1);
      else cd_dma(lds_scratch, DEAD, rs_a, 0u);
      const unsigned char* Bb = cd_smem + 2 * PATCH_CAP + (g & 3) * SLOT_B;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int toff = ky * d * PW + kx * d;
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
        const int n = n0 + oc * 8;
        const int gy = y0 + (wave * TM + i) * RPM + px / TW, gx = x0 + px % TW;
        rraw[it] = (gy < p.Ho && gx < p.Wo && n < p.Cout)
                       ? *reinterpret_cast<const uint4*>(resid + ((static_cast<long>(img) * p.Ho + gy) * p.Wo + gx) * p.Cout + n)
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
      const int n = n0 + oc * 8;
      const int mt = wave * TM + i;
      const int gy = y0 + mt * RPM + px / TW, gx = x0 + px % TW;
      const float4 s0 = ld4(stage + px * 100 + oc * 8), s1 = ld4(stage + px * 100 + oc * 8 + 4);
      if (gy >= p.Ho || gx >= p.Wo || n >= p.Cout) continue;
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
        const float4 t0 = ld4(rv_row + n), t1 = ld4(rv_row + n + 4);
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
      const long o = ((static_cast<long>(img) * p.Ho + gy) * p.Wo + gx) * p.Cout + n;
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

template <int TW, typename T>
static int launch_cd(CdArgs<T>& a, hipStream_t s) {
  constexpr int TH = 256 / TW;
  a.tiles_x = (a.Wo + TW - 1) / TW;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  a.tiles_n = (a.Cout + 95) / 96;
  const size_t lds = 2 * kCdPatchIss * kCdWaves * 1024 + 4 * kCdSlotRows * kCdRowB;       // 81920: exactly two per CU
  DS_RAISE_DYNAMIC_LDS((conv16_dma_kernel<TW, T>), 160 * 1024);
  const long blocks = static_cast<long>(a.N) * a.tiles_y * a.tiles_x * a.tiles_n;
  hipLaunchKernelGGL((conv16_dma_kernel<TW, T>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, s, a);
  note_kernel("conv16_dma_kernel<%d> [%dx%d pixels x 96 channels, LDS-DMA halo, 2 workgroups per CU]", TW, TH, TW);
  return check_launch("diffsal_conv_igemm(16-bit DMA halo)");
}

static bool cd_wide(const diffsal_conv_desc* d) {
  // 8 x 32 tiles where the width fills them; 16 x 16 where 32 would be mostly padding (14 x 24, 28 x 48 maps)
  const int w32 = (d->Wo + 31) / 32 * 32, w16 = (d->Wo + 15) / 16 * 16;
  const int h8 = (d->Ho + 7) / 8 * 8, h16 = (d->Ho + 15) / 16 * 16;
  return static_cast<long>(w32) * h8 <= static_cast<long>(w16) * h16;
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
  const bool wide = cd_wide(d);
  const int th = wide ? 8 : 16, tw = wide ? 32 : 16;
  if ((th + 2 * d->dil_h) * (tw + 2 * d->dil_w) * kCdRowB > kCdPatchIss * kCdWaves * 1024 - 1024) return 0;    // + the scratch KiB
  // Measured against what the planner took before (tools/bench_conv16.py, 4 .. 64 clips): ahead wherever the tiles are not mostly
  // padding (the 9 x 14 extended grid of a 7 x 12 map fills 0.49 of a tile: the generic kernel's flattened rows win) and the
  // launch is not a handful of workgroups with a long K walk (14 x 24 maps of the noise encoder at 4 clips, K = 6912: the generic
  // kernel splits K over the idle CUs)
  const long ty = (d->Ho + th - 1) / th, tx = (d->Wo + tw - 1) / tw;
  const long blocks = static_cast<long>(d->N) * ty * tx * ((d->Cout + 95) / 96);
  const double used = static_cast<double>(d->Ho) * d->Wo / static_cast<double>(ty * th * tx * tw);
  return force || (used >= 0.6 && blocks >= 96 && (blocks >= 512 || 9 * d->Cin <= 3456));
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
  return cd_wide(d) ? launch_cd<32, T>(a, s) : launch_cd<16, T>(a, s);
}

int conv16_dma_launch(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                      const float* shift, const float* rowvec, const void* residual, void* out, hipStream_t s) {
  if (d->dtype == DIFFSAL_BF16) return run_cd<__bf16>(d, in, w, bias, scale, shift, rowvec, residual, out, s);
  return run_cd<_Float16>(d, in, w, bias, scale, shift, rowvec, residual, out, s);
}

}  // namespace diffsal
