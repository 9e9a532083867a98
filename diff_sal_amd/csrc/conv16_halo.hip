// 3x3 stride-1 "same" convolution (dilation 1 or 2) on bf16 / fp16 storage with an LDS halo patch.
//
// The generic 16-bit implicit-GEMM kernel (igemm16.hip) fetches the A operand once per tap: nine 64-byte runs per pixel
// and 32-channel chunk travel L2 -> L1 -> registers -> LDS, and that delivery path (~27 B/clk/CU), not the matrix pipe,
// sets its speed (36 % of the 16-bit MFMA peak at best for a 128 x 96 tile).  Here a workgroup owns a TH x TW block of
// output pixels and stages, per 32-channel chunk, the (TH + 2d) x (TW + 2d) input patch ONCE; all nine taps read their
// A fragments from it with a constant LDS displacement (tap offset), so the A bytes per MFMA drop ~6x and a 512-pixel
// block amortises the nine weight slices (9 x Cout_tile x 64 B per chunk) over twice as many pixels as before.
//
//   tile A: 16 x 32 pixels, 8 wavefronts (each 2 image rows of 32 pixels), patch up to 20 x 36 pixels
//   tile B: 16 x 16 pixels, 4 wavefronts (each 4 image rows of 16 pixels), patch up to 20 x 20 pixels
//   N tile: 96 or 128 output channels (TN = 3 / 4 MFMA column tiles)
//
// One LDS stage (patch + nine weight slices, 80-byte pixel pitch: conflict-free ds_read_b128 for 16 consecutive pixels);
// the next chunk's loads are issued into registers before the 108 / 144 MFMAs of the current chunk and parked behind a
// barrier afterwards.  Same packed weight layout [Cout][Cin/32][9][32], same fused epilogue (bias, BN affine, per-image
// vector, activation, residual) and the same fp32 accumulation as the other implicit-GEMM kernels.  Selected by
// diffsal_conv_igemm for 16-bit storage when the shape qualifies (UpEmbed, ResnetBlock and mt_proj convolutions:
// R/models/saliency_decoder/common_block.py:196-216, sal_unet.py:104-112,407).
#include <cstdlib>

#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct HaloMma;
template <> struct HaloMma<__bf16> {
  typedef hbf16x8 vec;
  static __device__ __forceinline__ f32x16 run(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct HaloMma<_Float16> {
  typedef hf16x8 vec;
  static __device__ __forceinline__ f32x16 run(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T>
struct HaloArgs {
  const T* in;
  const T* w;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const T* residual;
  T* out;
  int N, H, W, Cin, Cout, K;   // K = 9 * Cin
  int dil, act, rowvec_ld;
  int tiles_x, tiles_y, tiles_n;
#ifdef DIFFSAL_DEV_STAMPS
  unsigned long long* stamps;   // development build only: per-workgroup phase time stamps (diffsal_set_halo_stamps)
#endif
};

#ifdef DIFFSAL_DEV_STAMPS
static unsigned long long* g_halo_stamps = nullptr;
static size_t g_halo_stamp_bytes = 0;
#endif

constexpr int HP = 40;   // elements per staged pixel / weight row: 32 data + 8 pad (80 bytes)

// Pipeline.  A "step" is one kernel row ky of one 32-channel chunk: 3 taps x 2 k-steps x TM x TN MFMAs per wavefront.
// LDS holds the input patch of TWO chunks (double buffer) and a two-slot ring of weight rows (3 taps x BN x 32
// channels each).  At step g a thread first parks what it fetched during step g-1 -- the weight row of step g+1 and one
// third of the NEXT chunk's patch -- into the buffers nobody reads during step g, then issues the fetches step g+1 will
// park, then runs the step's MFMAs; one barrier per step.  No phase without matrix work, 20 staging VGPRs.
template <int TW, int NW, int TN, typename T>
__global__ __launch_bounds__(NW * 64) void conv16_halo_kernel(HaloArgs<T> p) {
  typedef typename HaloMma<T>::vec vec;
  constexpr int TH = 16;
  constexpr int NT = NW * 64;
  constexpr int RPM = 32 / TW;                     // image rows per 32-pixel MFMA tile
  constexpr int TM = (TH * TW) / (NW * 32);        // MFMA row tiles per wavefront
  static_assert(TM == 2, "each wavefront owns 64 pixels");
  constexpr int BN = TN * 32;
  constexpr int B_ELEMS = 3 * BN * HP;             // one ring slot: [3 taps][BN][HP]
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef DIFFSAL_DEV_STAMPS
  auto stamp = [&](int k) { if (p.stamps && tid == 0) p.stamps[blockIdx.x * 8 + k] = wall_clock64(); };
#else
  auto stamp = [](int) {};
#endif
  stamp(0);
  // block -> (image, tile_y, tile_x, n tile); consecutive blocks share the patch's neighbourhood and the same weights
  int b;
  {  // XCD-aware order (see igemm.hip): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run of
     // tiles -- neighbouring patches share their halo rows and the N tiles of a patch share all of it, in ONE L2
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
  const int a_elems = PH * PW * HP;
  T* As = reinterpret_cast<T*>(smraw);             // [2][PH][PW][HP]
  T* Bs = As + 2 * a_elems;                        // [2][3][BN][HP]
  const int n_chunks = p.Cin >> 5;
  const int G = 3 * n_chunks;

  // ---- loader bookkeeping: 16-byte pieces; patch piece = (pixel, quarter), weight piece = (co, kx, quarter).
  // Branch-free: buffer loads return zero for out-of-range offsets, so padding / tails carry offset 0x80000000.
  constexpr int A_MAX = ((TH + 4) * (TW + 4) * 4 + NT - 1) / NT;
  constexpr int A_STEP = (A_MAX + 2) / 3;          // patch pieces a thread moves per step
  constexpr int B_PER = (3 * BN * 4 + NT - 1) / NT;
  constexpr unsigned DEAD = 0x80000000u;
  const T* in_img = p.in + static_cast<long>(img) * p.H * p.W * p.Cin;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(in_img), 0, p.H * p.W * p.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(p.w), 0, p.Cout * p.K * 2, 0x00020000);
  unsigned a_voff[3 * A_STEP], b_voff[B_PER];
  int b_dst[B_PER];
  const int a_pieces = PH * PW * 4;
#pragma unroll
  for (int i = 0; i < 3 * A_STEP; ++i) {
    const int idx = tid + i * NT;
    const int q = idx & 3, pix = idx >> 2;
    const int pr = pix / PW, pc = pix - pr * PW;
    const int gy = y0 - d + pr, gx = x0 - d + pc;
    const bool ok = idx < a_pieces && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    a_voff[i] = ok ? static_cast<unsigned>(((gy * p.W + gx) * p.Cin + q * 8) * 2) : DEAD;
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int idx = tid + i * NT;
    const int co = idx / 12, r = idx - co * 12;            // 12 pieces = 3 taps x 4 quarters: 192 contiguous bytes
    const bool ok = idx < 3 * BN * 4 && n0 + co < p.Cout;
    b_voff[i] = ok ? static_cast<unsigned>(((n0 + co) * p.K + r * 8) * 2) : DEAD;
    b_dst[i] = idx < 3 * BN * 4 ? ((r >> 2) * BN + co) * HP + (r & 3) * 8 : -1;
  }
  uint4 ra[A_STEP], rb[B_PER];
  // fetches that step gs will park: weight row of step gs+1; third (gs % 3) of the patch of chunk gs/3 + 1
  auto issue = [&](int gs) {
    const int c1 = gs / 3 + 1, part = gs - (c1 - 1) * 3;
    const unsigned a_dead = c1 < n_chunks ? 0u : DEAD;
    const unsigned b_dead = gs + 1 < G ? 0u : DEAD;
#pragma unroll
    for (int i = 0; i < A_STEP; ++i) {
      const unsigned vo = part == 0 ? a_voff[i] : (part == 1 ? a_voff[A_STEP + i] : a_voff[2 * A_STEP + i]);
      ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (vo + c1 * 64) | a_dead, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (b_voff[i] + (gs + 1) * 192) | b_dead, 0, 0));
  };
  auto park = [&](int gs) {
    const int c1 = gs / 3 + 1, part = gs - (c1 - 1) * 3;
    if (c1 < n_chunks) {
      T* dst = As + (c1 & 1) * a_elems;
#pragma unroll
      for (int i = 0; i < A_STEP; ++i) {
        const int idx = tid + (part * A_STEP + i) * NT;
        if (idx < a_pieces) *reinterpret_cast<uint4*>(dst + (idx >> 2) * HP + (idx & 3) * 8) = ra[i];
      }
    }
    if (gs + 1 < G) {
      T* dst = Bs + ((gs + 1) & 1) * B_ELEMS;
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        if (b_dst[i] >= 0) *reinterpret_cast<uint4*>(dst + b_dst[i]) = rb[i];
    }
  };

  // ---- fragment addressing: lane -> pixel (py, px) of its MFMA row tile; k half = lane >> 5
  const int lp = lane & 31, kh = lane >> 5;
  int a_base[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int py = (wave * TM + i) * RPM + lp / TW, px = lp % TW;
    a_base[i] = (py * PW + px) * HP + kh * 8;
  }
  const int b_base = lp * HP + kh * 8;
  stamp(1);

  {  // prologue: chunk 0's whole patch and weight row 0 land together, then the first regular fetch is put in flight
    uint4 pa[3 * A_STEP];
#pragma unroll
    for (int i = 0; i < 3 * A_STEP; ++i)
      pa[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_voff[i], 0, 0));
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_voff[i], 0, 0));
#pragma unroll
    for (int i = 0; i < 3 * A_STEP; ++i) {
      const int idx = tid + i * NT;
      if (idx < a_pieces) *reinterpret_cast<uint4*>(As + (idx >> 2) * HP + (idx & 3) * 8) = pa[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      if (b_dst[i] >= 0) *reinterpret_cast<uint4*>(Bs + b_dst[i]) = rb[i];
    issue(0);
    __syncthreads();
  }
  stamp(2);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const T* Ab = As + (chunk & 1) * a_elems;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int g = chunk * 3 + ky;
      park(g);                                           // what step g-1 fetched; nobody reads those buffers now
      issue(g + 1);                                      // lands during this step's MFMAs
      const T* Bb = Bs + (g & 1) * B_ELEMS;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int toff = (ky * d * PW + kx * d) * HP;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          vec fa[TM], fb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const vec*>(Ab + a_base[i] + toff + kk * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const vec*>(Bb + (kx * BN + j * 32) * HP + b_base + kk * 16);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = HaloMma<T>::run(fb[j], fa[i], acc[i][j]);   // D^T: rows = channels, cols = pixels
        }
      }
      __syncthreads();
    }
  }
  stamp(3);

  // ---- epilogue.  The weights went in as the MFMA "A" operand, so a lane holds, for pixel lane & 31 of its row tile,
  // output channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5): four consecutive channels per register quad, stored as one
  // 8-byte piece with the per-channel affine, per-image vector, activation and residual applied on the way (one rounding,
  // like igemm16.hip; no LDS staging, no barrier).
  const T* __restrict__ resid = p.residual;
  T* __restrict__ outp = p.out;
  const int hq = (lane >> 5) * 4;
  const float* rv_row = p.rowvec ? p.rowvec + static_cast<long>(img) * p.rowvec_ld : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mt = wave * TM + i;
    const int gy = y0 + mt * RPM + lp / TW, gx = x0 + lp % TW;
    if (gy >= p.H || gx >= p.W) continue;
    const long obase = ((static_cast<long>(img) * p.H + gy) * p.W + gx) * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + j * 32 + g * 8 + hq;
        if (n >= p.Cout) continue;
        const long o = obase + n;
        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
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
  stamp(4);
  stamp(5);
}

template <int TW, int NW, int TN, typename T>
static int launch_halo(HaloArgs<T>& a, hipStream_t s) {
  constexpr int TH = 16;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + TH - 1) / TH;
  a.tiles_n = (a.Cout + TN * 32 - 1) / (TN * 32);
  const size_t lds = (2 * static_cast<size_t>(TH + 2 * a.dil) * (TW + 2 * a.dil) + 2 * 3 * TN * 32) * HP * sizeof(T);
  DS_RAISE_DYNAMIC_LDS((conv16_halo_kernel<TW, NW, TN, T>), 160 * 1024);
  const long blocks = static_cast<long>(a.N) * a.tiles_y * a.tiles_x * a.tiles_n;
  hipLaunchKernelGGL((conv16_halo_kernel<TW, NW, TN, T>), dim3(static_cast<unsigned>(blocks)), dim3(NW * 64), lds, s, a);
  note_kernel("conv16_halo_kernel<%d, %d, %d> [16x%d pixels x %d channels]", TW, NW, TN, TW, TN * 32);
  return check_launch("diffsal_conv_igemm(16-bit halo)");
}

// tile choice shared by the eligibility test and the launcher
static void halo_tiles(const diffsal_conv_desc* d, bool* wide, bool* n128) {
  *n128 = (d->Cout % 96 != 0) && (d->Cout % 128 == 0);   // 96-wide N tiles unless only 128 divides Cout
  *wide = (d->W % 32 == 0) || d->W >= 64;                // 16 x 32 pixel tiles when the width fills them
}

template <typename T>
static int run_halo(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                    const float* shift, const float* rowvec, const void* residual, void* out, hipStream_t s) {
  HaloArgs<T> a;
  a.in = static_cast<const T*>(in); a.w = static_cast<const T*>(w); a.bias = bias; a.scale = scale; a.shift = shift;
  a.rowvec = rowvec; a.residual = static_cast<const T*>(residual); a.out = static_cast<T*>(out);
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.K = 9 * d->Cin; a.dil = d->dil_h; a.act = d->act;
  a.rowvec_ld = d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout;
#ifdef DIFFSAL_DEV_STAMPS
  a.stamps = g_halo_stamps;
  if (a.stamps) {   // 8 stamps per workgroup: refuse a buffer the launch would overrun
    bool wide_, n128_;
    halo_tiles(d, &wide_, &n128_);
    const long tw_ = wide_ ? 32 : 16, tn_ = (n128_ ? 4 : 3) * 32;
    const long blocks = static_cast<long>(d->N) * ((d->H + 15) / 16) * ((d->W + tw_ - 1) / tw_) * ((d->Cout + tn_ - 1) / tn_);
    if (static_cast<size_t>(blocks) * 8 * sizeof(unsigned long long) > g_halo_stamp_bytes) a.stamps = nullptr;
  }
#endif
  bool wide, n128;
  halo_tiles(d, &wide, &n128);
  if (wide) return n128 ? launch_halo<32, 8, 4, T>(a, s) : launch_halo<32, 8, 3, T>(a, s);
  return n128 ? launch_halo<16, 4, 4, T>(a, s) : launch_halo<16, 4, 3, T>(a, s);
}

// 1 if the halo kernel handles this descriptor (16-bit storage assumed), else 0
int conv16_halo_applies(const diffsal_conv_desc* d) {
  bool force = false;
  if (tune(TUNE_NO_HALO) == 1) return 0;
  force = tune(TUNE_FORCE_HALO) == 1;
  const bool shape_ok = d->KH == 3 && d->KW == 3 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == d->dil_w &&
                        (d->dil_h == 1 || d->dil_h == 2) && d->pad_t == d->dil_h && d->pad_l == d->dil_w && d->Ho == d->H &&
                        d->Wo == d->W && d->Cin % 32 == 0 && d->Cout % 4 == 0 && d->H >= 8 && d->W >= 16 && d->Cout >= 64;
  if (!shape_ok) return 0;
  bool wide, n128;
  halo_tiles(d, &wide, &n128);
  const size_t lds = (2 * static_cast<size_t>(16 + 2 * d->dil_h) * ((wide ? 32 : 16) + 2 * d->dil_h) + 6 * (n128 ? 128 : 96)) * HP * 2;
  if (lds > 160 * 1024) return 0;
  return force || static_cast<long>(d->N) * d->H * d->W >= 80000;   // below that the generic tiles fill the chip better
}

// the coalesced epilogue moves 8-byte output / residual pieces and 16-byte parameter pieces
int conv16_halo_pointers_ok(const diffsal_conv_desc* d, const float* bias, const float* scale, const float* shift,
                            const float* rowvec, const void* residual, const void* out) {
  auto al = [](const void* q, uintptr_t m) { return (reinterpret_cast<uintptr_t>(q) & m) == 0; };
  const int ld = d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout;
  return al(out, 7) && al(residual, 7) && al(bias, 15) && al(scale, 15) && al(shift, 15) && al(rowvec, 15) && (!rowvec || ld % 4 == 0);
}

int conv16_halo_launch(const diffsal_conv_desc* d, const void* in, const void* w, const float* bias, const float* scale,
                       const float* shift, const float* rowvec, const void* residual, void* out, hipStream_t s) {
  if (d->dtype == DIFFSAL_BF16) return run_halo<__bf16>(d, in, w, bias, scale, shift, rowvec, residual, out, s);
  return run_halo<_Float16>(d, in, w, bias, scale, shift, rowvec, residual, out, s);
}

}  // namespace diffsal

#ifdef DIFFSAL_DEV_STAMPS
// Development builds only (hipcc -DDIFFSAL_DEV_STAMPS; tools/probe_halo_stamps.py): a caller-owned device buffer receives 8 time
// stamps per workgroup of every following halo launch that fits in `bytes`; (nullptr, 0) turns it off.  Not in the shipped ABI.
extern "C" int diffsal_set_halo_stamps(void* device_buffer, size_t bytes) {
  diffsal::g_halo_stamps = static_cast<unsigned long long*>(device_buffer);
  diffsal::g_halo_stamp_bytes = device_buffer ? bytes : 0;
  return DIFFSAL_OK;
}
#endif
