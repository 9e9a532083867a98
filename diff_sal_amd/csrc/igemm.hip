// Implicit-GEMM convolution / linear layer on the gfx950 fp32 matrix cores.
//
//   out[m, co] = epilogue( sum_k A[m, k] * Wt[co, k] )
//
// A is the im2col view of a channels-last image (never materialised): m = (n, oy, ox),
// k = (ky, kx, ci) with ci fastest, so a 32-wide K slice is one contiguous 128-byte run of
// one input pixel.  Wt is the weight packed as [Cout][K]: both operands are K-contiguous
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
#include "common.h"

namespace diffsal {

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
  int KW, stride_h, stride_w, pad_t, pad_l, dil_h, dil_w;
  int act, rowvec_ld;
  int n_tiles_n, n_tiles;  // tiles along N, total tiles
};

constexpr int BK = 32;
constexpr int PITCH = BK + 4;  // dwords; 36*r mod 64 hits 16 distinct 4-bank slots for 16 rows

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmArgs p) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int A_PASSES = BM / 32;
  constexpr int B_PASSES = BN / 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");

  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * PITCH];
  float* As = smem;
  float* Bs = smem + BM * PITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous run of tiles (same-M tiles share their A rows in one L2).  Bijective for any count.
  int tile;
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
  const int lrow = tid >> 3;
  const int lcol = (tid & 7) * 4;

  int a_iy0[A_PASSES], a_ix0[A_PASSES];
  long a_base[A_PASSES];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < A_PASSES; ++j) {
    const int m = m0 + lrow + 32 * j;
    if (m < p.M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      a_iy0[j] = oy * p.stride_h - p.pad_t;
      a_ix0[j] = ox * p.stride_w - p.pad_l;
      a_base[j] = static_cast<long>(n) * p.H * p.W * p.Cin + lcol;
    } else {
      a_iy0[j] = -(1 << 28);  // forces the bounds test to fail for every tap
      a_ix0[j] = 0;
      a_base[j] = 0;
    }
  }
  long b_off[B_PASSES];
  bool b_ok[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int n = n0 + lrow + 32 * j;
    b_ok[j] = n < p.Cout;
    b_off[j] = static_cast<long>(b_ok[j] ? n : 0) * p.K + lcol;
  }

  float4 ra[A_PASSES], rb[B_PASSES];
  const int KT = p.K / BK;

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    const int tap = k0 / p.Cin;  // wave-uniform
    const int ci0 = k0 - tap * p.Cin;
    const int ky = tap / p.KW;
    const int kx = tap - ky * p.KW;
    const int dy = ky * p.dil_h, dx = kx * p.dil_w;
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) {
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
      const long off = a_base[j] + (static_cast<long>(iy) * p.W + ix) * p.Cin + ci0;
      ra[j] = ok ? ld4(p.in + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      rb[j] = b_ok[j] ? ld4(p.w + b_off[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) st4(&As[(lrow + 32 * j) * PITCH + lcol], ra[j]);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) st4(&Bs[(lrow + 32 * j) * PITCH + lcol], rb[j]);
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
  const float* a_frag = As + (wm * TM * 32 + frow) * PITCH + fk;
  const float* b_frag = Bs + (wn * TN * 32 + frow) * PITCH + fk;

  load_tile(0);
  for (int kt = 0; kt < KT; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < KT) load_tile(kt + 1);  // global loads fly under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = ld4(a_frag + i * 32 * PITCH + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = ld4(b_frag + j * 32 * PITCH + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float av = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float bv = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + col_l;
    if (n >= p.Cout) continue;
    const float bi = p.bias ? p.bias[n] : 0.f;
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + (wm * TM + i) * 32 + row_h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        v += bi;
        if (p.scale) v = v * sc + sh;
        if (p.rowvec) v += p.rowvec[static_cast<long>(m / HoWo) * p.rowvec_ld + n];
        if (p.act == DIFFSAL_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == DIFFSAL_ACT_GELU_ERF) v = gelu_erf(v);
        else if (p.act == DIFFSAL_ACT_SIGMOID) v = sigmoidf_(v);
        const long o = static_cast<long>(m) * p.Cout + n;
        if (p.residual) v += p.residual[o];
        p.out[o] = v;
      }
    }
  }
}

template <int WM, int WN, int TM, int TN>
static int launch(IgemmArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  a.n_tiles_n = (a.Cout + BN - 1) / BN;
  const int tiles_m = (a.M + BM - 1) / BM;
  a.n_tiles = a.n_tiles_n * tiles_m;
  hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN>), dim3(a.n_tiles), dim3(256), 0, s, a);
  return check_launch("diffsal_conv_igemm");
}

}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_conv_igemm(const diffsal_conv_desc* d, const float* in, const float* w,
                                  const float* bias, const float* scale, const float* shift,
                                  const float* rowvec, const float* residual, float* out,
                                  diffsal_stream_t stream) {
  DS_REQUIRE(d && in && w && out, DIFFSAL_E_ARG, "conv_igemm: null argument");
  DS_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0, DIFFSAL_E_SHAPE, "conv_igemm: Cin=%d must be a multiple of 32", d->Cin);
  DS_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride_h > 0 &&
                 d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0,
             DIFFSAL_E_SHAPE, "conv_igemm: non-positive dimension");
  DS_REQUIRE(d->Ho > 0 && d->Wo > 0, DIFFSAL_E_SHAPE, "conv_igemm: empty output %dx%d", d->Ho, d->Wo);
  DS_REQUIRE((scale == nullptr) == (shift == nullptr), DIFFSAL_E_ARG, "conv_igemm: scale and shift go together");
  DS_REQUIRE(aligned16(in) && aligned16(w), DIFFSAL_E_ALIGN, "conv_igemm: in/w must be 16-byte aligned");
  const long M = static_cast<long>(d->N) * d->Ho * d->Wo;
  DS_REQUIRE(M < (1L << 31) && M * d->Cout < (1L << 40), DIFFSAL_E_SHAPE, "conv_igemm: problem too large");

  IgemmArgs a;
  a.in = in; a.w = w; a.bias = bias; a.scale = scale; a.shift = shift; a.rowvec = rowvec;
  a.residual = residual; a.out = out;
  a.M = static_cast<int>(M);
  a.K = d->KH * d->KW * d->Cin;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KW = d->KW; a.stride_h = d->stride_h; a.stride_w = d->stride_w; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.dil_h = d->dil_h; a.dil_w = d->dil_w; a.act = d->act;
  a.rowvec_ld = d->rowvec_ld > 0 ? d->rowvec_ld : d->Cout;
  hipStream_t s = static_cast<hipStream_t>(stream);

  // Tile choice: the widest tile whose grid still fills the 256 CUs a few times over.
  const int Cout = d->Cout;
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
  const bool n96 = (Cout % 96 == 0) && (Cout % 128 != 0);
  if (n96 && Cout % 192 == 0 && blocks(128, 192) >= 768) return launch<2, 2, 2, 3>(a, s);
  if (n96 && blocks(128, 96) >= 192) return launch<4, 1, 1, 3>(a, s);
  if (!n96 && blocks(128, 128) >= 512) return launch<2, 2, 2, 2>(a, s);
  if (blocks(64, 128) >= 384 && Cout % 128 == 0) return launch<2, 2, 1, 2>(a, s);
  return launch<2, 2, 1, 1>(a, s);
}
