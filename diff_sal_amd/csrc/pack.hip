// Weight layout transforms between the parameter layout of the reference ([Cout][Cin][KH*KW], what state_dict holds)
// and the k order of the implicit-GEMM kernels (include/diffsal.h: k = (ci/32, tap, ci%32)).  Run once per training
// step per layer (the optimizer rewrites the parameters), so they are written as tiled transposes instead of being
// left to generic strided copies:
//   mode 0  pack      out[co][(ci/32, tap, ci%32)]               = w[co][ci][tap]
//   mode 1  dgrad     out[ci][(co/32, tap, co%32)]               = w[co][ci][taps-1-tap]   (flipped, transposed:
//                     the weight of the data-gradient convolution dX = conv(dY, .), see autograd_ops.ConvFn)
//   mode 2  unpack    dw[co][ci][tap]                            = dw_packed[co][(ci/32, tap, ci%32)]
//   mode 3  columns   out[(tap, ci)][co]                         = w[co][ci][tap]          (weight of the GEMM
//                     dXcols = dY W for convolutions whose taps never overlap (stride >= kernel): its output is
//                     scattered by col2im_disjoint below instead of running a zero-inserted convolution)
// One workgroup moves a 32(co) x 32(ci) x taps block through LDS; both sides are read/written as contiguous
// 32*taps-float rows with float4 accesses.
#include "common.h"

namespace diffsal {
namespace {

__device__ __forceinline__ void pack_tile_body(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int taps,
                                               int mode, int bx, int by, float* tile /* [32][32 * taps + 1] */) {
  const int row_len = 32 * taps, pitch = row_len + 1;
  const int ci0 = bx * 32, co0 = by * 32;
  const int K = Cin * taps;
  const int f4_per_row = row_len >> 2;       // row_len % 4 == 0
  // ---- load: rows are indexed by co for every mode (source row = 32*taps contiguous floats)
  for (int i = threadIdx.x; i < 32 * f4_per_row; i += 256) {
    const int r = i / f4_per_row, j = (i - r * f4_per_row) * 4;
    const int co = co0 + r;
    float4 v = make_float4(0, 0, 0, 0);
    if (co < Cout) {
      const long base = mode == 2 ? static_cast<long>(co) * K + static_cast<long>(bx) * row_len
                                  : (static_cast<long>(co) * Cin + ci0) * taps;
      v = ld4(src + base + j);
    }
    float* t = tile + r * pitch + j;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
  __syncthreads();
  // ---- store
  for (int i = threadIdx.x; i < 32 * f4_per_row; i += 256) {
    const int r = i / f4_per_row, j = (i - r * f4_per_row) * 4;
    float o[4];
    long base;
    if (mode == 0) {          // row = co; element j = tap * 32 + c  <-  tile[r][c * taps + tap]
      if (co0 + r >= Cout) continue;
      const int tap = j >> 5, c = j & 31;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = tile[r * pitch + (c + e) * taps + tap];
      base = static_cast<long>(co0 + r) * K + static_cast<long>(bx) * row_len;
    } else if (mode == 2) {   // row = co; element j = ci_l * taps + tap  <-  tile[r][tap * 32 + ci_l]
      if (co0 + r >= Cout) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = j + e, cl = q / taps, tap = q - cl * taps;
        o[e] = tile[r * pitch + tap * 32 + cl];
      }
      base = (static_cast<long>(co0 + r) * Cin + ci0) * taps;
    } else if (mode == 3) {   // rows (tap, ci_l) of 32 co each: handled below
      break;
    } else {                  // row = ci_l; element j = tap' * 32 + co_l  <-  tile[co_l][ci_l * taps + taps-1-tap']
      const int tp = j >> 5, c = j & 31;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = tile[(c + e) * pitch + r * taps + (taps - 1 - tp)];
      base = static_cast<long>(ci0 + r) * (static_cast<long>(Cout) * taps) + static_cast<long>(by) * row_len;
    }
    st4(dst + base + j, make_float4(o[0], o[1], o[2], o[3]));
  }
  if (mode == 3) {
    for (int i = threadIdx.x; i < 32 * taps * 8; i += 256) {
      const int r = i >> 3, j = (i & 7) * 4;
      const int tap = r >> 5, cl = r & 31;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = tile[(j + e) * pitch + cl * taps + tap];
      st4(dst + (static_cast<long>(tap) * Cin + ci0 + cl) * Cout + co0 + j, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
}

__global__ __launch_bounds__(256) void pack_tile_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout,
                                                        int Cin, int taps, int mode) {
  extern __shared__ float tile[];
  pack_tile_body(src, dst, Cout, Cin, taps, mode, blockIdx.x, blockIdx.y, tile);
}

// Every weight repack of a training step in ONE launch (the optimizer rewrites all parameters between steps, so each layer's
// forward, data-gradient and column layouts are rebuilt per step: ~165 launches of 6 us).  jobs: device table sorted by tile0;
// a workgroup finds its job by bisection.
__global__ __launch_bounds__(256) void pack_many_kernel(const diffsal_pack_job* __restrict__ jobs, int n_jobs) {
  extern __shared__ float tile[];
  const int t = blockIdx.x;
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
  }
  const diffsal_pack_job j = jobs[lo];
  const int lt = t - j.tile0, tx = j.Cin / 32;
  pack_tile_body(static_cast<const float*>(j.src), static_cast<float*>(j.dst), j.Cout, j.Cin, j.taps, j.mode, lt % tx, lt / tx, tile);
}

// dx[n, iy, ix, :] = cols[(n, oy, ox)][(ky, kx)][:] where (iy, ix) = (oy*s - pad + ky, ox*s - pad + kx) with ky < KH,
// kx < KW (at most one such pair because stride >= kernel), else 0.  Gather form: every dx element written once.
__global__ __launch_bounds__(256) void col2im_disjoint_kernel(const float* __restrict__ cols, float* __restrict__ dx, int N,
                                                              int H, int W, int C, int Ho, int Wo, int KH, int KW, int sh,
                                                              int sw, int pt, int pl) {
  const int c4n = C >> 2;
  const long total = static_cast<long>(N) * H * W * c4n;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c4 = static_cast<int>(i % c4n);
    long r = i / c4n;
    const int ix = static_cast<int>(r % W); r /= W;
    const int iy = static_cast<int>(r % H);
    const int n = static_cast<int>(r / H);
    const int ay = iy + pt, ax = ix + pl;
    const int oy = ay / sh, ky = ay - oy * sh, ox = ax / sw, kx = ax - ox * sw;
    float4 v = make_float4(0, 0, 0, 0);
    if (ky < KH && kx < KW && oy < Ho && ox < Wo)
      v = ld4(cols + ((static_cast<long>(n) * Ho + oy) * Wo + ox) * (static_cast<long>(KH) * KW * C) +
              static_cast<long>(ky * KW + kx) * C + c4 * 4);
    st4(dx + i * 4, v);
  }
}

// Overlapping taps (stride < kernel, e.g. the 3x3 stride-2 Downsample): every input pixel sums the (at most
// ceil(KH/sh) * ceil(KW/sw)) column entries that map onto it, in (ky, kx) order -- deterministic, no atomics, and the GEMM that
// produced `cols` did exactly the forward's MACs (a zero-inserted stride-1 convolution does stride_h * stride_w times more).
__global__ __launch_bounds__(256) void col2im_gather_kernel(const float* __restrict__ cols, float* __restrict__ dx, int N,
                                                            int H, int W, int C, int Ho, int Wo, int KH, int KW, int sh,
                                                            int sw, int pt, int pl) {
  const int c4n = C >> 2;
  const long total = static_cast<long>(N) * H * W * c4n;
  const long row_len = static_cast<long>(KH) * KW * C;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c4 = static_cast<int>(i % c4n);
    long r = i / c4n;
    const int ix = static_cast<int>(r % W); r /= W;
    const int iy = static_cast<int>(r % H);
    const int n = static_cast<int>(r / H);
    float4 v = make_float4(0, 0, 0, 0);
    for (int ky = 0; ky < KH; ++ky) {
      const int ay = iy + pt - ky;
      if (ay < 0 || ay % sh != 0 || ay / sh >= Ho) continue;
      for (int kx = 0; kx < KW; ++kx) {
        const int ax = ix + pl - kx;
        if (ax < 0 || ax % sw != 0 || ax / sw >= Wo) continue;
        const float4 t = ld4(cols + ((static_cast<long>(n) * Ho + ay / sh) * Wo + ax / sw) * row_len +
                             static_cast<long>(ky * KW + kx) * C + c4 * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    }
    st4(dx + i * 4, v);
  }
}

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// fp32 [..][32-k slices] -> per slice: 16 dwords of bf16 hi halves (k order) + 16 dwords of lo halves
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, long n4) {
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * 256) {
    const float4 v = ld4(src + i * 4);
    const long slice = i >> 3;
    const int pos = static_cast<int>(i & 7) * 4;          // first of 4 consecutive k inside the slice
    const __bf16 h0 = static_cast<__bf16>(v.x), h1 = static_cast<__bf16>(v.y), h2 = static_cast<__bf16>(v.z),
                 h3 = static_cast<__bf16>(v.w);
    const bf16x4_t hi = {h0, h1, h2, h3};
    const bf16x4_t lo = {static_cast<__bf16>(v.x - static_cast<float>(h0)), static_cast<__bf16>(v.y - static_cast<float>(h1)),
                         static_cast<__bf16>(v.z - static_cast<float>(h2)), static_cast<__bf16>(v.w - static_cast<float>(h3))};
    float* row = dst + slice * 32;
    *reinterpret_cast<uint2*>(row + (pos >> 1)) = __builtin_bit_cast(uint2, hi);
    *reinterpret_cast<uint2*>(row + 16 + (pos >> 1)) = __builtin_bit_cast(uint2, lo);
  }
}

}  // namespace
}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_pack_weight(const float* src, float* dst, int Cout, int Cin, int taps, int mode,
                                   diffsal_stream_t stream) {
  DS_REQUIRE(src && dst, DIFFSAL_E_ARG, "pack_weight: null argument");
  DS_REQUIRE(Cout > 0 && Cin > 0 && Cin % 32 == 0 && taps > 0 && taps <= 25 && mode >= 0 && mode <= 3, DIFFSAL_E_SHAPE,
             "pack_weight: bad shape Cout=%d Cin=%d taps=%d mode=%d (Cin %% 32 == 0, taps <= 25)", Cout, Cin, taps, mode);
  DS_REQUIRE((mode != 1 && mode != 3) || Cout % 32 == 0, DIFFSAL_E_SHAPE,
             "pack_weight: the data-gradient layouts need Cout %% 32 == 0, got %d", Cout);
  DS_REQUIRE(aligned16(src) && aligned16(dst), DIFFSAL_E_ARG, "pack_weight: buffers must be 16-byte aligned");
  const size_t lds = static_cast<size_t>(32) * (32 * taps + 1) * sizeof(float);
  if (lds > 64 * 1024) DS_RAISE_DYNAMIC_LDS((pack_tile_kernel), 128 * 1024);
  hipLaunchKernelGGL(pack_tile_kernel, dim3(Cin / 32, (Cout + 31) / 32), dim3(256), lds, static_cast<hipStream_t>(stream),
                     src, dst, Cout, Cin, taps, mode);
  return check_launch("pack_weight");
}

extern "C" int diffsal_pack_weight_many(const diffsal_pack_job* jobs_dev, int n_jobs, int total_tiles, int max_taps,
                                        diffsal_stream_t stream) {
  DS_REQUIRE(jobs_dev, DIFFSAL_E_ARG, "pack_weight_many: null job table");
  DS_REQUIRE(n_jobs > 0 && total_tiles > 0 && max_taps > 0 && max_taps <= 25, DIFFSAL_E_SHAPE,
             "pack_weight_many: n_jobs=%d total_tiles=%d max_taps=%d", n_jobs, total_tiles, max_taps);
  const size_t lds = static_cast<size_t>(32) * (32 * max_taps + 1) * sizeof(float);
  if (lds > 64 * 1024) DS_RAISE_DYNAMIC_LDS((pack_many_kernel), 128 * 1024);
  hipLaunchKernelGGL(pack_many_kernel, dim3(total_tiles), dim3(256), lds, static_cast<hipStream_t>(stream), jobs_dev, n_jobs);
  return check_launch("pack_weight_many");
}

extern "C" int diffsal_col2im_disjoint(const float* cols, float* dx, int N, int H, int W, int C, int Ho, int Wo, int KH,
                                       int KW, int stride_h, int stride_w, int pad_t, int pad_l, diffsal_stream_t stream) {
  DS_REQUIRE(cols && dx, DIFFSAL_E_ARG, "col2im_disjoint: null argument");
  DS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0, DIFFSAL_E_SHAPE,
             "col2im_disjoint: bad shape");
  DS_REQUIRE(stride_h >= KH && stride_w >= KW && pad_t >= 0 && pad_l >= 0, DIFFSAL_E_SHAPE,
             "col2im_disjoint: taps overlap (stride %dx%d < kernel %dx%d)", stride_h, stride_w, KH, KW);
  DS_REQUIRE(aligned16(cols) && aligned16(dx), DIFFSAL_E_ARG, "col2im_disjoint: buffers must be 16-byte aligned");
  const long total = static_cast<long>(N) * H * W * (C / 4);
  long g = (total + 255) / 256;
  g = g > 8192 ? 8192 : g;
  hipLaunchKernelGGL(col2im_disjoint_kernel, dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), cols,
                     dx, N, H, W, C, Ho, Wo, KH, KW, stride_h, stride_w, pad_t, pad_l);
  return check_launch("col2im_disjoint");
}

extern "C" int diffsal_col2im_gather(const float* cols, float* dx, int N, int H, int W, int C, int Ho, int Wo, int KH, int KW,
                                     int stride_h, int stride_w, int pad_t, int pad_l, diffsal_stream_t stream) {
  DS_REQUIRE(cols && dx, DIFFSAL_E_ARG, "col2im_gather: null argument");
  DS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0 && stride_h > 0 &&
                 stride_w > 0 && pad_t >= 0 && pad_l >= 0,
             DIFFSAL_E_SHAPE, "col2im_gather: bad shape");
  DS_REQUIRE(aligned16(cols) && aligned16(dx), DIFFSAL_E_ARG, "col2im_gather: buffers must be 16-byte aligned");
  const long total = static_cast<long>(N) * H * W * (C / 4);
  long g = (total + 255) / 256;
  g = g > 8192 ? 8192 : g;
  hipLaunchKernelGGL(col2im_gather_kernel, dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), cols, dx,
                     N, H, W, C, Ho, Wo, KH, KW, stride_h, stride_w, pad_t, pad_l);
  return check_launch("col2im_gather");
}

extern "C" int diffsal_split_weight(const float* src, float* dst, long n, diffsal_stream_t stream) {
  DS_REQUIRE(src && dst && src != dst, DIFFSAL_E_ARG, "split_weight: null or aliased argument");
  DS_REQUIRE(n > 0 && n % 32 == 0 && aligned16(src) && aligned16(dst), DIFFSAL_E_SHAPE,
             "split_weight: need n %% 32 == 0 and 16-byte aligned buffers");
  long g = (n / 4 + 255) / 256;
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(split_weight_kernel, dim3(static_cast<int>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst,
                     n / 4);
  return check_launch("split_weight");
}
