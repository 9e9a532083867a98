// Training-step tail on flat fp32 buffers: MSE loss + its gradient, global gradient norm, clip + Adam.
//
// Replaces, for the decoder's parameters: the MSE branch of get_kl_cc_sim_loss (R/models/sal_losses.py:189-192),
// torch.nn.utils.clip_grad_norm_ (R/diffusion_trainer.py:228-233) and torch.optim.Adam.step
// (R/diffusion_trainer.py:235, R/util/utils.py:116-123).  All three are HBM-streaming passes; reductions run in
// fp64 with a fixed combination order (block partials, then one block), so a step is bit-reproducible.
// The clip coefficient never visits the host: adam reads the norm the previous kernel left in device memory.
#include "common.h"

namespace diffsal {
namespace {

constexpr int kRedBlock = 256;

// Sum over the 256 threads of a block, fixed order (butterfly inside each wave, then waves 0..3 in order).
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < kRedBlock / kWave; ++i) t += sh[i];
  __syncthreads();
  return t;
}

// dpred = 2 scale (pred - target); part[block] = sum (pred - target)^2
__global__ __launch_bounds__(kRedBlock) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        float* __restrict__ dpred, double* __restrict__ part, long n4,
                                                        float gscale) {
  __shared__ double sh[kRedBlock / kWave];
  double acc = 0.0;
  for (long i = static_cast<long>(blockIdx.x) * kRedBlock + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * kRedBlock) {
    const float4 p = ld4(pred + i * 4), t = ld4(target + i * 4);
    const float4 d = make_float4(p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w);
    if (dpred) st4(dpred + i * 4, make_float4(gscale * d.x, gscale * d.y, gscale * d.z, gscale * d.w));
    acc += static_cast<double>(fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, d.w * d.w))));
  }
  const double s = block_sum_f64(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// part[block] = sum x^2
__global__ __launch_bounds__(kRedBlock) void sumsq_kernel(const float* __restrict__ x, double* __restrict__ part, long n4,
                                                          long n) {
  __shared__ double sh[kRedBlock / kWave];
  double acc = 0.0;
  for (long i = static_cast<long>(blockIdx.x) * kRedBlock + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * kRedBlock) {
    const float4 v = ld4(x + i * 4);
    acc += static_cast<double>(v.x) * v.x + static_cast<double>(v.y) * v.y + static_cast<double>(v.z) * v.z +
           static_cast<double>(v.w) * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < static_cast<int>(n - n4 * 4)) {
    const float v = x[n4 * 4 + threadIdx.x];
    acc += static_cast<double>(v) * v;
  }
  const double s = block_sum_f64(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = mode 0: scale * sum(part);  mode 1: scale * sqrt(sum(part))
__global__ __launch_bounds__(kRedBlock) void reduce_final_kernel(const double* __restrict__ part, int blocks,
                                                                 float* __restrict__ out, double scale, int mode) {
  __shared__ double sh[kRedBlock / kWave];
  double acc = 0.0;
  for (int i = threadIdx.x; i < blocks; i += kRedBlock) acc += part[i];
  const double s = block_sum_f64(acc, sh);
  if (threadIdx.x == 0) out[0] = static_cast<float>(mode == 1 ? scale * sqrt(s) : scale * s);
}

// out = x * s[0]  (chain-rule factor arriving as a device scalar)
__global__ __launch_bounds__(256) void scale_by_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                       float* __restrict__ out, long n4) {
  const float k = s[0];
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * 256) {
    const float4 v = ld4(x + i * 4);
    st4(out + i * 4, make_float4(v.x * k, v.y * k, v.z * k, v.w * k));
  }
}

// Gather up to kMultiCopyMax separately allocated gradient tensors into their slots of the flat buffer in one launch
// (the pointer table travels in the kernel arguments: no host-to-device copy, no synchronisation).
constexpr int kMultiCopyMax = 48;
struct MultiCopyArgs {
  const float* src[kMultiCopyMax];
  long dst_off[kMultiCopyMax];
  long n[kMultiCopyMax];
};

__global__ __launch_bounds__(256) void multi_copy_kernel(MultiCopyArgs a, float* __restrict__ dst) {
  const int t = blockIdx.y;
  const float* __restrict__ s = a.src[t];
  float* __restrict__ d = dst + a.dst_off[t];
  const long n = a.n[t], n4 = ((reinterpret_cast<uintptr_t>(s) & 15u) == 0) ? (n >> 2) : 0;   // dst slots are 256-B aligned
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * 256)
    st4(d + i * 4, ld4(s + i * 4));
  for (long i = n4 * 4 + static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long>(gridDim.x) * 256)
    d[i] = s[i];
}

struct AdamArgs {
  float gscale, omb1, beta2, omb2, eps, wd, step_size, bc2_sqrt, max_norm;  // omb = 1 - beta, rounded from double
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
  g *= coef;
  if (a.wd != 0.f) g = fmaf(a.wd, p, g);
  m = fmaf(g - m, a.omb1, m);                        // exp_avg.lerp_(grad, 1 - beta1)
  v = fmaf(a.omb2, g * g, v * a.beta2);              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = fmaf(-a.step_size, m / denom, p);              // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ norm, long n4,
                                                   long n, AdamArgs a, int store_grad) {
  // clip_grad_norm_: coef = min(1, max_norm / (total_norm + 1e-6)); the DDP average is folded in as gscale
  float coef = a.gscale;
  if (norm != nullptr && a.max_norm > 0.f) coef *= fminf(1.f, a.max_norm / (norm[0] + 1e-6f));
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * 256) {
    float4 pv = ld4(p + i * 4), mv = ld4(m + i * 4), vv = ld4(v + i * 4);
    const float4 gv = ld4(g + i * 4);
    adam_one(pv.x, gv.x, mv.x, vv.x, a, coef);
    adam_one(pv.y, gv.y, mv.y, vv.y, a, coef);
    adam_one(pv.z, gv.z, mv.z, vv.z, a, coef);
    adam_one(pv.w, gv.w, mv.w, vv.w, a, coef);
    st4(p + i * 4, pv); st4(m + i * 4, mv); st4(v + i * 4, vv);
    if (store_grad) st4(g + i * 4, make_float4(gv.x * coef, gv.y * coef, gv.z * coef, gv.w * coef));
  }
  if (blockIdx.x == 0 && threadIdx.x < static_cast<int>(n - n4 * 4)) {
    const long i = n4 * 4 + threadIdx.x;
    float pv = p[i], mv = m[i], vv = v[i];
    const float gv = g[i];
    adam_one(pv, gv, mv, vv, a, coef);
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (store_grad) g[i] = gv * coef;
  }
}

int red_grid(long n4, int blocks) {
  long need = (n4 + kRedBlock - 1) / kRedBlock;
  if (need < 1) need = 1;
  return static_cast<int>(need < blocks ? need : blocks);
}

}  // namespace
}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_reduce_blocks(void) { return 1024; }

extern "C" int diffsal_mse_loss(const float* pred, const float* target, float* dpred, float* loss, double* part, long n,
                                float loss_scale, diffsal_stream_t stream) {
  DS_REQUIRE(pred && target && loss && part, DIFFSAL_E_ARG, "mse_loss: null argument");
  DS_REQUIRE(n > 0 && n % 4 == 0, DIFFSAL_E_SHAPE, "mse_loss: element count %ld must be a positive multiple of 4", n);
  DS_REQUIRE(aligned16(pred) && aligned16(target) && (!dpred || aligned16(dpred)), DIFFSAL_E_ARG,
             "mse_loss: pointers must be 16-byte aligned");
  const int blocks = red_grid(n / 4, diffsal_reduce_blocks());
  hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(kRedBlock), 0, static_cast<hipStream_t>(stream), pred, target, dpred,
                     part, n / 4, 2.f * loss_scale);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kRedBlock), 0, static_cast<hipStream_t>(stream), part, blocks, loss,
                     static_cast<double>(loss_scale), 0);
  return check_launch("mse_loss");
}

extern "C" int diffsal_scale_by(const float* x, const float* s, float* out, long n, diffsal_stream_t stream) {
  DS_REQUIRE(x && s && out, DIFFSAL_E_ARG, "scale_by: null argument");
  DS_REQUIRE(n > 0 && n % 4 == 0 && aligned16(x) && aligned16(out), DIFFSAL_E_SHAPE,
             "scale_by: need n %% 4 == 0 and 16-byte aligned buffers");
  long need = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(scale_by_kernel, dim3(static_cast<int>(need < 4096 ? need : 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, s, out, n / 4);
  return check_launch("scale_by");
}

extern "C" int diffsal_grad_norm(const float* g, long n, float gscale, float* norm, double* part, diffsal_stream_t stream) {
  DS_REQUIRE(g && norm && part, DIFFSAL_E_ARG, "grad_norm: null argument");
  DS_REQUIRE(n > 0 && aligned16(g), DIFFSAL_E_SHAPE, "grad_norm: need n > 0 and a 16-byte aligned buffer");
  const int blocks = red_grid(n / 4, diffsal_reduce_blocks());
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(kRedBlock), 0, static_cast<hipStream_t>(stream), g, part, n / 4, n);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kRedBlock), 0, static_cast<hipStream_t>(stream), part, blocks, norm,
                     static_cast<double>(gscale), 1);
  return check_launch("grad_norm");
}

extern "C" int diffsal_adam_step(float* p, float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                                 double eps, double weight_decay, int step, float gscale, const float* norm, float max_norm,
                                 int store_clipped_grad, diffsal_stream_t stream) {
  DS_REQUIRE(p && g && m && v, DIFFSAL_E_ARG, "adam_step: null argument");
  DS_REQUIRE(n > 0 && step >= 1, DIFFSAL_E_SHAPE, "adam_step: need n > 0 and step >= 1");
  DS_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), DIFFSAL_E_ARG,
             "adam_step: buffers must be 16-byte aligned");
  // bias corrections in double on the host, as torch.optim.Adam does for a python-scalar step
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  AdamArgs a;
  a.gscale = gscale; a.omb1 = static_cast<float>(1.0 - beta1); a.beta2 = static_cast<float>(beta2);
  a.omb2 = static_cast<float>(1.0 - beta2); a.eps = static_cast<float>(eps); a.wd = static_cast<float>(weight_decay);
  a.step_size = static_cast<float>(lr / bc1);
  a.bc2_sqrt = static_cast<float>(sqrt(bc2));
  a.max_norm = max_norm;
  long need = (n / 4 + 255) / 256;
  if (need < 1) need = 1;
  const int grid = static_cast<int>(need < 256L * 16 ? need : 256L * 16);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, norm, n / 4, n, a,
                     store_clipped_grad);
  return check_launch("adam_step");
}

extern "C" int diffsal_multi_copy_max(void) { return kMultiCopyMax; }

extern "C" int diffsal_multi_copy(const float* const* srcs /*host array*/, const long* dst_offsets /*host*/,
                                  const long* sizes /*host*/, int count, float* dst, diffsal_stream_t stream) {
  DS_REQUIRE(srcs && dst_offsets && sizes && dst, DIFFSAL_E_ARG, "multi_copy: null argument");
  DS_REQUIRE(count >= 0, DIFFSAL_E_SHAPE, "multi_copy: negative count");
  for (int base = 0; base < count; base += kMultiCopyMax) {
    const int m = count - base < kMultiCopyMax ? count - base : kMultiCopyMax;
    MultiCopyArgs a;
    long biggest = 1;
    for (int i = 0; i < m; ++i) {
      DS_REQUIRE(srcs[base + i] && sizes[base + i] >= 0 && dst_offsets[base + i] >= 0 && dst_offsets[base + i] % 4 == 0,
                 DIFFSAL_E_ARG, "multi_copy: bad entry %d", base + i);
      a.src[i] = srcs[base + i]; a.dst_off[i] = dst_offsets[base + i]; a.n[i] = sizes[base + i];
      if (a.n[i] > biggest) biggest = a.n[i];
    }
    for (int i = m; i < kMultiCopyMax; ++i) { a.src[i] = nullptr; a.dst_off[i] = 0; a.n[i] = 0; }
    long gx = (biggest / 4 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipLaunchKernelGGL(multi_copy_kernel, dim3(static_cast<int>(gx), m), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                       dst);
  }
  return check_launch("multi_copy");
}
