// On-device saliency evaluation metrics: CC, SIM, NSS, KL-div per image and their batch means.
// Replaces the torch reductions of R/models/sal_losses.py:14-176 (nss2, cc_s2, kldiv2, normalize_map2,
// similarity2) that the reference trainer runs for every validation batch
// (get_kl_cc_sim_loss_wo_weight, R/diffusion_trainer.py:741,797,868).
//
// Two streaming passes over (pred, gt), both HBM-bound (8 B/pixel/pass), fp64 accumulation, fixed combination order:
//   pass 1: per (image, chunk) moments  sum s, sum s^2, sum g, sum g^2, sum s g, min/max s, min/max g
//   pass 2: per (image, chunk)          sum min(s~, g~)  and  sum g^ log(eps + g^ / (s^ + eps))
//           with s~ = min-max normalised and sum-normalised map, s^ = sum-normalised map (scalars from pass 1)
//   final : CC and NSS in closed form from the moments (the per-image std cancels in CC), batch means.
#include "common.h"

namespace diffsal {

constexpr int MT_CHUNKS = 64;   // workgroups per image
constexpr int MOM = 9;          // doubles per pass-1 partial
constexpr int P2 = 6;           // doubles per pass-2 partial: sim, kl, sum c s~, sum c, sum a s^ (KL), first index of min s

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
  return v;
}

// block-wide reduce of NV doubles per thread (sum for i < NS, min for NS <= i < NS+NMIN, max after), result in sh[0..NV)
template <int NV, int NS, int NMIN>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = i < NS ? wave_sum_d(v[i]) : (i < NS + NMIN ? wave_min_d(v[i]) : wave_max_d(v[i]));
    if (lane == 0) sh[wave * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    double r = sh[i];
    for (int w = 1; w < 4; ++w) {
      const double o = sh[w * NV + i];
      r = i < NS ? r + o : (i < NS + NMIN ? fmin(r, o) : fmax(r, o));
    }
    sh[4 * NV + i] = r;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void metric_moments_kernel(const float* __restrict__ s, const float* __restrict__ g,
                                                             double* __restrict__ part, long n) {
  __shared__ double sh[5 * MOM];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const long lo = n * chunk / MT_CHUNKS, hi = n * (chunk + 1) / MT_CHUNKS;
  const float* sb = s + static_cast<long>(b) * n;
  const float* gb = g + static_cast<long>(b) * n;
  double v[MOM] = {0, 0, 0, 0, 0, 1e300, 1e300, -1e300, -1e300};
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const double a = sb[i], c = gb[i];
    v[0] += a; v[1] += a * a; v[2] += c; v[3] += c * c; v[4] += a * c;
    v[5] = fmin(v[5], a); v[6] = fmin(v[6], c); v[7] = fmax(v[7], a); v[8] = fmax(v[8], c);
  }
  block_reduce<MOM, 5, 2>(v, sh);
  if (threadIdx.x < MOM) part[(static_cast<long>(b) * MT_CHUNKS + chunk) * MOM + threadIdx.x] = sh[4 * MOM + threadIdx.x];
}

__device__ __forceinline__ void image_moments(const double* __restrict__ part, int b, double (&m)[MOM]) {
  for (int i = 0; i < MOM; ++i) m[i] = i < 5 ? 0.0 : (i < 7 ? 1e300 : -1e300);
  for (int c = 0; c < MT_CHUNKS; ++c) {   // fixed order
    const double* p = part + (static_cast<long>(b) * MT_CHUNKS + c) * MOM;
    for (int i = 0; i < 5; ++i) m[i] += p[i];
    m[5] = fmin(m[5], p[5]); m[6] = fmin(m[6], p[6]); m[7] = fmax(m[7], p[7]); m[8] = fmax(m[8], p[8]);
  }
}

__global__ __launch_bounds__(256) void metric_pass2_kernel(const float* __restrict__ s, const float* __restrict__ g,
                                                           const double* __restrict__ part, double* __restrict__ part2,
                                                           long n) {
  __shared__ double sh[5 * P2];
  const int b = blockIdx.y, chunk = blockIdx.x;
  double m[MOM];
  image_moments(part, b, m);
  const double nn = static_cast<double>(n);
  // similarity2: (x - min) / (max - min), then / its sum      (sal_losses.py:134-176)
  const double rs = m[7] - m[5], rg = m[8] - m[6];
  const double sum_sn = (m[0] - nn * m[5]) / rs, sum_gn = (m[2] - nn * m[6]) / rg;
  // kldiv2: x / sum x                                         (sal_losses.py:100-131)
  const double eps = 2.2204e-16;
  const long lo = n * chunk / MT_CHUNKS, hi = n * (chunk + 1) / MT_CHUNKS;
  const float* sb = s + static_cast<long>(b) * n;
  const float* gb = g + static_cast<long>(b) * n;
  // v[2..4] feed the backward (diffsal_saliency_metrics_bwd): c = d min(s~, g~) / d s~ (1 below, 1/2 on a tie, 0 above: torch's
  // `minimum`), a = d KL / d s^; v[5] = first pixel that attains min s (torch.min(dim)'s index: the gradient of the min goes there)
  double v[P2] = {0, 0, 0, 0, 0, 1e300};
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const double a = sb[i], c = gb[i];
    const double st = (a - m[5]) / rs / sum_sn, gt = (c - m[6]) / rg / sum_gn;
    v[0] += fmin(st, gt);
    const double sk = a / m[0], gk = c / m[2];
    v[1] += gk * log(eps + gk / (sk + eps));
    const double cm = st < gt ? 1.0 : (st == gt ? 0.5 : 0.0);
    v[2] += cm * st;
    v[3] += cm;
    const double ak = -gk * gk / ((sk + eps) * (sk + eps) * (eps + gk / (sk + eps)));
    v[4] += ak * sk;
    if (a == m[5]) v[5] = fmin(v[5], static_cast<double>(i));
  }
  block_reduce<P2, 5, 1>(v, sh);
  if (threadIdx.x < P2) part2[(static_cast<long>(b) * MT_CHUNKS + chunk) * P2 + threadIdx.x] = sh[4 * P2 + threadIdx.x];
}

// d(w_cc CC + w_sim SIM + w_nss NSS + w_kl KL) / d pred for every pixel (the four terms are batch MEANS: each image carries 1 / B).
// Closed forms in the moments of pass 1 and the sums of pass 2 (fp64 per pixel; derivations in DESIGN.md):
//   CC  = csg / sqrt(css cgg)                  d/ds_i = (g_i - mu_g) / sqrt(css cgg) - CC (s_i - mu_s) / css
//   NSS = A / ((sigma + eps) G), A = sum (s - mu_s) g
//                                             d/ds_i = (g_i - G/n) / ((sigma+eps) G) - A (s_i - mu_s) / ((n-1) sigma (sigma+eps)^2 G)
//   KL  = sum g^ log(eps + g^ / (s^ + eps)), s^ = s / S      d/ds_i = (a_i - sum_j a_j s^_j) / S
//   SIM = sum min(s~, g~), s~ = (s - min s) / (S - n min s)  (the range of normalize_map2 cancels)
//                                             d/ds_i = (c_i - P) / U - [i = argmin] (sum c - n P) / U,  U = S - n min s, P = sum c s~
__global__ __launch_bounds__(256) void metric_bwd_kernel(const float* __restrict__ s, const float* __restrict__ g,
                                                         const double* __restrict__ part, const double* __restrict__ part2,
                                                         const float* __restrict__ w4, float* __restrict__ ds, long n, int B) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  double m[MOM];
  image_moments(part, b, m);
  double q[P2] = {0, 0, 0, 0, 0, 1e300};
  for (int c = 0; c < MT_CHUNKS; ++c) {
    const double* pp = part2 + (static_cast<long>(b) * MT_CHUNKS + c) * P2;
    for (int i = 0; i < 5; ++i) q[i] += pp[i];
    q[5] = fmin(q[5], pp[5]);
  }
  const double nn = static_cast<double>(n), eps = 2.2204e-16, invB = 1.0 / B;
  const double wcc = w4[0] * invB, wsim = w4[1] * invB, wnss = w4[2] * invB, wkl = w4[3] * invB;
  const double mu_s = m[0] / nn, mu_g = m[2] / nn;
  const double css = m[1] - nn * mu_s * mu_s, cgg = m[3] - nn * mu_g * mu_g, csg = m[4] - nn * mu_s * mu_g;
  const double cc = csg / sqrt(css * cgg), inv_sc = 1.0 / sqrt(css * cgg);
  const double sigma = sqrt(css / (nn - 1.0)), A = m[4] - mu_s * m[2];
  const double nss_a = 1.0 / ((sigma + eps) * m[2]), nss_b = A / ((nn - 1.0) * sigma * (sigma + eps) * (sigma + eps) * m[2]);
  const double rs = m[7] - m[5], rg = m[8] - m[6];
  const double sum_sn = (m[0] - nn * m[5]) / rs, sum_gn = (m[2] - nn * m[6]) / rg;
  const double U = m[0] - nn * m[5];
  const double Pq = q[2], Csum = q[3], Qkl = q[4];
  const long imin = static_cast<long>(q[5]);
  const long lo = n * chunk / MT_CHUNKS, hi = n * (chunk + 1) / MT_CHUNKS;
  const float* sb = s + static_cast<long>(b) * n;
  const float* gb = g + static_cast<long>(b) * n;
  float* db = ds + static_cast<long>(b) * n;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const double a = sb[i], c = gb[i];
    double gr = 0.0;
    if (wcc != 0.0) gr += wcc * ((c - mu_g) * inv_sc - cc * (a - mu_s) / css);
    if (wnss != 0.0) gr += wnss * ((c - m[2] / nn) * nss_a - nss_b * (a - mu_s));
    if (wkl != 0.0) {
      const double sk = a / m[0], gk = c / m[2];
      const double ak = -gk * gk / ((sk + eps) * (sk + eps) * (eps + gk / (sk + eps)));
      gr += wkl * (ak - Qkl) / m[0];
    }
    if (wsim != 0.0) {
      const double st = (a - m[5]) / rs / sum_sn, gt = (c - m[6]) / rg / sum_gn;
      const double cm = st < gt ? 1.0 : (st == gt ? 0.5 : 0.0);
      gr += wsim * ((cm - Pq) / U - (i == imin ? (Csum - nn * Pq) / U : 0.0));
    }
    db[i] = static_cast<float>(gr);
  }
}

// one thread per image, then thread 0 averages over the batch in index order
__global__ void metric_final_kernel(const double* __restrict__ part, const double* __restrict__ part2,
                                    float* __restrict__ per_image, float* __restrict__ mean_out, int B, long n) {
  extern __shared__ double shm[];  // [B][4]
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    double m[MOM];
    image_moments(part, b, m);
    const double nn = static_cast<double>(n);
    const double mu_s = m[0] / nn, mu_g = m[2] / nn;
    const double css = m[1] - nn * mu_s * mu_s, cgg = m[3] - nn * mu_g * mu_g, csg = m[4] - nn * mu_s * mu_g;
    const double cc = csg / sqrt(css * cgg);                 // std_s, std_g cancel in ab / sqrt(aa bb)  (:63-97)
    const double std_s = sqrt(css / (nn - 1.0));             // torch.std: unbiased
    const double nss = (m[4] - mu_s * m[2]) / (std_s + 2.2204e-16) / m[2];   // (:14-37)
    double sim = 0, kl = 0;
    for (int c = 0; c < MT_CHUNKS; ++c) {
      sim += part2[(static_cast<long>(b) * MT_CHUNKS + c) * P2];
      kl += part2[(static_cast<long>(b) * MT_CHUNKS + c) * P2 + 1];
    }
    shm[b * 4 + 0] = cc; shm[b * 4 + 1] = sim; shm[b * 4 + 2] = nss; shm[b * 4 + 3] = kl;
    if (per_image) {
      per_image[b * 4 + 0] = static_cast<float>(cc); per_image[b * 4 + 1] = static_cast<float>(sim);
      per_image[b * 4 + 2] = static_cast<float>(nss); per_image[b * 4 + 3] = static_cast<float>(kl);
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double a = 0;
    for (int b = 0; b < B; ++b) a += shm[b * 4 + threadIdx.x];
    mean_out[threadIdx.x] = static_cast<float>(a / B);
  }
}

}  // namespace diffsal

using namespace diffsal;

extern "C" size_t diffsal_saliency_metrics_ws_bytes(int B) {
  return static_cast<size_t>(B) * MT_CHUNKS * (MOM + P2) * sizeof(double);
}

extern "C" int diffsal_saliency_metrics(const float* pred, const float* gt, int B, long n, void* ws, size_t ws_bytes,
                                        float* per_image, float* mean_out, diffsal_stream_t stream) {
  DS_REQUIRE(pred && gt && ws && mean_out, DIFFSAL_E_ARG, "saliency_metrics: null argument");
  // the final kernel keeps 4 doubles per image in LDS (64 KiB without opting in to more): 2048 images per call
  DS_REQUIRE(B > 0 && B <= 2048 && n > 1, DIFFSAL_E_SHAPE, "saliency_metrics: bad shape B=%d (1..2048) n=%ld", B, n);
  DS_REQUIRE(ws_bytes >= diffsal_saliency_metrics_ws_bytes(B) && aligned16(ws), DIFFSAL_E_ARG,
             "saliency_metrics: workspace too small or misaligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(ws);
  double* part2 = part + static_cast<long>(B) * MT_CHUNKS * MOM;
  hipLaunchKernelGGL(metric_moments_kernel, dim3(MT_CHUNKS, B), dim3(256), 0, s, pred, gt, part, n);
  int rc = check_launch("saliency_metrics(moments)");
  if (rc) return rc;
  hipLaunchKernelGGL(metric_pass2_kernel, dim3(MT_CHUNKS, B), dim3(256), 0, s, pred, gt, part, part2, n);
  rc = check_launch("saliency_metrics(pass 2)");
  if (rc) return rc;
  hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(256), static_cast<size_t>(B) * 4 * sizeof(double), s, part, part2,
                     per_image, mean_out, B, n);
  return check_launch("saliency_metrics(final)");
}

// Backward of the four batch-mean terms of diffsal_saliency_metrics with respect to pred: `ws` is the workspace that call left
// behind (same pred / gt), weights4 = (w_cc, w_sim, w_nss, w_kl) on the DEVICE (the upstream gradients of the four means,
// already multiplied by the loss weights), dpred [B, n] fp32.  get_kl_cc_sim_loss / get_lossv2, R/models/sal_losses.py:179-259.
extern "C" int diffsal_saliency_metrics_bwd(const float* pred, const float* gt, int B, long n, const void* ws, size_t ws_bytes,
                                            const float* weights4, float* dpred, diffsal_stream_t stream) {
  DS_REQUIRE(pred && gt && ws && weights4 && dpred, DIFFSAL_E_ARG, "saliency_metrics_bwd: null argument");
  DS_REQUIRE(B > 0 && B <= 2048 && n > 1, DIFFSAL_E_SHAPE, "saliency_metrics_bwd: bad shape B=%d n=%ld", B, n);
  DS_REQUIRE(ws_bytes >= diffsal_saliency_metrics_ws_bytes(B) && aligned16(ws), DIFFSAL_E_ARG,
             "saliency_metrics_bwd: workspace too small or misaligned");
  const double* part = static_cast<const double*>(ws);
  const double* part2 = part + static_cast<long>(B) * MT_CHUNKS * MOM;
  hipLaunchKernelGGL(metric_bwd_kernel, dim3(MT_CHUNKS, B), dim3(256), 0, static_cast<hipStream_t>(stream), pred, gt, part, part2,
                     weights4, dpred, n, B);
  return check_launch("saliency_metrics_bwd");
}
