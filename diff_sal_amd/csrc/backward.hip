// Backward kernels of the HBM-bound operators (training step).  Each mirrors a forward kernel in
// norm.hip / misc.hip; gradients are exact fp32.  References are the autograd semantics of the PyTorch modules
// the forward kernels replace (see diffsal.h).
#include "common.h"

namespace diffsal {

static int ew_grid_b(long total) {
  long g = (total + 255) / 256;
  return static_cast<int>(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

// dx = dy * f'(.)  ;  mode 1: ReLU (ref = y), 2: GELU-erf (ref = pre-activation x), 3: sigmoid (ref = y)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref,
                                                      float* __restrict__ dx, long n4, int mode) {
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * 256) {
    const float4 g = ld4(dy + i * 4), r = ld4(ref + i * 4);
    float gv[4] = {g.x, g.y, g.z, g.w}, rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float d;
      if (mode == 1) d = rv[c] > 0.f ? 1.f : 0.f;
      else if (mode == 2) d = gelu_erf_grad(rv[c]);
      else d = rv[c] * (1.0f - rv[c]);
      gv[c] *= d;
    }
    st4(dx + i * 4, make_float4(gv[0], gv[1], gv[2], gv[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// Per-channel dual sums over rows, the reduction half of every normalisation layer's forward/backward.
//   mode 0: (x, x^2)                                   BatchNorm / GroupNorm forward statistics
//   mode 1: (dz, dz*xhat), dz = dy * [y > 0]           BatchNorm+ReLU backward   (mu, rs: [1][C])
//   mode 2: (dz, dz*xhat), dz = dy * swish'(z)         GroupNorm+swish backward  (mu, rs: [seg][C], z = xhat*gamma+beta)
//   mode 3: (dy, dy*xhat)                              plain normalisation backward
// part: [segments][chunks][2][C]; rows of segment g are [g*seg_rows, (g+1)*seg_rows).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float swish_grad(float z) {
  const float sg = 1.0f / (1.0f + expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}

__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ y, const float* __restrict__ mu,
                                                       const float* __restrict__ rs, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, double* __restrict__ part, int M,
                                                       int C, int seg_rows, int chunks, int mode, int stat_per_seg) {
  // per-thread partial sums are fp32 over a handful of rows; everything across threads / workgroups is fp64:
  // gradient sums such as d(beta) cancel heavily (sum << sum of magnitudes) and must not depend on atomics order
  extern __shared__ double shd[];  // [2][C]
  const int seg = blockIdx.y, chunk = blockIdx.x;
  const int c4n = C >> 2;
  const int rpp = 256 / c4n > 0 ? 256 / c4n : 1;
  const int c4 = threadIdx.x % c4n, rsub = threadIdx.x / c4n;
  const int c = c4 * 4;
  const long r0 = static_cast<long>(seg) * seg_rows;
  const long rb = r0 + static_cast<long>(seg_rows) * chunk / chunks;
  const long re = r0 + static_cast<long>(seg_rows) * (chunk + 1) / chunks;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (rsub < rpp) {
    float m4[4] = {0, 0, 0, 0}, r4[4] = {1, 1, 1, 1}, g4[4] = {1, 1, 1, 1}, b4[4] = {0, 0, 0, 0};
    if (mode != 0) {
      const long so = (stat_per_seg ? static_cast<long>(seg) * C : 0) + c;
#pragma unroll
      for (int k = 0; k < 4; ++k) { m4[k] = mu[so + k]; r4[k] = rs[so + k]; }
      if (mode == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { g4[k] = gamma[c + k]; b4[k] = beta[c + k]; }
      }
    }
    for (long m = rb + rsub; m < re; m += rpp) {
      const float4 xv = ld4(x + m * C + c);
      const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
      if (mode == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s1[k] += xa[k]; s2[k] += xa[k] * xa[k]; }
      } else {
        const float4 gv = ld4(dy + m * C + c);
        const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
        float ya[4] = {1, 1, 1, 1};
        if (mode == 1) { const float4 yv = ld4(y + m * C + c); ya[0] = yv.x; ya[1] = yv.y; ya[2] = yv.z; ya[3] = yv.w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (xa[k] - m4[k]) * r4[k];
          float dz = ga[k];
          if (mode == 1) dz = ya[k] > 0.f ? dz : 0.f;
          else if (mode == 2) dz *= swish_grad(xh * g4[k] + b4[k]);
          s1[k] += dz;
          s2[k] += dz * xh;
        }
      }
    }
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) shd[i] = 0.0;
  __syncthreads();
  if (rsub < rpp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(&shd[c + k], static_cast<double>(s1[k]));
      atomicAdd(&shd[C + c + k], static_cast<double>(s2[k]));
    }
  }
  __syncthreads();
  double* o = part + (static_cast<long>(seg) * chunks + chunk) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) o[i] = shd[i];
}

// out[seg][j] = sum over chunks of part[seg][chunk][j] (fp64 in, fp64 or fp32 out).  Workgroup = (8 columns, segment),
// 1024 threads: 128 thread groups take every 128th chunk (8 loads in flight per thread at 1024 chunks: the kernel is
// pure load latency), then 32 threads per column add 4 group sums each and one adds those 32 -- fixed order, short chains,
// and enough lanes even when there are only ~200 columns.
constexpr int RP_GROUPS = 128;
template <typename OUT>
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const double* __restrict__ part, OUT* __restrict__ out,
                                                               int chunks, int Wd) {
  __shared__ double sh[RP_GROUPS][9];
  __shared__ double sh2[32][9];
  const int cl = threadIdx.x & 7, kg = threadIdx.x >> 3;
  const int j = blockIdx.x * 8 + cl, g = blockIdx.y;
  double s = 0.0;
  if (j < Wd) {
    const double* col = part + static_cast<long>(g) * chunks * Wd + j;
    int k = kg;
    for (; k + 3 * RP_GROUPS < chunks; k += 4 * RP_GROUPS) {      // four independent loads per trip
      const double a = col[static_cast<long>(k) * Wd], b = col[static_cast<long>(k + RP_GROUPS) * Wd];
      const double c = col[static_cast<long>(k + 2 * RP_GROUPS) * Wd], d = col[static_cast<long>(k + 3 * RP_GROUPS) * Wd];
      s += (a + b) + (c + d);
    }
    for (; k < chunks; k += RP_GROUPS) s += col[static_cast<long>(k) * Wd];
  }
  sh[kg][cl] = s;
  __syncthreads();
  if (kg < 32) sh2[kg][cl] = (sh[4 * kg][cl] + sh[4 * kg + 1][cl]) + (sh[4 * kg + 2][cl] + sh[4 * kg + 3][cl]);
  __syncthreads();
  if (kg == 0 && j < Wd) {
    double t = sh2[0][cl];
#pragma unroll
    for (int q = 1; q < 32; ++q) t += sh2[q][cl];
    out[static_cast<long>(g) * Wd + j] = static_cast<OUT>(t);
  }
}

// GroupNorm / BatchNorm statistics -> per-(segment, channel) vectors, one workgroup for everything (tiny).
//   sums[seg][2][C] = (sum x, sum x^2) per channel; groups of cpg = C/G channels share statistics over n = rows*cpg
//   values (BatchNorm: G = C, one segment).  mu, rs, scale = rs*gamma, shift = beta - mu*scale.
//   BatchNorm extras (bn_mean != NULL): batch mean / biased var out, running stats updated like nn.BatchNorm2d.
__global__ __launch_bounds__(256) void norm_finalize_fwd_kernel(const double* __restrict__ sums,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ mu,
                                                                float* __restrict__ rs, float* __restrict__ scale,
                                                                float* __restrict__ shift, int segs, int C, int G, double n,
                                                                double eps, float* __restrict__ bn_mean,
                                                                float* __restrict__ bn_var, float* __restrict__ run_mean,
                                                                float* __restrict__ run_var, float momentum, float unbias) {
  const int cpg = C / G;
  for (int i = threadIdx.x; i < segs * C; i += 256) {
    const int sg = i / C, c = i - sg * C;
    const int g0 = (c / cpg) * cpg;
    double s1 = 0.0, s2 = 0.0;
    for (int q = 0; q < cpg; ++q) {
      s1 += sums[(static_cast<long>(sg) * 2 + 0) * C + g0 + q];
      s2 += sums[(static_cast<long>(sg) * 2 + 1) * C + g0 + q];
    }
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float m = static_cast<float>(mean), r = static_cast<float>(1.0 / sqrt(var + eps));
    const float sc = r * gamma[c];
    mu[i] = m; rs[i] = r; scale[i] = sc; shift[i] = beta[c] - m * sc;
    if (bn_mean != nullptr) {
      const float vf = static_cast<float>(var);
      bn_mean[i] = m; bn_var[i] = vf;
      if (run_mean != nullptr) {
        run_mean[c] = run_mean[c] * (1.f - momentum) + momentum * m;
        run_var[c] = run_var[c] * (1.f - momentum) + momentum * (vf * unbias);
      }
    }
  }
}

// Backward companion: t[seg][2][C] = (sum dz, sum dz*xhat) -> dbeta = sum_seg t0, dgamma = sum_seg t1,
//   k1 = rs*gamma, k2 = rs * groupsum(t0*gamma)/n, k3 = rs * groupsum(t1*gamma)/n   (inputs of norm_bwd_apply)
__global__ __launch_bounds__(256) void norm_finalize_bwd_kernel(const double* __restrict__ t, const float* __restrict__ gamma,
                                                                const float* __restrict__ rs, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float* __restrict__ k1,
                                                                float* __restrict__ k2, float* __restrict__ k3, int segs,
                                                                int C, int G, double n) {
  const int cpg = C / G;
  for (int i = threadIdx.x; i < segs * C; i += 256) {
    const int sg = i / C, c = i - sg * C;
    const int g0 = (c / cpg) * cpg;
    double a = 0.0, b = 0.0;
    for (int q = 0; q < cpg; ++q) {
      const double gq = static_cast<double>(gamma[g0 + q]);
      a += t[(static_cast<long>(sg) * 2 + 0) * C + g0 + q] * gq;
      b += t[(static_cast<long>(sg) * 2 + 1) * C + g0 + q] * gq;
    }
    const double r = static_cast<double>(rs[i]);
    k1[i] = rs[i] * gamma[c];
    k2[i] = static_cast<float>(r * (a / n));
    k3[i] = static_cast<float>(r * (b / n));
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    double d0 = 0.0, d1 = 0.0;
    for (int sg = 0; sg < segs; ++sg) {
      d0 += t[(static_cast<long>(sg) * 2 + 0) * C + c];
      d1 += t[(static_cast<long>(sg) * 2 + 1) * C + c];
    }
    dbeta[c] = static_cast<float>(d0);
    dgamma[c] = static_cast<float>(d1);
  }
}

// y = act(x * s[seg, c] + t[seg, c])   (BatchNorm / GroupNorm forward once the statistics are folded into s, t)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                         const float* __restrict__ sh_, float* __restrict__ out, long M,
                                                         int C, int seg_rows, int act) {
  const int c4n = C >> 2;
  const long total = M * c4n;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    const long m = i / c4n;
    const long so = (m / seg_rows) * C + c;
    const float4 v = ld4(x + m * C + c), s4 = ld4(sc + so), t4 = ld4(sh_ + so);
    float o[4] = {v.x * s4.x + t4.x, v.y * s4.y + t4.y, v.z * s4.z + t4.z, v.w * s4.w + t4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (act == DIFFSAL_ACT_RELU) o[k] = fmaxf(o[k], 0.f);
      else if (act == DIFFSAL_ACT_GELU_ERF) o[k] = gelu_erf(o[k]);
      else if (act == 4) o[k] = swishf(o[k]);
    }
    st4(out + m * C + c, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// dx = k1*dz - k2 - k3*xhat with per-(segment, channel) coefficients; dz as in rowstats (same modes 1..3)
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ y, const float* __restrict__ mu,
                                                             const float* __restrict__ rs, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ k1,
                                                             const float* __restrict__ k2, const float* __restrict__ k3,
                                                             float* __restrict__ dx, long M, int C, int seg_rows,
                                                             int mode) {
  const int c4n = C >> 2;
  const long total = M * c4n;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    const long m = i / c4n;
    const long so = (m / seg_rows) * C + c;
    const float4 xv = ld4(x + m * C + c), gv = ld4(dy + m * C + c);
    const float4 m4 = ld4(mu + so), r4 = ld4(rs + so), a4 = ld4(k1 + so), b4 = ld4(k2 + so), c4v = ld4(k3 + so);
    const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
    const float ma[4] = {m4.x, m4.y, m4.z, m4.w}, ra[4] = {r4.x, r4.y, r4.z, r4.w};
    const float ka[4] = {a4.x, a4.y, a4.z, a4.w}, kb[4] = {b4.x, b4.y, b4.z, b4.w}, kc[4] = {c4v.x, c4v.y, c4v.z, c4v.w};
    float ya[4] = {1, 1, 1, 1}, gm[4] = {1, 1, 1, 1}, bt[4] = {0, 0, 0, 0};
    if (mode == 1) { const float4 yv = ld4(y + m * C + c); ya[0] = yv.x; ya[1] = yv.y; ya[2] = yv.z; ya[3] = yv.w; }
    if (mode == 2) {
      const float4 g4 = ld4(gamma + c), be = ld4(beta + c);
      gm[0] = g4.x; gm[1] = g4.y; gm[2] = g4.z; gm[3] = g4.w; bt[0] = be.x; bt[1] = be.y; bt[2] = be.z; bt[3] = be.w;
    }
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (xa[k] - ma[k]) * ra[k];
      float dz = ga[k];
      if (mode == 1) dz = ya[k] > 0.f ? dz : 0.f;
      else if (mode == 2) dz *= swish_grad(xh * gm[k] + bt[k]);
      o[k] = ka[k] * dz - kb[k] - kc[k] * xh;
    }
    st4(dx + m * C + c, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// LayerNorm backward over C (rows in registers, statistics recomputed): dx, and per-block partial dgamma / dbeta.
template <int G, int NV>
__device__ __forceinline__ void layernorm_bwd_body(const float* __restrict__ x, const float* __restrict__ dy,
                                                   const float* __restrict__ gamma, const float* __restrict__ add,
                                                   float* __restrict__ dx, double* __restrict__ part, int M, int C, float eps,
                                                   double* shd) {
  constexpr int ROWS = 256 / G;
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  float4 dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { dg[i] = make_float4(0, 0, 0, 0); db[i] = make_float4(0, 0, 0, 0); }
  for (long row = static_cast<long>(blockIdx.x) * ROWS + gr; row < M; row += static_cast<long>(gridDim.x) * ROWS) {
    float4 xv[NV], gv[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (gl + i * G) * 4;
      xv[i] = c < C ? ld4(x + row * C + c) : make_float4(0, 0, 0, 0);
      gv[i] = c < C ? ld4(dy + row * C + c) : make_float4(0, 0, 0, 0);
      s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    }
    s = group_sum<G>(s);
    const float mean = s / static_cast<float>(C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (gl + i * G) * 4;
      if (c < C) {
        const float a = xv[i].x - mean, b = xv[i].y - mean, cc = xv[i].z - mean, d = xv[i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    q = group_sum<G>(q);
    const float rstd = 1.0f / sqrtf(q / static_cast<float>(C) + eps);
    float a1 = 0.f, a2 = 0.f;  // sum(dxhat), sum(dxhat * xhat)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (gl + i * G) * 4;
      if (c < C) {
        const float4 gm = ld4(gamma + c);
        const float xh[4] = {(xv[i].x - mean) * rstd, (xv[i].y - mean) * rstd, (xv[i].z - mean) * rstd, (xv[i].w - mean) * rstd};
        const float gy[4] = {gv[i].x, gv[i].y, gv[i].z, gv[i].w};
        const float gg[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1 += gy[k] * gg[k]; a2 += gy[k] * gg[k] * xh[k]; }
        dg[i].x += gy[0] * xh[0]; dg[i].y += gy[1] * xh[1]; dg[i].z += gy[2] * xh[2]; dg[i].w += gy[3] * xh[3];
        db[i].x += gy[0]; db[i].y += gy[1]; db[i].z += gy[2]; db[i].w += gy[3];
      }
    }
    a1 = group_sum<G>(a1) / static_cast<float>(C);
    a2 = group_sum<G>(a2) / static_cast<float>(C);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (gl + i * G) * 4;
      if (c < C) {
        const float4 gm = ld4(gamma + c);
        float4 o;
        o.x = rstd * (gv[i].x * gm.x - a1 - (xv[i].x - mean) * rstd * a2);
        o.y = rstd * (gv[i].y * gm.y - a1 - (xv[i].y - mean) * rstd * a2);
        o.z = rstd * (gv[i].z * gm.z - a1 - (xv[i].z - mean) * rstd * a2);
        o.w = rstd * (gv[i].w * gm.w - a1 - (xv[i].w - mean) * rstd * a2);
        if (add) {     // the gradient that reaches x on its other path (the residual connection around the normalised branch)
          const float4 e = ld4(add + row * C + c);
          o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
        }
        st4(dx + row * C + c, o);
      }
    }
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) shd[i] = 0.0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (gl + i * G) * 4;
    if (c < C) {
      const float dgv[4] = {dg[i].x, dg[i].y, dg[i].z, dg[i].w}, dbv[4] = {db[i].x, db[i].y, db[i].z, db[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(&shd[c + k], static_cast<double>(dgv[k]));
        atomicAdd(&shd[C + c + k], static_cast<double>(dbv[k]));
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) part[static_cast<long>(blockIdx.x) * 2 * C + i] = shd[i];
}

template <int G, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ add,
                                                            float* __restrict__ dx, double* __restrict__ part, int M, int C,
                                                            float eps) {
  extern __shared__ double shd[];  // [2][C]
  layernorm_bwd_body<G, NV>(x, dy, gamma, add, dx, part, M, C, eps, shd);
}

// up to three tensors of one width (blockIdx.y; see LnMulti in norm.hip): part[t][gridDim.x][2][C]; a workgroup without rows
// leaves zeros
struct LnBwdMulti {
  const float* x[3]; const float* dy[3]; const float* gamma[3]; float* dx[3];
  double* part;
  int M[3];
  float eps[3];
};

template <int G, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_multi_kernel(LnBwdMulti p, int C) {
  extern __shared__ double shd[];  // [2][C]
  const int t = blockIdx.y;
  layernorm_bwd_body<G, NV>(p.x[t], p.dy[t], p.gamma[t], nullptr, p.dx[t], p.part + static_cast<long>(t) * gridDim.x * 2 * C, p.M[t], C,
                            p.eps[t], shd);
}

// y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) from a counter-based hash of (seed, element index): the same call
// with the same seed is the backward (R/models/saliency_decoder/sal_unet.py:109,133, Dropout(0.1) in train mode).
__device__ __forceinline__ unsigned hash_u32(unsigned a, unsigned b) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, long n,
                                                      float p, unsigned seed_lo, unsigned seed_hi) {
  const float scale = 1.0f / (1.0f - p);
  const unsigned thr = static_cast<unsigned>(p * 4294967296.0);
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long>(gridDim.x) * 256) {
    const unsigned h = hash_u32(hash_u32(static_cast<unsigned>(i), seed_lo), static_cast<unsigned>(i >> 32) ^ seed_hi);
    out[i] = h >= thr ? x[i] * scale : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// Depthwise convolution on NHWC (training path of the q / k / v token projections, attention.py:36-76):
// forward, data gradient and weight gradient.  w: [taps][C].  thread = (output element, 4 channels).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ out, int N, int H, int W, int C, int Ho,
                                                         int Wo, int k, int stride, int pad) {
  const int c4n = C >> 2;
  const long total = static_cast<long>(N) * Ho * Wo * c4n;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    long pix = i / c4n;
    const int ox = static_cast<int>(pix % Wo); pix /= Wo;
    const int oy = static_cast<int>(pix % Ho);
    const int n = static_cast<int>(pix / Ho);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * stride - pad + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * stride - pad + kx;
        if (ix < 0 || ix >= W) continue;
        const float4 a = ld4(x + ((static_cast<long>(n) * H + iy) * W + ix) * C + c);
        const float4 ww = ld4(w + static_cast<long>(ky * k + kx) * C + c);
        acc.x = fmaf(a.x, ww.x, acc.x); acc.y = fmaf(a.y, ww.y, acc.y);
        acc.z = fmaf(a.z, ww.z, acc.z); acc.w = fmaf(a.w, ww.w, acc.w);
      }
    }
    st4(out + i * 4, acc);
  }
}

__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const float* __restrict__ du, const float* __restrict__ w,
                                                              float* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                                              int Wo, int k, int stride, int pad) {
  const int c4n = C >> 2;
  const long total = static_cast<long>(N) * H * W * c4n;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % c4n) * 4;
    long pix = i / c4n;
    const int ix = static_cast<int>(pix % W); pix /= W;
    const int iy = static_cast<int>(pix % H);
    const int n = static_cast<int>(pix / H);
    float4 acc = make_float4(0, 0, 0, 0);
    // taps with (iy + pad - ky) divisible by stride and the quotient inside the output
    for (int ky = (iy + pad) % stride; ky < k; ky += stride) {
      const int oy = (iy + pad - ky) / stride;
      if (iy + pad - ky < 0 || oy >= Ho) continue;
      for (int kx = (ix + pad) % stride; kx < k; kx += stride) {
        const int ox = (ix + pad - kx) / stride;
        if (ix + pad - kx < 0 || ox >= Wo) continue;
        const float4 g = ld4(du + ((static_cast<long>(n) * Ho + oy) * Wo + ox) * C + c);
        const float4 ww = ld4(w + static_cast<long>(ky * k + kx) * C + c);
        acc.x = fmaf(g.x, ww.x, acc.x); acc.y = fmaf(g.y, ww.y, acc.y);
        acc.z = fmaf(g.z, ww.z, acc.z); acc.w = fmaf(g.w, ww.w, acc.w);
      }
    }
    st4(dx + i * 4, acc);
  }
}

// part[tap][chunk][C]: one workgroup = one tap and one chunk of output pixels; threads = (pixel lane, 4 channels)
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ du,
                                                                double* __restrict__ part, int N, int H, int W, int C,
                                                                int Ho, int Wo, int k, int stride, int pad, int chunks) {
  extern __shared__ double shd[];  // [C]
  const int tap = blockIdx.y, chunk = blockIdx.x;
  const int ky = tap / k, kx = tap - ky * k;
  const int c4n = C >> 2;
  const int ppp = 256 / c4n > 0 ? 256 / c4n : 1;
  const int c4 = threadIdx.x % c4n, ps = threadIdx.x / c4n;
  const long P = static_cast<long>(N) * Ho * Wo;
  const long pb = P * chunk / chunks, pe = P * (chunk + 1) / chunks;
  float4 s = make_float4(0, 0, 0, 0);
  if (ps < ppp)
    for (long pix = pb + ps; pix < pe; pix += ppp) {
      long t = pix;
      const int ox = static_cast<int>(t % Wo); t /= Wo;
      const int oy = static_cast<int>(t % Ho);
      const int n = static_cast<int>(t / Ho);
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const float4 a = ld4(x + ((static_cast<long>(n) * H + iy) * W + ix) * C + c4 * 4);
      const float4 g = ld4(du + pix * C + c4 * 4);
      s.x = fmaf(a.x, g.x, s.x); s.y = fmaf(a.y, g.y, s.y); s.z = fmaf(a.z, g.z, s.z); s.w = fmaf(a.w, g.w, s.w);
    }
  for (int i = threadIdx.x; i < C; i += 256) shd[i] = 0.0;
  __syncthreads();
  if (ps < ppp) {
    atomicAdd(&shd[c4 * 4 + 0], static_cast<double>(s.x)); atomicAdd(&shd[c4 * 4 + 1], static_cast<double>(s.y));
    atomicAdd(&shd[c4 * 4 + 2], static_cast<double>(s.z)); atomicAdd(&shd[c4 * 4 + 3], static_cast<double>(s.w));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) part[(static_cast<long>(tap) * chunks + chunk) * C + i] = shd[i];
}

// ------------------------------------------------------------------------------------------------
// Attention backward (pooled K/V, Lk <= 32), pass 1 of 2: per query, recompute P = softmax(q K^T scale), form
// dS = P (dP - <P, dP>) scale with dP = dO V^T, write dq = dS K and store P and dS as [N*Lq][ld] rows
// (column = head * Lk + t).  Pass 2 is two segmented weight-gradient GEMMs on the matrix cores
// (wgrad.hip: dK = dS^T Q, dV = P^T dO, one segment per image) -- no atomics anywhere, so the result is
// deterministic.  Same lane mapping as the forward kernel.
// ------------------------------------------------------------------------------------------------
template <int LK, int G>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ dout,
                                                            float* __restrict__ dq, float* __restrict__ p_out,
                                                            float* __restrict__ ds_out, int Lq, int Lk, int C, int heads,
                                                            int ld, float scale) {
  // one workgroup = one image, one HEAD (blockIdx.z) and a run of queries: K | V of that head in LDS, [Lk][d] each
  extern __shared__ float sh[];
  const int d = C / heads, nf4 = d >> 2;
  const int hd = blockIdx.z;
  const int cb = hd * d;
  float* Ks = sh;
  float* Vs = sh + Lk * d;
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < Lk * nf4; i += 256) {
    const int t = i / nf4, j = (i - t * nf4) * 4;
    st4(Ks + t * d + j, ld4(k + (static_cast<long>(n) * Lk + t) * C + cb + j));
    st4(Vs + t * d + j, ld4(v + (static_cast<long>(n) * Lk + t) * C + cb + j));
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int QPW = 64 / G;
  const int g = lane % G;
  const int l = blockIdx.x * (4 * QPW) + wave * QPW + lane / G;
  const bool valid = l < Lq;
  const int lc = valid ? l : Lq - 1;
  const float* qr = q + (static_cast<long>(n) * Lq + lc) * C + cb;
  const float* gr = dout + (static_cast<long>(n) * Lq + lc) * C + cb;
  float sc[LK], dp[LK];
#pragma unroll
  for (int t = 0; t < LK; ++t) { sc[t] = 0.f; dp[t] = 0.f; }
  for (int i = g; i < nf4; i += G) {
    const float4 qv = ld4(qr + 4 * i), gv = ld4(gr + 4 * i);
#pragma unroll
    for (int t = 0; t < LK; ++t) {
      if (t < Lk) {
        const float4 kv = ld4(Ks + t * d + 4 * i), vv = ld4(Vs + t * d + 4 * i);
        sc[t] = fmaf(qv.x, kv.x, fmaf(qv.y, kv.y, fmaf(qv.z, kv.z, fmaf(qv.w, kv.w, sc[t]))));
        dp[t] = fmaf(gv.x, vv.x, fmaf(gv.y, vv.y, fmaf(gv.z, vv.z, fmaf(gv.w, vv.w, dp[t]))));
      }
    }
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < LK; ++t)
    if (t < Lk) { sc[t] = group_sum<G>(sc[t]) * scale; dp[t] = group_sum<G>(dp[t]); mx = fmaxf(mx, sc[t]); }
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < LK; ++t)
    if (t < Lk) { sc[t] = expf(sc[t] - mx); sum += sc[t]; }
  const float inv = 1.0f / sum;
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < LK; ++t)
    if (t < Lk) { sc[t] *= inv; dot += sc[t] * dp[t]; }       // sc = P
  float ds[LK];
#pragma unroll
  for (int t = 0; t < LK; ++t) ds[t] = t < Lk ? sc[t] * (dp[t] - dot) * scale : 0.f;   // dS * scale
  float* dqr = dq + (static_cast<long>(n) * Lq + lc) * C + cb;
  for (int i = g; i < nf4; i += G) {
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < LK; ++t) {
      if (t < Lk) {
        const float4 kv = ld4(Ks + t * d + 4 * i);
        acc.x = fmaf(ds[t], kv.x, acc.x); acc.y = fmaf(ds[t], kv.y, acc.y);
        acc.z = fmaf(ds[t], kv.z, acc.z); acc.w = fmaf(ds[t], kv.w, acc.w);
      }
    }
    if (valid) st4(dqr + 4 * i, acc);
  }
  // every lane of a query group holds the same P / dS after the butterfly reductions: lane t % G writes column t
  if (valid) {
    float* pr = p_out + (static_cast<long>(n) * Lq + l) * ld + hd * Lk;
    float* sr = ds_out + (static_cast<long>(n) * Lq + l) * ld + hd * Lk;
#pragma unroll
    for (int t = 0; t < LK; ++t)
      if (t < Lk && (t % G) == g) { pr[t] = sc[t]; sr[t] = ds[t]; }
  }
}

// ------------------------------------------------------------------------------------------------
// Adjoint of the bilinear resize (align_corners=False), gather form (deterministic, no atomics):
// dx[n, iy, ix, :] = sum over the outputs (Y, X) whose two taps per axis include (iy, ix) of wy * wx * dy[n, Y, X, :].
// Backward of nn.Upsample(x2) / F.interpolate (common_block.py:197, sal_unet.py:325-327,482-484).
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                         int h, int w, int H, int W, int C, float sy, float sx) {
  const int cv = C / VEC;
  const long total = static_cast<long>(N) * h * w * cv;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % cv) * VEC;
    long pix = i / cv;
    const int ix = static_cast<int>(pix % w); pix /= w;
    const int iy = static_cast<int>(pix % h);
    const int n = static_cast<int>(pix / h);
    const int Ylo = max(0, static_cast<int>(floorf((iy - 0.5f) / sy - 0.5f)) - 1);
    const int Yhi = min(H - 1, static_cast<int>(ceilf((iy + 1.5f) / sy - 0.5f)) + 1);
    const int Xlo = max(0, static_cast<int>(floorf((ix - 0.5f) / sx - 0.5f)) - 1);
    const int Xhi = min(W - 1, static_cast<int>(ceilf((ix + 1.5f) / sx - 0.5f)) + 1);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      int y0, y1; float ly;
      bilin_coord(Y, sy, h, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int X = Xlo; X <= Xhi; ++X) {
        int x0, x1; float lx;
        bilin_coord(X, sx, w, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        const float* g = dy + ((static_cast<long>(n) * H + Y) * W + X) * C + c;
        const float ww = wy * wx;
        if constexpr (VEC == 4) {
          const float4 v = ld4(g);
          acc[0] = fmaf(ww, v.x, acc[0]); acc[1] = fmaf(ww, v.y, acc[1]);
          acc[2] = fmaf(ww, v.z, acc[2]); acc[3] = fmaf(ww, v.w, acc[3]);
        } else {
          acc[0] = fmaf(ww, g[0], acc[0]);
        }
      }
    }
    float* o = dx + i * VEC;
    if constexpr (VEC == 4) st4(o, make_float4(acc[0], acc[1], acc[2], acc[3]));
    else o[0] = acc[0];
  }
}

// One axis of the same adjoint on [outer][L][inner] -> [outer][l][inner]: rows then columns costs ~2(f+1) taps per
// element and pass instead of (2f+3)^2 in the joint form above (f = up-factor), and the first pass shrinks the data by f.
template <int VEC>
__global__ __launch_bounds__(256) void resize_bwd_axis_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                              long outer, int L, int l, int inner, float scale) {
  const int cv = inner / VEC;
  const long total = outer * l * cv;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int c = static_cast<int>(i % cv) * VEC;
    long r = i / cv;
    const int il = static_cast<int>(r % l);
    const long o = r / l;
    const int lo = max(0, static_cast<int>(floorf((il - 0.5f) / scale - 0.5f)) - 1);
    const int hi = min(L - 1, static_cast<int>(ceilf((il + 1.5f) / scale - 0.5f)) + 1);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int Y = lo; Y <= hi; ++Y) {
      int y0, y1; float ly;
      bilin_coord(Y, scale, l, y0, y1, ly);
      const float wy = (y0 == il ? 1.f - ly : 0.f) + (y1 == il ? ly : 0.f);
      if (wy == 0.f) continue;
      const float* g = dy + (o * L + Y) * inner + c;
      if constexpr (VEC == 4) {
        const float4 v = ld4(g);
        acc[0] = fmaf(wy, v.x, acc[0]); acc[1] = fmaf(wy, v.y, acc[1]);
        acc[2] = fmaf(wy, v.z, acc[2]); acc[3] = fmaf(wy, v.w, acc[3]);
      } else {
        acc[0] = fmaf(wy, g[0], acc[0]);
      }
    }
    float* out = dx + i * VEC;
    if constexpr (VEC == 4) st4(out, make_float4(acc[0], acc[1], acc[2], acc[3]));
    else out[0] = acc[0];
  }
}

// frames [B][Tin*hw][C] (first Q = Tv*hw rows) -> NCTHW [B][C][Q]: backward of pack_frames for the visual features
__global__ __launch_bounds__(256) void unpack_frames_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                            int Q, long in_batch_stride, int tiles_q) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y;
  const int tq = blockIdx.x % tiles_q, tcx = blockIdx.x / tiles_q;
  const int q0 = tq * 64, c0 = tcx * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = in + static_cast<long>(b) * in_batch_stride;
  for (int r = ty; r < 64; r += 4) {  // r: q in tile, tx: channel (contiguous reads)
    const int q = q0 + r, c = c0 + tx;
    tile[r][tx] = (q < Q && c < C) ? src[static_cast<long>(q) * C + c] : 0.f;
  }
  __syncthreads();
  float* dst = out + static_cast<long>(b) * C * Q;
  for (int r = ty; r < 64; r += 4) {  // r: channel in tile, tx: q (contiguous writes)
    const int c = c0 + r, q = q0 + tx;
    if (c < C && q < Q) dst[static_cast<long>(c) * Q + q] = tile[tx][r];
  }
}

// head (1x1 conv C -> 1 + sigmoid) backward: dpre[m] = ds[m] * s[m] (1 - s[m]);
// dy[m, c] = dpre[m] w[c];  part[block][C+1] = (sum_m dpre[m] y[m, c], sum_m dpre[m])
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                       const float* __restrict__ s_out, const float* __restrict__ ds,
                                                       float* __restrict__ dy, double* __restrict__ part, long M, int C) {
  extern __shared__ double shd[];  // [C + 1]
  const int c4n = C >> 2;
  const int rpp = 256 / c4n > 0 ? 256 / c4n : 1;
  const int c4 = threadIdx.x % c4n, rs = threadIdx.x / c4n;
  float4 acc = make_float4(0, 0, 0, 0);
  float accb = 0.f;
  if (rs < rpp) {
    const float4 wv = ld4(w + c4 * 4);
    for (long m = static_cast<long>(blockIdx.x) * rpp + rs; m < M; m += static_cast<long>(gridDim.x) * rpp) {
      const float sv = s_out[m];
      const float dp = ds[m] * sv * (1.f - sv);
      const float4 yv = ld4(y + m * C + c4 * 4);
      st4(dy + m * C + c4 * 4, make_float4(dp * wv.x, dp * wv.y, dp * wv.z, dp * wv.w));
      acc.x = fmaf(dp, yv.x, acc.x); acc.y = fmaf(dp, yv.y, acc.y); acc.z = fmaf(dp, yv.z, acc.z); acc.w = fmaf(dp, yv.w, acc.w);
      if (c4 == 0) accb += dp;
    }
  }
  for (int i = threadIdx.x; i <= C; i += 256) shd[i] = 0.0;
  __syncthreads();
  if (rs < rpp) {
    atomicAdd(&shd[c4 * 4 + 0], static_cast<double>(acc.x)); atomicAdd(&shd[c4 * 4 + 1], static_cast<double>(acc.y));
    atomicAdd(&shd[c4 * 4 + 2], static_cast<double>(acc.z)); atomicAdd(&shd[c4 * 4 + 3], static_cast<double>(acc.w));
    if (c4 == 0) atomicAdd(&shd[C], static_cast<double>(accb));
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= C; i += 256) part[static_cast<long>(blockIdx.x) * (C + 1) + i] = shd[i];
}

// conv_in (1 -> C, 3x3, pad 1) parameter gradients: part[tap 0..8 | bias][chunk][C]
__global__ __launch_bounds__(256) void conv_in_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          double* __restrict__ part, int B, int H, int W, int C,
                                                          int chunks) {
  extern __shared__ double shd[];  // [C]
  const int tap = blockIdx.y, chunk = blockIdx.x;  // tap 9 = bias
  const int ky = tap / 3, kx = tap - ky * 3;
  const int c4n = C >> 2;
  const int ppp = 256 / c4n > 0 ? 256 / c4n : 1;
  const int c4 = threadIdx.x % c4n, ps = threadIdx.x / c4n;
  const long P = static_cast<long>(B) * H * W;
  const long pb = P * chunk / chunks, pe = P * (chunk + 1) / chunks;
  float4 s = make_float4(0, 0, 0, 0);
  if (ps < ppp)
    for (long pix = pb + ps; pix < pe; pix += ppp) {
      float xv = 1.f;
      if (tap < 9) {
        long t = pix;
        const int xw = static_cast<int>(t % W); t /= W;
        const int yh = static_cast<int>(t % H);
        const long n = t / H;
        const int iy = yh + ky - 1, ix = xw + kx - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        xv = x[(n * H + iy) * W + ix];
      }
      const float4 g = ld4(dy + pix * C + c4 * 4);
      s.x = fmaf(xv, g.x, s.x); s.y = fmaf(xv, g.y, s.y); s.z = fmaf(xv, g.z, s.z); s.w = fmaf(xv, g.w, s.w);
    }
  for (int i = threadIdx.x; i < C; i += 256) shd[i] = 0.0;
  __syncthreads();
  if (ps < ppp) {
    atomicAdd(&shd[c4 * 4 + 0], static_cast<double>(s.x)); atomicAdd(&shd[c4 * 4 + 1], static_cast<double>(s.y));
    atomicAdd(&shd[c4 * 4 + 2], static_cast<double>(s.z)); atomicAdd(&shd[c4 * 4 + 3], static_cast<double>(s.w));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) part[(static_cast<long>(tap) * chunks + chunk) * C + i] = shd[i];
}

// dense_small backward: out = W f(in) + b  (f = swish if swish_in).  One thread per output entry, tiny sizes.
//   dW[n,k] = sum_b dout[b,n] f(in[b,k]);  db[n] = sum_b dout[b,n];  din[b,k] = f'(in[b,k]) sum_n dout[b,n] W[n,k]
__global__ __launch_bounds__(256) void dense_small_bwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ dout, float* __restrict__ dw,
                                                              float* __restrict__ db, float* __restrict__ din, int B,
                                                              int K, int N, int swish_in) {
  const long nw = static_cast<long>(N) * K;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < nw + N; i += static_cast<long>(gridDim.x) * 256) {
    if (i < nw) {
      const int n = static_cast<int>(i / K), k = static_cast<int>(i - static_cast<long>(n) * K);
      float s = 0.f;
      for (int b = 0; b < B; ++b) {
        const float v = in[b * K + k];
        s = fmaf(dout[b * N + n], swish_in ? swishf(v) : v, s);
      }
      dw[i] = s;
    } else {
      const int n = static_cast<int>(i - nw);
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += dout[b * N + n];
      db[n] = s;
    }
  }
}

// din[b,k] = f'(in[b,k]) sum_n dout[b,n] W[n,k]: workgroup = (32 k, one b); 8 thread groups each take every 8th n
// (W rows are k-contiguous: coalesced), combined in a fixed order.  (A thread per output would walk N rows serially.)
__global__ __launch_bounds__(256) void dense_small_din_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ dout, float* __restrict__ din,
                                                              int K, int N, int swish_in) {
  __shared__ float sh[8][33];
  const int kl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int k = blockIdx.x * 32 + kl, b = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (k < K) {
    int n = g;
    for (; n + 24 < N; n += 32) {
      s0 = fmaf(dout[b * N + n], w[static_cast<long>(n) * K + k], s0);
      s1 = fmaf(dout[b * N + n + 8], w[static_cast<long>(n + 8) * K + k], s1);
      s2 = fmaf(dout[b * N + n + 16], w[static_cast<long>(n + 16) * K + k], s2);
      s3 = fmaf(dout[b * N + n + 24], w[static_cast<long>(n + 24) * K + k], s3);
    }
    for (; n < N; n += 8) s0 = fmaf(dout[b * N + n], w[static_cast<long>(n) * K + k], s0);
  }
  sh[g][kl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && k < K) {
    float t = sh[0][kl];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += sh[j][kl];
    const long o = static_cast<long>(b) * K + k;
    din[o] = swish_in ? t * swish_grad(in[o]) : t;
  }
}

// ------------------------------------------------------------------------------------------------
// Audio fusion backward (transformer.py:133-146).  Forward: a = nearest-up(a_small), m = mean_t a*x,
// s = softmax_x(m), out[b,c,t,y,x] = a*s (NCTHW).  One workgroup = (b, output row y, 32-channel slab):
//   ds = sum_t dout*a;  dm = s (ds - sum_x ds s);  dx = dm a / T;  da = dout s + dm x / T
// da of the `up` output rows / columns sharing one audio cell: columns are summed in registers by the one thread
// that owns the cell, rows go to part[(b,t,ys,xs)][dyr][c] and are added by diffsal_colsum (fixed order): no atomics.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void audio_fuse_bwd_kernel(const float* __restrict__ a_small, const float* __restrict__ x,
                                                             const float* __restrict__ dout, float* __restrict__ dx,
                                                             float* __restrict__ part, int T, int H, int W, int C,
                                                             int h, int w, int up) {
  extern __shared__ float sh[];  // s[32][W+1] | dm[32][W+1] | da[T][w][32]
  const int WP = W + 1;
  float* sS = sh;
  float* sD = sh + 32 * WP;
  float* sA = sh + 64 * WP;
  const int cslabs = C / 32;
  int bid = blockIdx.x;
  const int cs = bid % cslabs; bid /= cslabs;
  const int y = bid % H;
  const int b = bid / H;
  const int ys = y / up, dyr = y - ys * up;
  const int cl = threadIdx.x & 31, xl = threadIdx.x >> 5;
  const int c = cs * 32 + cl;
  const float invT = 1.0f / static_cast<float>(T);
  // A: m[c][x] (lanes = channels)
  for (int xx = xl; xx < W; xx += 8) {
    const int xs = xx / up;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
      const float av = a_small[((static_cast<long>(b) * T + t) * h * w + ys * w + xs) * C + c];
      acc = fmaf(av, x[(((static_cast<long>(b) * T + t) * H + y) * W + xx) * C + c], acc);
    }
    sS[cl * WP + xx] = acc * invT;
  }
  __syncthreads();
  // B: softmax over x per channel row (8 lanes per row)
  {
    const int row = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    float mx = -3.0e38f;
    for (int xx = l8; xx < W; xx += 8) mx = fmaxf(mx, sS[row * WP + xx]);
    mx = group_max<8>(mx);
    float sum = 0.f;
    for (int xx = l8; xx < W; xx += 8) { const float e = expf(sS[row * WP + xx] - mx); sS[row * WP + xx] = e; sum += e; }
    sum = group_sum<8>(sum);
    const float inv = 1.0f / sum;
    for (int xx = l8; xx < W; xx += 8) sS[row * WP + xx] *= inv;
  }
  __syncthreads();
  // C: thread = (channel, audio cell): ds[c][x] = sum_t dout*a for the cell's `up` columns, da = sum dout*s over them
  for (int i = threadIdx.x; i < 32 * w; i += 256) {
    const int xs = i % w, cc = i / w;         // lanes walk x: dout rows are x-contiguous
    const int cg = cs * 32 + cc;
    float dsu[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) dsu[u] = 0.f;
    for (int t = 0; t < T; ++t) {
      const float av = a_small[((static_cast<long>(b) * T + t) * h * w + ys * w + xs) * C + cg];
      const float* g = dout + (((static_cast<long>(b) * C + cg) * T + t) * H + y) * W + xs * up;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < up) {
          const float gv = g[u];
          dsu[u] = fmaf(gv, av, dsu[u]);
          acc = fmaf(gv, sS[cc * WP + xs * up + u], acc);
        }
      sA[(t * w + xs) * 32 + cc] = acc;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (u < up) sD[cc * WP + xs * up + u] = dsu[u];
  }
  __syncthreads();
  // D: dm = s (ds - sum_x ds s)
  {
    const int row = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    float dot = 0.f;
    for (int xx = l8; xx < W; xx += 8) dot += sD[row * WP + xx] * sS[row * WP + xx];
    dot = group_sum<8>(dot);
    for (int xx = l8; xx < W; xx += 8) sD[row * WP + xx] = sS[row * WP + xx] * (sD[row * WP + xx] - dot);
  }
  __syncthreads();
  // E: thread = (channel, audio cell), lanes = channels: dx = dm a / T ; da += sum over the cell's columns of dm x / T
  for (int i = threadIdx.x; i < 32 * w; i += 256) {
    const int c2l = i & 31, xs = i >> 5;
    const int c2 = cs * 32 + c2l;
    for (int t = 0; t < T; ++t) {
      const float av = a_small[((static_cast<long>(b) * T + t) * h * w + ys * w + xs) * C + c2];
      float acc = 0.f;
      for (int u = 0; u < up; ++u) {
        const int xx = xs * up + u;
        const float dm = sD[c2l * WP + xx] * invT;
        const long xo = (((static_cast<long>(b) * T + t) * H + y) * W + xx) * C + c2;
        dx[xo] = dm * av;
        acc = fmaf(dm, x[xo], acc);
      }
      sA[(t * w + xs) * 32 + c2l] += acc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * w * 32; i += 256) {
    const int cc = i & 31, r = i >> 5;
    const int xs = r % w, t = r / w;
    part[((((static_cast<long>(b) * T + t) * h + ys) * w + xs) * up + dyr) * C + cs * 32 + cc] = sA[i];
  }
}

#define DS_ROW_DISPATCH_B(C, CALL)                                \
  do {                                                            \
    const int c4 = (C) / 4;                                       \
    if (c4 <= 8) { CALL(8, 1); }                                  \
    else if (c4 <= 16) { CALL(16, 1); }                           \
    else if (c4 <= 32) { CALL(32, 1); }                           \
    else if (c4 <= 64) { CALL(64, 1); }                           \
    else if (c4 <= 128) { CALL(64, 2); }                          \
    else if (c4 <= 192) { CALL(64, 3); }                          \
    else if (c4 <= 256) { CALL(64, 4); }                          \
    else { set_error("channel count %d > 1024 unsupported", (C)); return DIFFSAL_E_SHAPE; } \
  } while (0)

}  // namespace diffsal

using namespace diffsal;

extern "C" int diffsal_act_bwd(const float* dy, const float* ref, float* dx, size_t n, int mode,
                               diffsal_stream_t stream) {
  DS_REQUIRE(dy && ref && dx, DIFFSAL_E_ARG, "act_bwd: null argument");
  DS_REQUIRE(n % 4 == 0 && mode >= 1 && mode <= 3, DIFFSAL_E_SHAPE, "act_bwd: n=%zu must be a multiple of 4, mode 1..3", n);
  DS_REQUIRE(aligned16(dy) && aligned16(ref) && aligned16(dx), DIFFSAL_E_ALIGN, "act_bwd: misaligned pointer");
  if (n == 0) return DIFFSAL_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid_b(static_cast<long>(n / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dy, ref, dx, static_cast<long>(n / 4), mode);
  return check_launch("act_bwd");
}

extern "C" int diffsal_rowstats_chunks(int M, int seg_rows) {
  const int segs = M / seg_rows;
  // ~1024 workgroups in flight (4 per CU) as long as a chunk keeps >= 64 rows
  int chunks = 1024 / (segs > 0 ? segs : 1);
  chunks = chunks < 1 ? 1 : (chunks > 1024 ? 1024 : chunks);
  while (chunks > 1 && seg_rows / chunks < 64) chunks >>= 1;
  return chunks;
}

extern "C" int diffsal_rowstats(const float* x, const float* dy, const float* y, const float* mu, const float* rs,
                                const float* gamma, const float* beta, double* part, int M, int C, int seg_rows,
                                int mode, int stat_per_seg, diffsal_stream_t stream) {
  DS_REQUIRE(x && part, DIFFSAL_E_ARG, "rowstats: null argument");
  DS_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && seg_rows > 0 && M % seg_rows == 0 && mode >= 0 && mode <= 3,
             DIFFSAL_E_SHAPE, "rowstats: bad shape M=%d C=%d seg_rows=%d mode=%d", M, C, seg_rows, mode);
  DS_REQUIRE(mode == 0 || (dy && mu && rs), DIFFSAL_E_ARG, "rowstats: backward modes need dy, mu, rs");
  DS_REQUIRE(mode != 1 || y, DIFFSAL_E_ARG, "rowstats: mode 1 needs y");
  DS_REQUIRE(mode != 2 || (gamma && beta), DIFFSAL_E_ARG, "rowstats: mode 2 needs gamma, beta");
  const int chunks = diffsal_rowstats_chunks(M, seg_rows);
  hipLaunchKernelGGL(rowstats_kernel, dim3(chunks, M / seg_rows), dim3(256), 2 * C * sizeof(double),
                     static_cast<hipStream_t>(stream), x, dy, y, mu, rs, gamma, beta, part, M, C, seg_rows, chunks, mode,
                     stat_per_seg);
  return check_launch("rowstats");
}

extern "C" int diffsal_reduce_partials(const double* part, void* out, int segs, int chunks, int width, int out_is_f64,
                                       diffsal_stream_t stream) {
  DS_REQUIRE(part && out, DIFFSAL_E_ARG, "reduce_partials: null argument");
  DS_REQUIRE(segs > 0 && chunks > 0 && width > 0, DIFFSAL_E_SHAPE, "reduce_partials: bad shape");
  const dim3 grid((width + 7) / 8, segs);
  if (out_is_f64)
    hipLaunchKernelGGL((reduce_partials_kernel<double>), grid, dim3(1024), 0, static_cast<hipStream_t>(stream), part,
                       static_cast<double*>(out), chunks, width);
  else
    hipLaunchKernelGGL((reduce_partials_kernel<float>), grid, dim3(1024), 0, static_cast<hipStream_t>(stream), part,
                       static_cast<float*>(out), chunks, width);
  return check_launch("reduce_partials");
}

extern "C" int diffsal_norm_finalize_fwd(const double* sums, const float* gamma, const float* beta, float* mu, float* rs,
                                         float* scale, float* shift, int segs, int C, int groups, double n, double eps,
                                         float* bn_mean, float* bn_var, float* running_mean, float* running_var,
                                         float momentum, float unbias, diffsal_stream_t stream) {
  DS_REQUIRE(sums && gamma && beta && mu && rs && scale && shift, DIFFSAL_E_ARG, "norm_finalize_fwd: null argument");
  DS_REQUIRE(segs > 0 && C > 0 && groups > 0 && C % groups == 0 && n > 0, DIFFSAL_E_SHAPE, "norm_finalize_fwd: bad shape");
  DS_REQUIRE((bn_mean == nullptr) == (bn_var == nullptr) && (running_mean == nullptr) == (running_var == nullptr),
             DIFFSAL_E_ARG, "norm_finalize_fwd: BatchNorm outputs come in pairs");
  DS_REQUIRE(bn_mean == nullptr || segs == 1, DIFFSAL_E_SHAPE, "norm_finalize_fwd: BatchNorm has one segment");
  hipLaunchKernelGGL(norm_finalize_fwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sums, gamma, beta, mu,
                     rs, scale, shift, segs, C, groups, n, eps, bn_mean, bn_var, running_mean, running_var, momentum, unbias);
  return check_launch("norm_finalize_fwd");
}

extern "C" int diffsal_norm_finalize_bwd(const double* t, const float* gamma, const float* rs, float* dgamma, float* dbeta,
                                         float* k1, float* k2, float* k3, int segs, int C, int groups, double n,
                                         diffsal_stream_t stream) {
  DS_REQUIRE(t && gamma && rs && dgamma && dbeta && k1 && k2 && k3, DIFFSAL_E_ARG, "norm_finalize_bwd: null argument");
  DS_REQUIRE(segs > 0 && C > 0 && groups > 0 && C % groups == 0 && n > 0, DIFFSAL_E_SHAPE, "norm_finalize_bwd: bad shape");
  hipLaunchKernelGGL(norm_finalize_bwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), t, gamma, rs, dgamma,
                     dbeta, k1, k2, k3, segs, C, groups, n);
  return check_launch("norm_finalize_bwd");
}

extern "C" int diffsal_affine_act(const float* x, const float* scale, const float* shift, float* out, int M, int C,
                                  int seg_rows, int act, diffsal_stream_t stream) {
  DS_REQUIRE(x && scale && shift && out, DIFFSAL_E_ARG, "affine_act: null argument");
  DS_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && seg_rows > 0, DIFFSAL_E_SHAPE, "affine_act: bad shape");
  hipLaunchKernelGGL(affine_act_kernel, dim3(ew_grid_b(static_cast<long>(M) * (C / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, scale, shift, out, static_cast<long>(M), C, seg_rows, act);
  return check_launch("affine_act");
}

extern "C" int diffsal_norm_bwd_apply(const float* x, const float* dy, const float* y, const float* mu, const float* rs,
                                      const float* gamma, const float* beta, const float* k1, const float* k2,
                                      const float* k3, float* dx, int M, int C, int seg_rows, int mode,
                                      diffsal_stream_t stream) {
  DS_REQUIRE(x && dy && mu && rs && k1 && k2 && k3 && dx, DIFFSAL_E_ARG, "norm_bwd_apply: null argument");
  DS_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && seg_rows > 0 && mode >= 1 && mode <= 3, DIFFSAL_E_SHAPE,
             "norm_bwd_apply: bad shape");
  DS_REQUIRE(mode != 1 || y, DIFFSAL_E_ARG, "norm_bwd_apply: mode 1 needs y");
  DS_REQUIRE(mode != 2 || (gamma && beta), DIFFSAL_E_ARG, "norm_bwd_apply: mode 2 needs gamma, beta");
  hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(ew_grid_b(static_cast<long>(M) * (C / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, dy, y, mu, rs, gamma, beta, k1, k2, k3, dx,
                     static_cast<long>(M), C, seg_rows, mode);
  return check_launch("norm_bwd_apply");
}

extern "C" int diffsal_layernorm_bwd_blocks(int M, int C) {
  const int c4 = C / 4;
  const int G = c4 <= 8 ? 8 : (c4 <= 16 ? 16 : (c4 <= 32 ? 32 : 64));
  long g = (static_cast<long>(M) + 256 / G - 1) / (256 / G);
  return static_cast<int>(g > 1024 ? 1024 : g);
}

extern "C" int diffsal_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* add, float* dx,
                                     double* part, int M, int C, float eps, diffsal_stream_t stream) {
  DS_REQUIRE(x && dy && gamma && dx && part, DIFFSAL_E_ARG, "layernorm_bwd: null argument");
  DS_REQUIRE(!add || aligned16(add), DIFFSAL_E_ALIGN, "layernorm_bwd: misaligned pointer");
  DS_REQUIRE(M > 0 && C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "layernorm_bwd: bad shape M=%d C=%d", M, C);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = diffsal_layernorm_bwd_blocks(M, C);
#define CALL(G, NV)                                                                                                  \
  hipLaunchKernelGGL((layernorm_bwd_kernel<G, NV>), dim3(blocks), dim3(256), 2 * C * sizeof(double), s, x, dy, gamma, add, \
                     dx, part, M, C, eps)
  DS_ROW_DISPATCH_B(C, CALL);
#undef CALL
  return check_launch("layernorm_bwd");
}

extern "C" int diffsal_layernorm_bwd_multi(const float* const* x, const float* const* dy, const float* const* gamma, float* const* dx,
                                           double* part, const int* M, int n, int C, const float* eps, diffsal_stream_t stream) {
  DS_REQUIRE(x && dy && gamma && dx && part && M && eps && n >= 1 && n <= 3, DIFFSAL_E_ARG,
             "layernorm_bwd_multi: null argument or n=%d not in 1..3", n);
  DS_REQUIRE(C > 0 && C % 4 == 0, DIFFSAL_E_SHAPE, "layernorm_bwd_multi: bad width C=%d", C);
  LnBwdMulti a{};
  int most = 0;
  for (int t = 0; t < n; ++t) {
    DS_REQUIRE(x[t] && dy[t] && gamma[t] && dx[t] && M[t] > 0, DIFFSAL_E_ARG, "layernorm_bwd_multi: null tensor %d", t);
    DS_REQUIRE(aligned16(x[t]) && aligned16(dy[t]) && aligned16(gamma[t]) && aligned16(dx[t]), DIFFSAL_E_ALIGN,
               "layernorm_bwd_multi: misaligned tensor %d", t);
    a.x[t] = x[t]; a.dy[t] = dy[t]; a.gamma[t] = gamma[t]; a.dx[t] = dx[t]; a.M[t] = M[t]; a.eps[t] = eps[t];
    most = M[t] > most ? M[t] : most;
  }
  a.part = part;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = diffsal_layernorm_bwd_blocks(most, C);      // part [n][blocks][2][C]
#define CALL(G, NV) \
  hipLaunchKernelGGL((layernorm_bwd_multi_kernel<G, NV>), dim3(blocks, n), dim3(256), 2 * C * sizeof(double), s, a, C)
  DS_ROW_DISPATCH_B(C, CALL);
#undef CALL
  return check_launch("layernorm_bwd_multi");
}

extern "C" int diffsal_dropout(const float* x, float* out, size_t n, float p, uint64_t seed, diffsal_stream_t stream) {
  DS_REQUIRE(x && out, DIFFSAL_E_ARG, "dropout: null argument");
  DS_REQUIRE(p >= 0.f && p < 1.f, DIFFSAL_E_SHAPE, "dropout: p=%f out of [0,1)", p);
  if (n == 0) return DIFFSAL_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid_b(static_cast<long>(n))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, out, static_cast<long>(n), p,
                     static_cast<unsigned>(seed & 0xFFFFFFFFu), static_cast<unsigned>(seed >> 32));
  return check_launch("dropout");
}

extern "C" int diffsal_dwconv(const float* x, const float* w, float* out, int N, int H, int W, int C, int k, int stride,
                              int pad, diffsal_stream_t stream) {
  DS_REQUIRE(x && w && out, DIFFSAL_E_ARG, "dwconv: null argument");
  DS_REQUIRE(N > 0 && C > 0 && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0, DIFFSAL_E_SHAPE, "dwconv: bad shape");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  DS_REQUIRE(Ho > 0 && Wo > 0, DIFFSAL_E_SHAPE, "dwconv: empty output");
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(ew_grid_b(static_cast<long>(N) * Ho * Wo * (C / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, w, out, N, H, W, C, Ho, Wo, k, stride, pad);
  return check_launch("dwconv");
}

extern "C" int diffsal_dwconv_bwd_data(const float* du, const float* w, float* dx, int N, int H, int W, int C, int k,
                                       int stride, int pad, diffsal_stream_t stream) {
  DS_REQUIRE(du && w && dx, DIFFSAL_E_ARG, "dwconv_bwd_data: null argument");
  DS_REQUIRE(N > 0 && C > 0 && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0, DIFFSAL_E_SHAPE, "dwconv_bwd_data: bad shape");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(ew_grid_b(static_cast<long>(N) * H * W * (C / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), du, w, dx, N, H, W, C, Ho, Wo, k, stride, pad);
  return check_launch("dwconv_bwd_data");
}

extern "C" int diffsal_dwconv_bwd_weight_chunks(int N, int H, int W, int k, int stride, int pad) {
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const long P = static_cast<long>(N) * Ho * Wo;
  long chunks = 2048 / (k * k);
  chunks = chunks < 1 ? 1 : chunks;
  while (chunks > 1 && P / chunks < 32) chunks >>= 1;
  return static_cast<int>(chunks > 256 ? 256 : chunks);
}

extern "C" int diffsal_dwconv_bwd_weight(const float* x, const float* du, double* part, int N, int H, int W, int C, int k,
                                         int stride, int pad, diffsal_stream_t stream) {
  DS_REQUIRE(x && du && part, DIFFSAL_E_ARG, "dwconv_bwd_weight: null argument");
  DS_REQUIRE(N > 0 && C > 0 && C % 4 == 0 && C <= 1024 && k > 0 && stride > 0 && pad >= 0, DIFFSAL_E_SHAPE,
             "dwconv_bwd_weight: bad shape (C <= 1024: one thread per 4 channels of a row)");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int chunks = diffsal_dwconv_bwd_weight_chunks(N, H, W, k, stride, pad);
  hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3(chunks, k * k), dim3(256), C * sizeof(double),
                     static_cast<hipStream_t>(stream), x, du, part, N, H, W, C, Ho, Wo, k, stride, pad, chunks);
  return check_launch("dwconv_bwd_weight");
}

template <int LK, int G>
static void launch_attention_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq,
                                 float* p_out, float* ds_out, int N, int Lq, int Lk, int C, int heads, int ld, float scale,
                                 int blocks, hipStream_t s) {
  const size_t lds = static_cast<size_t>(2) * Lk * (C / heads) * sizeof(float);
  if (lds > 64 * 1024) DS_RAISE_DYNAMIC_LDS((attention_bwd_kernel<LK, G>), 160 * 1024);
  hipLaunchKernelGGL((attention_bwd_kernel<LK, G>), dim3(blocks, N, heads), dim3(256), lds, s, q, k, v, dout, dq, p_out,
                     ds_out, Lq, Lk, C, heads, ld, scale);
}

static int attention_g(int C, int heads) {
  const int nf4 = C / heads / 4;
  int G = 1;
  while (G < 16 && nf4 / (2 * G) >= 3) G *= 2;
  return G == 2 ? 4 : G;
}

extern "C" int diffsal_attention_bwd_blocks(int Lq, int C, int heads) {
  const int qpb = 4 * (64 / attention_g(C, heads));  // one head per workgroup: all four waves take queries
  return (Lq + qpb - 1) / qpb;
}

extern "C" int diffsal_attention_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq,
                                     float* p_out, float* ds_out, int N, int Lq, int Lk, int C, int heads, int ld,
                                     float scale, diffsal_stream_t stream) {
  DS_REQUIRE(q && k && v && dout && dq && p_out && ds_out, DIFFSAL_E_ARG, "attention_bwd: null argument");
  DS_REQUIRE(ld >= heads * Lk, DIFFSAL_E_SHAPE, "attention_bwd: ld=%d < heads*Lk=%d", ld, heads * Lk);
  DS_REQUIRE(N > 0 && Lq > 0 && Lk > 0 && Lk <= 32 && (heads == 1 || heads == 2 || heads == 4) && C % heads == 0 &&
                 (C / heads) % 4 == 0,
             DIFFSAL_E_SHAPE, "attention_bwd: bad shape");
  DS_REQUIRE(static_cast<size_t>(2) * Lk * (C / heads) * sizeof(float) <= 160 * 1024, DIFFSAL_E_SHAPE,
             "attention_bwd: K/V of one head exceed LDS (Lk=%d d=%d)", Lk, C / heads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int G = attention_g(C, heads);
  const int blocks = diffsal_attention_bwd_blocks(Lq, C, heads);
#define ABW(L)                                                                                                        \
  switch (G) {                                                                                                        \
    case 16: launch_attention_bwd<L, 16>(q, k, v, dout, dq, p_out, ds_out, N, Lq, Lk, C, heads, ld, scale, blocks, s); break;      \
    case 8: launch_attention_bwd<L, 8>(q, k, v, dout, dq, p_out, ds_out, N, Lq, Lk, C, heads, ld, scale, blocks, s); break;        \
    case 4: launch_attention_bwd<L, 4>(q, k, v, dout, dq, p_out, ds_out, N, Lq, Lk, C, heads, ld, scale, blocks, s); break;        \
    default: launch_attention_bwd<L, 1>(q, k, v, dout, dq, p_out, ds_out, N, Lq, Lk, C, heads, ld, scale, blocks, s); break;       \
  }
  if (Lk <= 4) { ABW(4) } else if (Lk <= 8) { ABW(8) } else if (Lk <= 18) { ABW(18) } else { ABW(32) }
#undef ABW
  return check_launch("attention_bwd");
}

extern "C" int diffsal_resize_bilinear_bwd(const float* dy, float* dx, int N, int h, int w, int H, int W, int C,
                                           void* ws, size_t ws_bytes, diffsal_stream_t stream) {
  DS_REQUIRE(dy && dx, DIFFSAL_E_ARG, "resize_bilinear_bwd: null argument");
  DS_REQUIRE(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, DIFFSAL_E_SHAPE, "resize_bilinear_bwd: bad shape");
  const float sy = static_cast<float>(h) / static_cast<float>(H), sx = static_cast<float>(w) / static_cast<float>(W);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t tmp_bytes = static_cast<size_t>(N) * h * W * C * sizeof(float);
  if (C % 4 == 0 && aligned16(dy) && aligned16(dx) && ws && aligned16(ws) && ws_bytes >= tmp_bytes && (H != h || W != w)) {
    // separable: rows ([N][H][W*C] -> [N][h][W*C]) into ws, then columns ([N*h][W][C] -> [N*h][w][C])
    float* tmp = static_cast<float*>(ws);
    hipLaunchKernelGGL((resize_bwd_axis_kernel<4>), dim3(ew_grid_b(static_cast<long>(N) * h * W * (C / 4))), dim3(256), 0, s,
                       dy, tmp, static_cast<long>(N), H, h, W * C, sy);
    hipLaunchKernelGGL((resize_bwd_axis_kernel<4>), dim3(ew_grid_b(static_cast<long>(N) * h * w * (C / 4))), dim3(256), 0, s,
                       tmp, dx, static_cast<long>(N) * h, W, w, C, sx);
  } else if (C % 4 == 0 && aligned16(dy) && aligned16(dx))
    hipLaunchKernelGGL((resize_bwd_kernel<4>), dim3(ew_grid_b(static_cast<long>(N) * h * w * (C / 4))), dim3(256), 0, s,
                       dy, dx, N, h, w, H, W, C, sy, sx);
  else
    hipLaunchKernelGGL((resize_bwd_kernel<1>), dim3(ew_grid_b(static_cast<long>(N) * h * w * C)), dim3(256), 0, s, dy, dx,
                       N, h, w, H, W, C, sy, sx);
  return check_launch("resize_bilinear_bwd");
}

extern "C" int diffsal_unpack_frames(const float* frames, float* vis_grad, int B, int C, int Tv, int Tin, int hw,
                                     diffsal_stream_t stream) {
  DS_REQUIRE(frames && vis_grad, DIFFSAL_E_ARG, "unpack_frames: null argument");
  DS_REQUIRE(B > 0 && C > 0 && Tv > 0 && Tin >= Tv && hw > 0, DIFFSAL_E_SHAPE, "unpack_frames: bad shape");
  const int Q = Tv * hw, tiles_q = (Q + 63) / 64, tiles_c = (C + 63) / 64;
  hipLaunchKernelGGL(unpack_frames_kernel, dim3(tiles_q * tiles_c, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     frames, vis_grad, C, Q, static_cast<long>(Tin) * hw * C, tiles_q);
  return check_launch("unpack_frames");
}

extern "C" int diffsal_head_bwd(const float* y, const float* w, const float* s_out, const float* ds, float* dy,
                                double* part, int blocks, int M, int C, diffsal_stream_t stream) {
  DS_REQUIRE(y && w && s_out && ds && dy && part, DIFFSAL_E_ARG, "head_bwd: null argument");
  DS_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && blocks > 0, DIFFSAL_E_SHAPE, "head_bwd: bad shape");
  hipLaunchKernelGGL(head_bwd_kernel, dim3(blocks), dim3(256), (C + 1) * sizeof(double),
                     static_cast<hipStream_t>(stream), y, w, s_out, ds, dy, part, static_cast<long>(M), C);
  return check_launch("head_bwd");
}

extern "C" int diffsal_conv_in_bwd(const float* x, const float* dy, double* part, int B, int H, int W, int C, int chunks,
                                   diffsal_stream_t stream) {
  DS_REQUIRE(x && dy && part, DIFFSAL_E_ARG, "conv_in_bwd: null argument");
  DS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 1024 && chunks > 0, DIFFSAL_E_SHAPE,
             "conv_in_bwd: bad shape");
  hipLaunchKernelGGL(conv_in_bwd_kernel, dim3(chunks, 10), dim3(256), C * sizeof(double),
                     static_cast<hipStream_t>(stream), x, dy, part, B, H, W, C, chunks);
  return check_launch("conv_in_bwd");
}

extern "C" int diffsal_dense_small_bwd(const float* in, const float* w, const float* dout, float* dw, float* db,
                                       float* din, int B, int K, int N, int swish_in, diffsal_stream_t stream) {
  DS_REQUIRE(in && w && dout && dw && db && din, DIFFSAL_E_ARG, "dense_small_bwd: null argument");
  DS_REQUIRE(B > 0 && K > 0 && N > 0, DIFFSAL_E_SHAPE, "dense_small_bwd: bad shape");
  const long total = static_cast<long>(N) * K + N;
  hipLaunchKernelGGL(dense_small_bwd_kernel, dim3(ew_grid_b(total)), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     w, dout, dw, db, din, B, K, N, swish_in);
  hipLaunchKernelGGL(dense_small_din_kernel, dim3((K + 31) / 32, B), dim3(256), 0, static_cast<hipStream_t>(stream), in, w,
                     dout, din, K, N, swish_in);
  return check_launch("dense_small_bwd");
}

extern "C" int diffsal_audio_fuse_bwd(const float* a_small, const float* x, const float* dout, float* dx, float* part,
                                      int B, int T, int H, int W, int C, int h, int w, diffsal_stream_t stream) {
  DS_REQUIRE(a_small && x && dout && dx && part, DIFFSAL_E_ARG, "audio_fuse_bwd: null argument");
  DS_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0 && h > 0 && w > 0, DIFFSAL_E_SHAPE,
             "audio_fuse_bwd: bad shape");
  int up = 1;
  if (h != H && w != W) up = H / h;
  DS_REQUIRE(up >= 1 && up <= 8 && h * up == H && w * up == W, DIFFSAL_E_SHAPE,
             "audio_fuse_bwd: incompatible audio map (integer up-factor <= 8 expected)");
  const size_t lds = (static_cast<size_t>(64) * (W + 1) + static_cast<size_t>(T) * w * 32) * sizeof(float);
  DS_REQUIRE(lds <= 64 * 1024, DIFFSAL_E_SHAPE, "audio_fuse_bwd: row too wide for LDS (W=%d)", W);
  hipLaunchKernelGGL(audio_fuse_bwd_kernel, dim3(B * H * (C / 32)), dim3(256), lds, static_cast<hipStream_t>(stream),
                     a_small, x, dout, dx, part, T, H, W, C, h, w, up);
  return check_launch("audio_fuse_bwd");
}
