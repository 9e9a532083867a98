// MViTv2 pooling attention, head dimension 96 (every block of the tiny / small / base models): the three attention_pool
// convolutions of a block (R/models/mvit.py:446-494) in ONE launch, forward and data gradient, and the relative-position
// projections (mvit.py:363-410) forward and query gradient.
//
// Layout of all four kernels: a group of 8 lanes owns one token and each lane 12 of its 96 channels as three float4 at
// channel offsets 4*gl + 32*i -- a wavefront instruction then touches eight 128-byte runs instead of 64 scattered 16-byte
// pieces (the one-wavefront-per-token forms in mvit.hip were bound by L1 line requests: 87 us for 43 k queries of the
// projections, 3 x 44 us for the poolings of one stage-3 block).  The 27 filter taps sit in LDS with a 28th all-zero row:
// an absent tap selects that row and reads the class token (a valid address), so a kernel plane is nine independent loads
// with no branch.  Tap order and fmaf chains are those of pool3d_ln_kernel / pool3d_bwd_data_kernel: the convolution results
// are bit-identical to the per-tensor kernels.
#include "common.h"

namespace diffsal {

namespace {

constexpr int PD = 96;          // head dimension
constexpr int PQ = PD / 4;      // float4 per token row
constexpr int PROWS = 32;       // tokens per 256-thread pass

__device__ __forceinline__ int div_fast_(int a, int b, float inv_b) {   // exact a / b for 0 <= a < 2^23 (see mvit.hip)
  int q = static_cast<int>((static_cast<float>(a) + 0.5f) * inv_b);
  const int r = a - q * b;
  q += (r >= b) - (r < 0);
  return q;
}

// workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of the virtual order, so that the taps
// of neighbouring tokens hit the same L2
__device__ __forceinline__ int xcd_contiguous(int b, int nwg) {
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return xcd * q + (xcd < r ? xcd : r) + (b >> 3);
}

__device__ __forceinline__ float4 fma4(float4 a, float4 w, float4 c) {
  c.x = fmaf(a.x, w.x, c.x); c.y = fmaf(a.y, w.y, c.y); c.z = fmaf(a.z, w.z, c.z); c.w = fmaf(a.w, w.w, c.w);
  return c;
}
__device__ __forceinline__ float dot4(float4 a, float4 b, float s) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, s))));
}

}  // namespace

struct QkvPoolArgs {
  const void* qkv;         // forward / filter gradient: the fused projection [B][N][3][heads][96] (forward: fp32 or 16-bit storage)
  const float* w27[3];     // [27][96] per tensor (tap-major)
  const float* gamma[3];   // forward, LN form
  const float* beta[3];
  float* out[3];           // forward: [B*heads][1 + Lo][96]
  const float* dy[3];      // backward: gradients of out
  float* dqkv;             // backward: [B][N][3][heads][96], every element written
  float eps[3];
  int B, heads, T, H, W;
  int To[2], Ho[2], Wo[2], st[2], sh[2], sw[2];   // [0]: q, [1]: k and v
  int rows[2];             // forward: B*heads*(1 + Lo); backward: B*N*heads for both
  int blocks[2];           // workgroups per tensor
  int iters;               // 32-token passes per workgroup
  int w_channel_major;     // 1: filters (and their gradients) in the parameter's own [96][27] layout instead of tap-major [27][96]
  // run forms (below): per tensor the run length (0: the token-per-lane-group form above handles it), its workgroups and the
  // passes per workgroup
  int run_R[3], run_blocks[3], run_iters;
};

// the 27 x 96 filter of one tensor -> LDS as [tap][channel quad] (+ a 28th all-zero row), from either memory layout
__device__ __forceinline__ void stage_filter(float4* w_s, const float* __restrict__ w, int channel_major) {
  for (int i = threadIdx.x; i < 27 * PQ; i += 256) {
    if (channel_major) {
      const int tap = i / PQ, c = (i - tap * PQ) * 4;
      w_s[i] = make_float4(w[c * 27 + tap], w[(c + 1) * 27 + tap], w[(c + 2) * 27 + tap], w[(c + 3) * 27 + tap]);
    } else {
      w_s[i] = ld4(w + 4 * i);
    }
  }
  if (threadIdx.x < PQ) w_s[27 * PQ + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------------------------------
// forward: out_x[bh][n] = LN_x( depthwise Conv3d 3x3x3, pad 1, stride (st, sh, sw) of the x slice of qkv ), class token passed
// through the convolution.  LN = false: convolution only (training: the LayerNorm is its own differentiable operator).
// ------------------------------------------------------------------------------------------------------------------------
// TS: storage type of qkv (the 16-bit encoder path keeps its token GEMMs in bf16 / fp16); taps, LayerNorm and outputs are fp32.
template <bool LN, typename TS>
__global__ __launch_bounds__(256) void qkv_pool_kernel(QkvPoolArgs p) {
  __shared__ float4 w_s[28 * PQ];
  int blk = xcd_contiguous(blockIdx.x, gridDim.x), which = 0;
  if (blk >= p.blocks[0]) {
    blk -= p.blocks[0]; which = 1;
    if (blk >= p.blocks[1]) { blk -= p.blocks[1]; which = 2; }
  }
  const int g = which ? 1 : 0;
  stage_filter(w_s, p.w27[which], p.w_channel_major);
  __syncthreads();
  const int gl = threadIdx.x & 7, gr = threadIdx.x >> 3;
  const int T = p.T, H = p.H, W = p.W, To = p.To[g], Ho = p.Ho[g], Wo = p.Wo[g], st = p.st[g], sh = p.sh[g], sw = p.sw[g];
  const int Lo = To * Ho * Wo, rows = p.rows[g], heads = p.heads;
  const int tok_stride = 3 * heads * PD;
  const long clip_stride = static_cast<long>(1 + T * H * W) * tok_stride;
  const float inv_row = 1.0f / static_cast<float>(Lo + 1), inv_heads = 1.0f / static_cast<float>(heads);
  const float inv_wo = 1.0f / static_cast<float>(Wo), inv_ho = 1.0f / static_cast<float>(Ho);
  float* __restrict__ out = p.out[which];
  for (int pass = 0; pass < p.iters; ++pass) {
    const int row = (blk * p.iters + pass) * PROWS + gr;
    if (row - gr >= rows) break;
    const bool live = row < rows;
    const int rc = live ? row : rows - 1;
    const int bh = div_fast_(rc, Lo + 1, inv_row);
    const int n = rc - bh * (Lo + 1);
    const int b = div_fast_(bh, heads, inv_heads);
    const int head = bh - b * heads;
    const TS* base = static_cast<const TS*>(p.qkv) + b * clip_stride + (which * heads + head) * PD + gl * 4;
    float4 acc[3];
    if (n == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] = ld4(base + 32 * i);
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int l = n - 1;
      const int lw = div_fast_(l, Wo, inv_wo);
      const int wo = l - lw * Wo;
      const int to = div_fast_(lw, Ho, inv_ho);
      const int ho = lw - to * Ho;
      const int t0 = to * st - 1, y0 = ho * sh - 1, x0 = wo * sw - 1;
#pragma unroll 1
      for (int kt = 0; kt < 3; ++kt) {
        const int it = t0 + kt;
        const bool vt = it >= 0 && it < T;
        float4 a[9][3];
        int sel[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = y0 + ky;
          const bool vy = vt && iy >= 0 && iy < H;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int ix = x0 + kx;
            const bool v = vy && ix >= 0 && ix < W;
            const int tok = v ? 1 + (it * H + iy) * W + ix : 0;
            const TS* src = base + tok * tok_stride;
#pragma unroll
            for (int i = 0; i < 3; ++i) a[ky * 3 + kx][i] = ld4(src + 32 * i);
            sel[ky * 3 + kx] = v ? (kt * 9 + ky * 3 + kx) * PQ : 27 * PQ;
          }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i] = fma4(a[j][i], w_s[sel[j] + gl + 8 * i], acc[i]);
      }
    }
    if constexpr (LN) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) s += (acc[i].x + acc[i].y) + (acc[i].z + acc[i].w);
      const float mean = group_sum<8>(s) * (1.0f / PD);
      float qv = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        acc[i].x -= mean; acc[i].y -= mean; acc[i].z -= mean; acc[i].w -= mean;
        qv += (acc[i].x * acc[i].x + acc[i].y * acc[i].y) + (acc[i].z * acc[i].z + acc[i].w * acc[i].w);
      }
      const float rstd = 1.0f / sqrtf(group_sum<8>(qv) * (1.0f / PD) + p.eps[which]);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float4 ga = ld4(p.gamma[which] + gl * 4 + 32 * i), be = ld4(p.beta[which] + gl * 4 + 32 * i);
        acc[i].x = acc[i].x * rstd * ga.x + be.x; acc[i].y = acc[i].y * rstd * ga.y + be.y;
        acc[i].z = acc[i].z * rstd * ga.z + be.z; acc[i].w = acc[i].w * rstd * ga.w + be.w;
      }
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < 3; ++i) st4(out + static_cast<long>(rc) * PD + gl * 4 + 32 * i, acc[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// data gradient (gather form, no atomics): dqkv[b, n, x, head, :] = sum over the taps whose output position exists of
// dy_x[bh, out] * w_x[tap]; class token: copy.  NCS candidates per spatial axis as in pool3d_bwd_data_kernel (3 / 2 / 1 for
// stride 1 / 2 / >= 3), picked per tensor by a workgroup-uniform switch.
// ------------------------------------------------------------------------------------------------------------------------
template <int NCS>
__device__ __forceinline__ void pool_bwd_token(float4 (&acc)[3], const float4* w_s, const float* dyb, int gl, int it, int iy, int ix,
                                               int To, int Ho, int Wo, int s_sp) {
  const int k0y = (iy + 1) % s_sp, k0x = (ix + 1) % s_sp;
#pragma unroll 1
  for (int kt = 0; kt < 3; ++kt) {
    const int nt = it + 1 - kt;                                   // temporal stride 1
    const bool vt = nt >= 0 && nt < To;
    float4 a[NCS * NCS][3];
    int sel[NCS * NCS];
#pragma unroll
    for (int jy = 0; jy < NCS; ++jy) {
      const int ky = k0y + jy * s_sp, dy_ = iy + 1 - ky;
      const int ny = dy_ / s_sp;
      const bool vy = vt && ky <= 2 && dy_ >= 0 && ny < Ho;
#pragma unroll
      for (int jx = 0; jx < NCS; ++jx) {
        const int kx = k0x + jx * s_sp, dx_ = ix + 1 - kx;
        const int nx = dx_ / s_sp;
        const bool v = vy && kx <= 2 && dx_ >= 0 && nx < Wo;
        const int o = v ? 1 + (nt * Ho + ny) * Wo + nx : 0;
        const float* src = dyb + o * PD;
#pragma unroll
        for (int i = 0; i < 3; ++i) a[jy * NCS + jx][i] = ld4(src + 32 * i);
        sel[jy * NCS + jx] = v ? ((kt * 3 + ky) * 3 + kx) * PQ : 27 * PQ;
      }
    }
#pragma unroll
    for (int j = 0; j < NCS * NCS; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] = fma4(a[j][i], w_s[sel[j] + gl + 8 * i], acc[i]);
  }
}

__global__ __launch_bounds__(256) void qkv_pool_bwd_data_kernel(QkvPoolArgs p) {
  __shared__ float4 w_s[28 * PQ];
  int blk = xcd_contiguous(blockIdx.x, gridDim.x), which = 0;
  if (blk >= p.blocks[0]) {
    blk -= p.blocks[0]; which = 1;
    if (blk >= p.blocks[1]) { blk -= p.blocks[1]; which = 2; }
  }
  const int g = which ? 1 : 0;
  if (p.run_R[which]) return;                  // this tensor goes through the run form
  stage_filter(w_s, p.w27[which], p.w_channel_major);
  __syncthreads();
  const int gl = threadIdx.x & 7, gr = threadIdx.x >> 3;
  const int H = p.H, W = p.W, To = p.To[g], Ho = p.Ho[g], Wo = p.Wo[g], s_sp = p.sh[g];
  const int N = 1 + p.T * H * W, Lo = To * Ho * Wo, rows = p.rows[0], heads = p.heads;
  const int ncs = s_sp == 1 ? 3 : (s_sp == 2 ? 2 : 1);
  const float inv_heads = 1.0f / static_cast<float>(heads), inv_n = 1.0f / static_cast<float>(N);
  const float inv_w = 1.0f / static_cast<float>(W), inv_h = 1.0f / static_cast<float>(H);
  const float* __restrict__ dy = p.dy[which];
  for (int pass = 0; pass < p.iters; ++pass) {
    const int row = (blk * p.iters + pass) * PROWS + gr;       // (b, n, head)
    if (row - gr >= rows) break;
    const bool live = row < rows;
    const int rc = live ? row : rows - 1;
    const int bn = div_fast_(rc, heads, inv_heads);
    const int head = rc - bn * heads;
    const int b = div_fast_(bn, N, inv_n);
    const int n = bn - b * N;
    const float* dyb = dy + static_cast<long>(b * heads + head) * (Lo + 1) * PD + gl * 4;
    float4 acc[3];
    if (n == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] = ld4(dyb + 32 * i);
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int l = n - 1;
      const int lw = div_fast_(l, W, inv_w);
      const int ix = l - lw * W;
      const int it = div_fast_(lw, H, inv_h);
      const int iy = lw - it * H;
      if (ncs == 3) pool_bwd_token<3>(acc, w_s, dyb, gl, it, iy, ix, To, Ho, Wo, s_sp);
      else if (ncs == 2) pool_bwd_token<2>(acc, w_s, dyb, gl, it, iy, ix, To, Ho, Wo, s_sp);
      else pool_bwd_token<1>(acc, w_s, dyb, gl, it, iy, ix, To, Ho, Wo, s_sp);
    }
    if (live) {
      float* dst = p.dqkv + (static_cast<long>(bn) * 3 + which) * heads * PD + head * PD + gl * 4;
#pragma unroll
      for (int i = 0; i < 3; ++i) st4(dst + 32 * i, acc[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Run forms of the two gradient kernels (fp32, temporal stride 1, equal spatial strides 1 or 2, row length a
// multiple of 4).  The token-per-lane-group forms issue one 16-byte load per (token, tap, 4 channels): 27 x 24 requests per
// stride-1 token, and the texture addresser (64 B / clk / CU) is what bounds them (64 us for the 43 k tokens x 3 tensors of a
// stage-3 block: 8 % of the memory rate).  Here a thread owns a RUN of R consecutive tokens of one row and 4 channels: a kernel
// row's inputs are loaded once for the whole run ((R - 1) s + 3 loads for 3 R taps; R = 8, s = 1: 11.25 loads per token instead
// of 27) and a token's 24 channel quads sit in consecutive lanes (384 contiguous bytes).  Tap order and fmaf chains per output
// are those of the forms above: the data gradient is bit-identical to them; the filter gradient (thread = (token lane, kernel
// plane, quad) as above, a run per step) sums its outputs in another order.  The forward stays on the token form: it runs at
// 3 TB/s there and the run form measured the same or slower (tools/bench_pool.py).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int RUNS = 10;                       // data gradient: runs per 256-thread pass (240 threads = 10 runs x 24 channel quads)

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int R, int S>
__device__ __forceinline__ void pool_bwd_data_runs(const QkvPoolArgs& p, const float4* w_s, int which, int blk) {
  constexpr int NC = S == 1 ? R + 2 : R / 2 + 1;
  const int g = which ? 1 : 0;
  const int quad = threadIdx.x % PQ, rl = threadIdx.x / PQ;
  if (rl >= RUNS) return;
  const int T = p.T, H = p.H, W = p.W, To = p.To[g], Ho = p.Ho[g], Wo = p.Wo[g], heads = p.heads;
  const int N = 1 + T * H * W, Lo = To * Ho * Wo, rpr = W / R;
  const int n_video = p.B * heads * T * H * rpr, n_runs = n_video + p.B * heads;
  const float* __restrict__ dy = p.dy[which];
  for (int pass = 0; pass < p.run_iters; ++pass) {
    const int run = (blk * p.run_iters + pass) * RUNS + rl;
    if (run >= n_runs) return;
    if (run >= n_video) {
      const int bh = run - n_video, b = bh / heads, head = bh - b * heads;
      st4(p.dqkv + (static_cast<long>(b) * N * 3 + which) * heads * PD + head * PD + quad * 4,
          ld4(dy + static_cast<long>(bh) * (Lo + 1) * PD + quad * 4));
      continue;
    }
    int r = run;
    const int j = r % rpr; r /= rpr;
    const int iy = r % H; r /= H;
    const int it = r % T;
    const int bh = r / T, b = bh / heads, head = bh - b * heads;
    const float* dyb = dy + static_cast<long>(bh) * (Lo + 1) * PD + quad * 4;
    const int ix0 = j * R, nx_base = S == 1 ? ix0 - 1 : ix0 / 2;
    float4 acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = zero4();
#pragma unroll 1
    for (int kt = 0; kt < 3; ++kt) {
      const int nt = it + 1 - kt;
      if (nt < 0 || nt >= To) continue;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int d = iy + 1 - ky;
        if (d < 0 || (S == 2 && (d & 1))) continue;
        const int ny = S == 1 ? d : d >> 1;
        if (ny >= Ho) continue;
        const float* rowp = dyb + static_cast<long>(1 + (nt * Ho + ny) * Wo) * PD;
        float4 seg[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int nx = nx_base + c;
          seg[c] = ld4((nx >= 0 && nx < Wo) ? rowp + nx * PD : dyb);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int nx = nx_base + c;
          if (!(nx >= 0 && nx < Wo)) seg[c] = zero4();
        }
        const float4* wr = w_s + (kt * 3 + ky) * 3 * PQ + quad;
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            constexpr int dummy = 0; (void)dummy;
            const int e = i + 1 - kx;                                  // output column x stride = input column + 1 - kx
            if (S == 1) acc[i] = fma4(seg[e + 1], wr[kx * PQ], acc[i]);
            else if (e >= 0 && (e & 1) == 0) acc[i] = fma4(seg[e >> 1], wr[kx * PQ], acc[i]);
          }
      }
    }
    float* dst = p.dqkv + ((static_cast<long>(b) * N + 1 + (it * H + iy) * W + ix0) * 3 + which) * heads * PD + head * PD + quad * 4;
#pragma unroll
    for (int i = 0; i < R; ++i) st4(dst + static_cast<long>(i) * 3 * heads * PD, acc[i]);
  }
}

template <int R, int S>
__device__ __forceinline__ void pool_bwd_weight_runs(const QkvPoolArgs& p, int which, float4 (&acc)[9], int kt, int tl, int quad, int lo,
                                                     int hi) {
  constexpr int NC = (R - 1) * S + 3;
  const int g = which ? 1 : 0;
  const int T = p.T, H = p.H, W = p.W, To = p.To[g], Ho = p.Ho[g], Wo = p.Wo[g], heads = p.heads;
  const int Lo = To * Ho * Wo, rpr = Wo / R;
  const int tok_stride = 3 * heads * PD;
  const long clip_stride = static_cast<long>(1 + T * H * W) * tok_stride;
  const float* __restrict__ qkv = static_cast<const float*>(p.qkv);
  const float* __restrict__ dy = p.dy[which];
  for (int run = lo + tl; run < hi; run += 3) {
    int r = run;
    const int j = r % rpr; r /= rpr;
    const int ho = r % Ho; r /= Ho;
    const int to = r % To;
    const int bh = r / To, b = bh / heads, head = bh - b * heads;
    const int it = to - 1 + kt;
    if (it < 0 || it >= T) continue;
    const float* base = qkv + b * clip_stride + (which * heads + head) * PD + quad * 4;
    const int wo0 = j * R, x0 = wo0 * S - 1, y0 = ho * S - 1;
    const float* dyp = dy + (static_cast<long>(bh) * (Lo + 1) + 1 + (to * Ho + ho) * Wo + wo0) * PD + quad * 4;
    float4 gy[R];
#pragma unroll
    for (int i = 0; i < R; ++i) gy[i] = ld4(dyp + i * PD);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y0 + ky;
      if (iy < 0 || iy >= H) continue;
      const float* rowp = base + static_cast<long>(1 + (it * H + iy) * W) * tok_stride;
      float4 seg[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ix = x0 + c;
        seg[c] = ld4((ix >= 0 && ix < W) ? rowp + ix * tok_stride : base);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ix = x0 + c;
        if (!(ix >= 0 && ix < W)) seg[c] = zero4();
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < R; ++i) acc[ky * 3 + kx] = fma4(seg[i * S + kx], gy[i], acc[ky * 3 + kx]);
    }
  }
}

__device__ __forceinline__ int run_which(const QkvPoolArgs& p, int& blk) {
  int which = 0;
  if (blk >= p.run_blocks[0]) {
    blk -= p.run_blocks[0]; which = 1;
    if (blk >= p.run_blocks[1]) { blk -= p.run_blocks[1]; which = 2; }
  }
  return which;
}

__global__ __launch_bounds__(256) void qkv_pool_bwd_data_runs_kernel(QkvPoolArgs p) {
  __shared__ float4 w_s[28 * PQ];
  int blk = xcd_contiguous(blockIdx.x, gridDim.x);
  const int which = run_which(p, blk), g = which ? 1 : 0;
  stage_filter(w_s, p.w27[which], p.w_channel_major);
  __syncthreads();
  const int R = p.run_R[which], S = p.sh[g];
  if (S == 1 && R == 8) pool_bwd_data_runs<8, 1>(p, w_s, which, blk);
  else if (S == 1) pool_bwd_data_runs<4, 1>(p, w_s, which, blk);
  else if (R == 8) pool_bwd_data_runs<8, 2>(p, w_s, which, blk);
  else pool_bwd_data_runs<4, 2>(p, w_s, which, blk);
}

// ------------------------------------------------------------------------------------------------------------------------
// filter gradient of the three poolings in one launch: dw_x[tap][c] = sum over output tokens of dy_x[out][c] * in_x[tap of out][c].
// blockIdx.y = tensor, blockIdx.x = chunk of its output tokens; thread = (token lane 0..2, kernel plane kt, channel quad), nine
// taps each; the three token lanes are combined in a fixed order in LDS (double) and the chunk sums go to part[x][chunk][27*96]
// for diffsal_reduce_partials(segs = 3).  Same products and per-thread order as pool3d_bwd_weight_kernel; 512 chunks per
// tensor instead of 1024 per launch (a quarter of the partial-sum traffic for the three tensors).
// ------------------------------------------------------------------------------------------------------------------------
// chunks of output tokens per tensor (gridDim.x): 512, or 170 (two workgroups per CU over the three tensors, a third of the
// partial-sum traffic) when the query tensor has <= 6000 runs -- a thread then still walks <= 12 of them; measured per stage in
// tools/bench_pool.py (stage 3: 47 -> 42 us, stage 4: 44 -> 36 us; stage 1 / 2 lose with 170)
static int qkv_wchunks(long q_runs) { return q_runs > 6000 ? 512 : 170; }

__global__ __launch_bounds__(256) void qkv_pool_bwd_weight_kernel(QkvPoolArgs p, double* __restrict__ part) {
  __shared__ double shw[27 * PD];
  const int which = blockIdx.y, g = which ? 1 : 0;
  const int grp = threadIdx.x / PQ, c = (threadIdx.x % PQ) * 4;
  constexpr int TL = 3;
  const int tl = grp / 3, kt = grp - tl * 3;
  const bool live = tl < TL;
  const int T = p.T, H = p.H, W = p.W, To = p.To[g], Ho = p.Ho[g], Wo = p.Wo[g], st = p.st[g], sh = p.sh[g], sw = p.sw[g];
  const int Lo = To * Ho * Wo, heads = p.heads;
  const int rows = p.B * heads * Lo;                                   // video tokens of the output
  const int tok_stride = 3 * heads * PD;
  const long clip_stride = static_cast<long>(1 + T * H * W) * tok_stride;
  const int lo = static_cast<int>(static_cast<long>(rows) * blockIdx.x / gridDim.x);
  const int hi = static_cast<int>(static_cast<long>(rows) * (blockIdx.x + 1) / gridDim.x);
  const float inv_lo = 1.0f / static_cast<float>(Lo), inv_heads = 1.0f / static_cast<float>(heads);
  const float inv_wo = 1.0f / static_cast<float>(Wo), inv_ho = 1.0f / static_cast<float>(Ho);
  const float* __restrict__ dy = p.dy[which];
  float4 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && p.run_R[which]) {                  // run form: a run of R outputs of one row per step, its inputs loaded once per kernel row
    const int R = p.run_R[which];
    const int n_video = p.B * heads * To * Ho * (Wo / R);
    const int rlo = static_cast<int>(static_cast<long>(n_video) * blockIdx.x / gridDim.x);
    const int rhi = static_cast<int>(static_cast<long>(n_video) * (blockIdx.x + 1) / gridDim.x);
    if (sh == 1 && R == 8) pool_bwd_weight_runs<8, 1>(p, which, acc, kt, tl, c / 4, rlo, rhi);
    else if (sh == 1) pool_bwd_weight_runs<4, 1>(p, which, acc, kt, tl, c / 4, rlo, rhi);
    else pool_bwd_weight_runs<4, 2>(p, which, acc, kt, tl, c / 4, rlo, rhi);
  } else if (live) {
    for (int r = lo + tl; r < hi; r += TL) {
      const int bh = div_fast_(r, Lo, inv_lo);
      const int l = r - bh * Lo;
      const int b = div_fast_(bh, heads, inv_heads);
      const int head = bh - b * heads;
      const int lw = div_fast_(l, Wo, inv_wo);
      const int wo = l - lw * Wo;
      const int to = div_fast_(lw, Ho, inv_ho);
      const int ho = lw - to * Ho;
      const float4 gy = ld4(dy + (static_cast<long>(bh) * (Lo + 1) + 1 + l) * PD + c);
      const float* base = static_cast<const float*>(p.qkv) + b * clip_stride + (which * heads + head) * PD + c;
      const int it = to * st - 1 + kt;
      const bool vt = it >= 0 && it < T;
      float4 a[9];
      unsigned present = 0;      // an absent tap reads the class token (finite) and is multiplied by a zeroed gradient: the
#pragma unroll                   // selects sit on gy, not on the loaded values, so the nine loads stay in flight together
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = ho * sh - 1 + ky;
        const bool vy = vt && iy >= 0 && iy < H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = wo * sw - 1 + kx;
          const bool v = vy && ix >= 0 && ix < W;
          a[ky * 3 + kx] = ld4(base + (v ? 1 + (it * H + iy) * W + ix : 0) * tok_stride);
          present |= v ? 1u << (ky * 3 + kx) : 0u;
        }
      }
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const bool v = (present >> j) & 1u;
        const float4 gj = make_float4(v ? gy.x : 0.f, v ? gy.y : 0.f, v ? gy.z : 0.f, v ? gy.w : 0.f);
        acc[j] = fma4(a[j], gj, acc[j]);
      }
    }
  }
  for (int i = threadIdx.x; i < 27 * PD; i += 256) shw[i] = 0.0;
  __syncthreads();
  for (int turn = 0; turn < TL; ++turn) {          // fixed order over the token lanes: deterministic
    if (live && tl == turn) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        double* d = shw + (kt * 9 + i) * PD + c;
        d[0] += acc[i].x; d[1] += acc[i].y; d[2] += acc[i].z; d[3] += acc[i].w;
      }
    }
    __syncthreads();
  }
  double* o = part + (static_cast<long>(which) * gridDim.x + blockIdx.x) * 27 * PD;
  for (int i = threadIdx.x; i < 27 * PD; i += 256) {
    const int tap = i / PD, ch = i - tap * PD;
    o[p.w_channel_major ? ch * 27 + tap : i] = shw[i];
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Relative-position projections, slots as in mvit.hip (t: [0, 8), h: [8, 24), w: [24, 48) of the 48 columns per query):
//   forward   extra[row][slot0 + j] = q[row] . R_axis[coordinate][j]
//   backward  dq[row] (+)= sum_e dextra[row][e] * R_e
// ------------------------------------------------------------------------------------------------------------------------
constexpr int RT0 = 0, RH0 = 8;     // column layouts (mvit.hip): w columns start at 24 (E = 48) or 16 (E = 32)

__global__ __launch_bounds__(256) void relpos_project96_kernel(const float* __restrict__ q, const float* __restrict__ Rt,
                                                               const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                               float* __restrict__ extra, int qt, int qh, int qw, int kt, int kh,
                                                               int kw, int rows, int RE) {
  const int RW0 = RE == 32 ? 16 : 24;
  const int gl = threadIdx.x & 7;
  const int row = blockIdx.x * PROWS + (threadIdx.x >> 3);
  const bool live = row < rows;
  const int rc = live ? row : rows - 1;
  const int L = qt * qh * qw;
  const int n = rc % (L + 1);
  const int l = n > 0 ? n - 1 : 0;
  const int x = l % qw, y = (l / qw) % qh, t = l / (qw * qh);
  float4 qv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) qv[i] = ld4(q + static_cast<long>(rc) * PD + gl * 4 + 32 * i);
  float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // lane gl keeps columns e = 8*i + gl
  auto axis = [&](const float* R, int kk, int slot0) {   // R: this query's [kk][96] rows
    for (int j0 = 0; j0 < kk; j0 += 4) {
      float s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u < kk ? j0 + u : kk - 1;
        const float* r = R + j * PD + gl * 4;
        s[u] = dot4(qv[2], ld4(r + 64), dot4(qv[1], ld4(r + 32), dot4(qv[0], ld4(r), 0.f)));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v = group_sum<8>(s[u]);
        const int e = slot0 + j0 + u;
        const bool mine = j0 + u < kk && (e & 7) == gl;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = (mine && (e >> 3) == i) ? v : o[i];
      }
    }
  };
  axis(Rt + static_cast<long>(t) * kt * PD, kt, RT0);
  axis(Rh + static_cast<long>(y) * kh * PD, kh, RH0);
  axis(Rw + static_cast<long>(x) * kw * PD, kw, RW0);
  if (live) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (8 * i < RE) extra[static_cast<long>(rc) * RE + 8 * i + gl] = n > 0 ? o[i] : 0.f;
  }
}

__global__ __launch_bounds__(256) void relpos_bwd_q96_kernel(const float* __restrict__ dE, const float* __restrict__ Rt,
                                                             const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                             float* __restrict__ dq, int qt, int qh, int qw, int kt, int kh,
                                                             int kw, int accumulate, int rows, int RE) {
  const int RW0 = RE == 32 ? 16 : 24;
  const int gl = threadIdx.x & 7;
  const int row = blockIdx.x * PROWS + (threadIdx.x >> 3);
  const bool live = row < rows;
  const int rc = live ? row : rows - 1;
  const int L = qt * qh * qw;
  const int n = rc % (L + 1);
  const int l = n > 0 ? n - 1 : 0;
  const int x = l % qw, y = (l / qw) % qh, t = l / (qw * qh);
  const float* e = dE + static_cast<long>(rc) * RE;
  float4 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto axis = [&](const float* R, int kk, int slot0) {
    for (int j0 = 0; j0 < kk; j0 += 4) {
      const float4 e4 = ld4(e + slot0 + j0);
      const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
      float4 r[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u < kk ? j0 + u : kk - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) r[u][i] = ld4(R + j * PD + gl * 4 + 32 * i);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float c = j0 + u < kk ? ev[u] : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          acc[i].x = fmaf(c, r[u][i].x, acc[i].x); acc[i].y = fmaf(c, r[u][i].y, acc[i].y);
          acc[i].z = fmaf(c, r[u][i].z, acc[i].z); acc[i].w = fmaf(c, r[u][i].w, acc[i].w);
        }
      }
    }
  };
  axis(Rt + static_cast<long>(t) * kt * PD, kt, RT0);
  axis(Rh + static_cast<long>(y) * kh * PD, kh, RH0);
  axis(Rw + static_cast<long>(x) * kw * PD, kw, RW0);
  if (!live) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float* d = dq + static_cast<long>(rc) * PD + gl * 4 + 32 * i;
    float4 v = n > 0 ? acc[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (accumulate) {
      const float4 old = ld4(d);
      v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
    }
    st4(d, v);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The gathered relative-position tables of one block (resize_decomposed_rel_pos, mvit.py:330-361: linear resample of the
// learnt [len][D] table to 2*max(q,k)-1 rows, then a gather with the (query, key) index grid) as ONE sparse row map per
// table: out[m] = w[m][0] * rel[idx[m][0]] + w[m][1] * rel[idx[m][1]]  (m = query coordinate * k + key coordinate; the map
// depends on the grid geometry only and is built once on the host).  Backward: the transposed map in CSR form,
// drel[r] = sum over its entries of w * dout[m], fixed order, no atomics.  Three tables (t, h, w) per launch.
// ------------------------------------------------------------------------------------------------------------------------
struct RelTablesArgs {
  const float* src[3];   // forward: rel [R][D]; backward: dout [M][D]
  float* dst[3];         // forward: out [M][D]; backward: drel [R][D]
  const int* idx[3];     // forward: [M][2]; backward: CSR columns [nnz]
  const float* w[3];     // forward: [M][2]; backward: CSR weights [nnz]
  const int* ptr[3];     // backward: CSR row starts [R + 1]
  int rows[3];           // forward: M; backward: R
  int D4;                // D / 4
};

__global__ __launch_bounds__(256) void rel_tables_fwd_kernel(RelTablesArgs p) {
  const int t = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.rows[t] * p.D4) return;
  const int m = i / p.D4, c = (i - m * p.D4) * 4, D = p.D4 * 4;
  const int i0 = p.idx[t][2 * m], i1 = p.idx[t][2 * m + 1];
  const float w0 = p.w[t][2 * m], w1 = p.w[t][2 * m + 1];
  const float4 a = ld4(p.src[t] + static_cast<long>(i0) * D + c), b = ld4(p.src[t] + static_cast<long>(i1) * D + c);
  st4(p.dst[t] + static_cast<long>(m) * D + c, make_float4(w0 * a.x + w1 * b.x, w0 * a.y + w1 * b.y, w0 * a.z + w1 * b.z,
                                                            w0 * a.w + w1 * b.w));
}

__global__ __launch_bounds__(256) void rel_tables_bwd_kernel(RelTablesArgs p) {
  const int t = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.rows[t] * p.D4) return;
  const int r = i / p.D4, c = (i - r * p.D4) * 4, D = p.D4 * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int e = p.ptr[t][r]; e < p.ptr[t][r + 1]; ++e) {
    const float wv = p.w[t][e];
    const float4 g = ld4(p.src[t] + static_cast<long>(p.idx[t][e]) * D + c);
    acc.x = fmaf(wv, g.x, acc.x); acc.y = fmaf(wv, g.y, acc.y); acc.z = fmaf(wv, g.z, acc.z); acc.w = fmaf(wv, g.w, acc.w);
  }
  st4(p.dst[t] + static_cast<long>(r) * D + c, acc);
}

// 32-token passes per workgroup: enough workgroups to fill the chip eight times over, then longer workgroups (the 10 KB
// filter stage is paid once per workgroup)
static int passes_for(long rows_total) {
  const long per = rows_total / (PROWS * 2048L);
  return per < 1 ? 1 : (per > 8 ? 8 : static_cast<int>(per));
}

}  // namespace diffsal

using namespace diffsal;

static int fill_geometry(QkvPoolArgs& a, int B, int heads, int T, int H, int W, const int* stride_q, const int* stride_kv,
                         const char* who) {
  DS_REQUIRE(B > 0 && heads > 0 && T > 0 && H > 0 && W > 0, DIFFSAL_E_SHAPE, "%s: bad shape", who);
  for (int g = 0; g < 2; ++g) {
    const int* s = g ? stride_kv : stride_q;
    DS_REQUIRE(s[0] > 0 && s[1] > 0 && s[2] > 0, DIFFSAL_E_SHAPE, "%s: bad stride", who);
    a.st[g] = s[0]; a.sh[g] = s[1]; a.sw[g] = s[2];
    a.To[g] = (T - 1) / s[0] + 1; a.Ho[g] = (H - 1) / s[1] + 1; a.Wo[g] = (W - 1) / s[2] + 1;
  }
  a.B = B; a.heads = heads; a.T = T; a.H = H; a.W = W;
  return DIFFSAL_OK;
}

// Run-form plan of one launch: axis_len(g) = the row length the runs walk (outputs for the forward / filter gradient, inputs
// for the data gradient); r8_s2: run length 8 at stride 2 (data gradient only).  -> number of run-form workgroups.
static unsigned plan_runs(QkvPoolArgs& a, bool allowed, bool data, long* total_runs) {
  long total = 0;
  long n_runs[3] = {0, 0, 0};
  for (int x = 0; x < 3; ++x) {
    const int g = x ? 1 : 0;
    a.run_R[x] = 0;
    if (!allowed || a.st[g] != 1 || a.sh[g] != a.sw[g] || a.sh[g] > 2) continue;
    const int len = data ? a.W : a.Wo[g];
    int R = 0;
    if (len % 8 == 0 && (a.sh[g] == 1 || data)) R = 8;
    else if (len % 4 == 0) R = 4;
    if (!R) continue;
    a.run_R[x] = R;
    n_runs[x] = static_cast<long>(a.B) * a.heads * (data ? static_cast<long>(a.T) * a.H : static_cast<long>(a.To[g]) * a.Ho[g]) * (len / R) +
                static_cast<long>(a.B) * a.heads;
    total += n_runs[x];
  }
  const long per = total / (RUNS * 2048L);
  a.run_iters = per < 1 ? 1 : (per > 8 ? 8 : static_cast<int>(per));
  unsigned grid = 0;
  for (int x = 0; x < 3; ++x) {
    a.run_blocks[x] = static_cast<int>((n_runs[x] + static_cast<long>(RUNS) * a.run_iters - 1) / (static_cast<long>(RUNS) * a.run_iters));
    grid += static_cast<unsigned>(a.run_blocks[x]);
  }
  if (total_runs) *total_runs = total;
  return grid;
}

extern "C" int diffsal_qkv_pool(const void* qkv, const float* const* w27, const float* const* gamma, const float* const* beta,
                                const float* eps, float* const* out, int B, int heads, int D, int T, int H, int W,
                                const int* stride_q, const int* stride_kv, int dtype, int w_channel_major,
                                diffsal_stream_t stream) {
  DS_REQUIRE(qkv && w27 && out && stride_q && stride_kv && w27[0] && w27[1] && w27[2] && out[0] && out[1] && out[2],
             DIFFSAL_E_ARG, "qkv_pool: null argument");
  DS_REQUIRE(D == PD, DIFFSAL_E_SHAPE, "qkv_pool: head dimension %d (built for 96; use diffsal_pool3d_ln per tensor)", D);
  const bool ln = gamma != nullptr;
  DS_REQUIRE(!ln || (beta && eps && gamma[0] && gamma[1] && gamma[2] && beta[0] && beta[1] && beta[2]), DIFFSAL_E_ARG,
             "qkv_pool: LayerNorm form needs gamma, beta and eps of all three tensors");
  QkvPoolArgs a{};
  int rc = fill_geometry(a, B, heads, T, H, W, stride_q, stride_kv, "qkv_pool");
  if (rc) return rc;
  a.qkv = qkv;
  a.w_channel_major = w_channel_major ? 1 : 0;
  DS_REQUIRE(aligned16(qkv), DIFFSAL_E_ALIGN, "qkv_pool: misaligned qkv");
  DS_REQUIRE(dtype == DIFFSAL_F32 || dtype == DIFFSAL_BF16 || dtype == DIFFSAL_F16, DIFFSAL_E_ARG, "qkv_pool: dtype %d", dtype);
  for (int x = 0; x < 3; ++x) {
    a.w27[x] = w27[x]; a.out[x] = out[x];
    DS_REQUIRE(aligned16(w27[x]) && aligned16(out[x]), DIFFSAL_E_ALIGN, "qkv_pool: misaligned pointer");
    if (ln) {
      a.gamma[x] = gamma[x]; a.beta[x] = beta[x]; a.eps[x] = eps[x];
      DS_REQUIRE(aligned16(gamma[x]) && aligned16(beta[x]), DIFFSAL_E_ALIGN, "qkv_pool: misaligned LayerNorm parameter");
    }
  }
  long total = 0;
  for (int g = 0; g < 2; ++g) {
    const long rows = static_cast<long>(B) * heads * (static_cast<long>(a.To[g]) * a.Ho[g] * a.Wo[g] + 1);
    DS_REQUIRE(rows < (1L << 23), DIFFSAL_E_SHAPE, "qkv_pool: %ld output rows (the index arithmetic covers < 2^23)", rows);
    a.rows[g] = static_cast<int>(rows);
    total += rows * (g ? 2 : 1);
  }
  DS_REQUIRE(static_cast<long>(1 + T * H * W) * 3 * heads * PD < (1L << 31), DIFFSAL_E_SHAPE, "qkv_pool: clip too large");
  a.iters = passes_for(total);
  for (int g = 0; g < 2; ++g) a.blocks[g] = (a.rows[g] + PROWS * a.iters - 1) / (PROWS * a.iters);
  const unsigned grid = static_cast<unsigned>(a.blocks[0] + 2 * a.blocks[1]);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(TT)                                                                              \
  do {                                                                                        \
    if (ln) hipLaunchKernelGGL((qkv_pool_kernel<true, TT>), dim3(grid), dim3(256), 0, s, a);  \
    else hipLaunchKernelGGL((qkv_pool_kernel<false, TT>), dim3(grid), dim3(256), 0, s, a);    \
  } while (0)
  DS_DTYPE_DISPATCH(dtype, "qkv_pool", CALL);
#undef CALL
  return check_launch("qkv_pool");
}

extern "C" int diffsal_qkv_pool_bwd_data(const float* const* dy, const float* const* w27, float* dqkv, int B, int heads, int D,
                                         int T, int H, int W, const int* stride_q, const int* stride_kv, int w_channel_major,
                                         diffsal_stream_t stream) {
  DS_REQUIRE(dy && w27 && dqkv && stride_q && stride_kv && dy[0] && dy[1] && dy[2] && w27[0] && w27[1] && w27[2], DIFFSAL_E_ARG,
             "qkv_pool_bwd_data: null argument");
  DS_REQUIRE(D == PD, DIFFSAL_E_SHAPE, "qkv_pool_bwd_data: head dimension %d (built for 96)", D);
  DS_REQUIRE(stride_q[0] == 1 && stride_kv[0] == 1 && stride_q[1] == stride_q[2] && stride_kv[1] == stride_kv[2],
             DIFFSAL_E_SHAPE, "qkv_pool_bwd_data: temporal stride 1 and equal spatial strides only (use diffsal_pool3d_bwd_data)");
  QkvPoolArgs a{};
  int rc = fill_geometry(a, B, heads, T, H, W, stride_q, stride_kv, "qkv_pool_bwd_data");
  if (rc) return rc;
  a.dqkv = dqkv;
  a.w_channel_major = w_channel_major ? 1 : 0;
  DS_REQUIRE(aligned16(dqkv), DIFFSAL_E_ALIGN, "qkv_pool_bwd_data: misaligned dqkv");
  for (int x = 0; x < 3; ++x) {
    a.w27[x] = w27[x]; a.dy[x] = dy[x];
    DS_REQUIRE(aligned16(w27[x]) && aligned16(dy[x]), DIFFSAL_E_ALIGN, "qkv_pool_bwd_data: misaligned pointer");
  }
  const long rows = static_cast<long>(B) * (1 + static_cast<long>(T) * H * W) * heads;
  DS_REQUIRE(rows < (1L << 23), DIFFSAL_E_SHAPE, "qkv_pool_bwd_data: %ld input rows (the index arithmetic covers < 2^23)", rows);
  a.rows[0] = a.rows[1] = static_cast<int>(rows);
  a.iters = passes_for(3 * rows);
  a.blocks[0] = a.blocks[1] = (a.rows[0] + PROWS * a.iters - 1) / (PROWS * a.iters);
  const unsigned run_grid = plan_runs(a, tune(TUNE_NO_POOL_RUNS) <= 0, true, nullptr);
  if (run_grid) {
    hipLaunchKernelGGL(qkv_pool_bwd_data_runs_kernel, dim3(run_grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    rc = check_launch("qkv_pool_bwd_data(runs)");
    if (rc || (a.run_R[0] && a.run_R[1] && a.run_R[2])) return rc;
  }
  hipLaunchKernelGGL(qkv_pool_bwd_data_kernel, dim3(static_cast<unsigned>(3 * a.blocks[0])), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return check_launch("qkv_pool_bwd_data");
}

extern "C" int diffsal_qkv_pool_bwd_weight_chunks(int B, int heads, int T, int H, int W, const int* stride_q) {
  if (!stride_q || stride_q[0] < 1 || stride_q[1] < 1 || stride_q[2] < 1) return 512;
  const int Wo = (W - 1) / stride_q[2] + 1;
  const int R = (stride_q[2] == 1 && Wo % 8 == 0) ? 8 : 4;
  return qkv_wchunks(static_cast<long>(B) * heads * ((T - 1) / stride_q[0] + 1) * ((H - 1) / stride_q[1] + 1) * Wo / R);
}

extern "C" int diffsal_qkv_pool_bwd_weight(const float* qkv, const float* const* dy, double* part, int B, int heads, int D, int T,
                                           int H, int W, const int* stride_q, const int* stride_kv, int w_channel_major,
                                           diffsal_stream_t stream) {
  DS_REQUIRE(qkv && dy && part && stride_q && stride_kv && dy[0] && dy[1] && dy[2], DIFFSAL_E_ARG,
             "qkv_pool_bwd_weight: null argument");
  DS_REQUIRE(D == PD, DIFFSAL_E_SHAPE, "qkv_pool_bwd_weight: head dimension %d (built for 96)", D);
  QkvPoolArgs a{};
  int rc = fill_geometry(a, B, heads, T, H, W, stride_q, stride_kv, "qkv_pool_bwd_weight");
  if (rc) return rc;
  a.qkv = qkv;
  a.w_channel_major = w_channel_major ? 1 : 0;
  DS_REQUIRE(aligned16(qkv), DIFFSAL_E_ALIGN, "qkv_pool_bwd_weight: misaligned qkv");
  for (int x = 0; x < 3; ++x) {
    a.dy[x] = dy[x];
    DS_REQUIRE(aligned16(dy[x]), DIFFSAL_E_ALIGN, "qkv_pool_bwd_weight: misaligned gradient");
  }
  DS_REQUIRE(static_cast<long>(B) * heads * a.To[0] * a.Ho[0] * a.Wo[0] < (1L << 23), DIFFSAL_E_SHAPE,
             "qkv_pool_bwd_weight: too many output tokens (the index arithmetic covers < 2^23)");
  DS_REQUIRE(static_cast<long>(1 + T * H * W) * 3 * heads * PD < (1L << 31), DIFFSAL_E_SHAPE, "qkv_pool_bwd_weight: clip too large");
  plan_runs(a, tune(TUNE_NO_POOL_RUNS) <= 0, false, nullptr);
  const int chunks = diffsal_qkv_pool_bwd_weight_chunks(B, heads, T, H, W, stride_q);
  hipLaunchKernelGGL(qkv_pool_bwd_weight_kernel, dim3(chunks, 3), dim3(256), 0, static_cast<hipStream_t>(stream), a, part);
  return check_launch("qkv_pool_bwd_weight");
}

static int rel_tables_launch(bool bwd, const float* const* src, float* const* dst, const int* const* idx, const float* const* w,
                             const int* const* ptr, const int* rows, int D, diffsal_stream_t stream, const char* who) {
  DS_REQUIRE(src && dst && idx && w && rows && (!bwd || ptr), DIFFSAL_E_ARG, "%s: null argument", who);
  DS_REQUIRE(D > 0 && D % 4 == 0, DIFFSAL_E_SHAPE, "%s: D=%d", who, D);
  RelTablesArgs a{};
  int most = 0;
  for (int t = 0; t < 3; ++t) {
    DS_REQUIRE(src[t] && dst[t] && idx[t] && w[t] && (!bwd || ptr[t]) && rows[t] > 0, DIFFSAL_E_ARG, "%s: null table %d", who, t);
    DS_REQUIRE(aligned16(src[t]) && aligned16(dst[t]), DIFFSAL_E_ALIGN, "%s: misaligned table %d", who, t);
    a.src[t] = src[t]; a.dst[t] = dst[t]; a.idx[t] = idx[t]; a.w[t] = w[t]; a.ptr[t] = bwd ? ptr[t] : nullptr; a.rows[t] = rows[t];
    most = rows[t] > most ? rows[t] : most;
  }
  a.D4 = D / 4;
  const dim3 grid((static_cast<unsigned>(most) * a.D4 + 255) / 256, 3);
  if (bwd) hipLaunchKernelGGL(rel_tables_bwd_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(rel_tables_fwd_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch(who);
}

extern "C" int diffsal_rel_tables(const float* const* rel, const int* const* idx2, const float* const* w2, float* const* out,
                                  const int* M, int D, diffsal_stream_t stream) {
  return rel_tables_launch(false, rel, out, idx2, w2, nullptr, M, D, stream, "rel_tables");
}

extern "C" int diffsal_rel_tables_bwd(const float* const* dout, const int* const* csr_ptr, const int* const* csr_col,
                                      const float* const* csr_w, float* const* drel, const int* R, int D,
                                      diffsal_stream_t stream) {
  return rel_tables_launch(true, dout, drel, csr_col, csr_w, csr_ptr, R, D, stream, "rel_tables_bwd");
}

// D == 96 forms of diffsal_relpos_project / the query half of diffsal_relpos_project_bwd (mvit.hip dispatches here)
namespace diffsal {
int relpos_project96(const float* q, const float* Rt, const float* Rh, const float* Rw, float* extra, long rows, int qt, int qh,
                     int qw, int kt, int kh, int kw, int E, hipStream_t s) {
  hipLaunchKernelGGL(relpos_project96_kernel, dim3(static_cast<unsigned>((rows + PROWS - 1) / PROWS)), dim3(256), 0, s, q, Rt,
                     Rh, Rw, extra, qt, qh, qw, kt, kh, kw, static_cast<int>(rows), E);
  return check_launch("relpos_project");
}
int relpos_bwd_q96(const float* dE, const float* Rt, const float* Rh, const float* Rw, float* dq, long rows, int qt, int qh,
                   int qw, int kt, int kh, int kw, int accumulate, int E, hipStream_t s) {
  hipLaunchKernelGGL(relpos_bwd_q96_kernel, dim3(static_cast<unsigned>((rows + PROWS - 1) / PROWS)), dim3(256), 0, s, dE, Rt, Rh,
                     Rw, dq, qt, qh, qw, kt, kh, kw, accumulate, static_cast<int>(rows), E);
  return check_launch("relpos_project_bwd(q)");
}
}  // namespace diffsal
