// 3x3 stride-1 "same" convolutions (dilation d = padding, d in {1, 2}) of the fp32 denoiser as Winograd F(2x2, 3x3):
// 16 multiplications per 2x2 output tile, input channel and output channel instead of 36 -- the matrix pipe does 2.25x
// fewer MFMAs for the same convolution.  Replaces diffsal_conv_igemm on ResnetBlock.conv1/conv2 (R/models/saliency_decoder/
// sal_unet.py:104-142) and UpEmbed's second convolution (common_block.py:196-216) when the shape qualifies
// (diffsal_conv_wino_supported); results differ from the direct fp32 convolution by ~1e-6 relative (transform rounding), far
// inside the 1e-3 parity bar; DIFFSAL_NO_WINOGRAD=1 keeps the direct kernel.
//
//   Y = A^T [ sum_ci (G g G^T) o (B^T x B) ] A        per tile; o = element-wise over the 16 positions xi = (i, j)
//
// A dilated convolution is d*d independent undilated ones on the polyphase sub-grids (rows y = py mod d, columns x = px
// mod d): a tile is (image, py, px, ty, tx) and covers outputs (py + d(2ty + a), px + d(2tx + b)), a, b in {0, 1}.
//
// Three steps:
//  1. wino_input_kernel   V[chunk][tile block][xi][64 tiles][8 ci] = B^T x B   (8 input channels per chunk; 4x the input bytes,
//                         written once, read once per block of 64 output channels -- from the Infinity Cache for the layers that
//                         use this path)
//  2. wino_gemm_kernel    a workgroup owns 64 tiles x 64 output channels and ALL 16 xi: 4 waves as 2 x 2, each 32 tiles x 32
//                         channels x 16 xi = 16 MFMA accumulators (256 registers, one wave per SIMD); both operands arrive as
//                         contiguous 32 KB blocks per chunk by LDS-DMA (U is the host-transformed weight in the same blocked
//                         layout) into two LDS stages, one barrier per chunk of 64 MFMAs per wave; the output transform A^T M A
//                         runs on the accumulators (all 16 xi of a (tile, channel) sit in one lane) and feeds the usual epilogue.
//  3. wino_reduce_kernel  only when the input channels are split over workgroups (few tiles x channel blocks): sums the partial
//                         outputs -- already transformed, the transform is linear -- in a fixed order and applies the epilogue.
#include "common.h"

namespace diffsal {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoGeom {
  int N, H, W, Cin, Cout, d;
  int TY, TX;          // tiles per sub-grid
  int n_tiles;         // N * d*d * TY * TX
  int tile_blocks;     // ceil(n_tiles / 64)
  int chunks;          // Cin / 8
  int cout_blocks;     // ceil(Cout / 64)
};

__device__ __forceinline__ void wino_tile_coords(const WinoGeom& g, int t, int& n, int& y0, int& x0) {
  const int per_class = g.TY * g.TX, per_img = g.d * g.d * per_class;
  n = t / per_img;
  const int r = t - n * per_img;
  const int cls = r / per_class, rr = r - cls * per_class;
  const int py = cls / g.d, px = cls - py * g.d;
  const int ty = rr / g.TX, tx = rr - ty * g.TX;
  y0 = py + g.d * 2 * ty;      // first output row of the tile; its rows are y0, y0 + d; input rows y0 + d (i - 1), i = 0..3
  x0 = px + g.d * 2 * tx;
}

// ------------------------------------------------------------------------------------------------------------------------
// 1. input transform.  item = (tile, channel quad); lanes: 8 quads (one 128-byte run of a pixel) fastest, then tiles.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, WinoGeom g) {
  const int T = g.tile_blocks * 64;
  const long items = static_cast<long>(T) * (g.Cin / 4);
  for (long it = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; it < items; it += static_cast<long>(gridDim.x) * 256) {
    const int qlo = static_cast<int>(it & 7);
    const long rest = it >> 3;
    const int t = static_cast<int>(rest % T);
    const int q4 = static_cast<int>(rest / T) * 8 + qlo;
    float4 dd[4][4];
    if (t < g.n_tiles) {
      int n, y0, x0;
      wino_tile_coords(g, t, n, y0, x0);
      const float* base = x + static_cast<long>(n) * g.H * g.W * g.Cin + q4 * 4;
      unsigned ok = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = y0 + g.d * (i - 1);
        const bool vy = y >= 0 && y < g.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int xx = x0 + g.d * (j - 1);
          const bool v = vy && xx >= 0 && xx < g.W;
          dd[i][j] = ld4(base + (v ? (y * g.W + xx) : 0) * static_cast<long>(g.Cin));   // raw loads first, masks after
          ok |= v ? 1u << (i * 4 + j) : 0u;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!((ok >> (i * 4 + j)) & 1u)) dd[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) dd[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
    auto add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    float4 tt[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {      // B^T d : rows
      tt[0][j] = sub(dd[0][j], dd[2][j]);
      tt[1][j] = add(dd[1][j], dd[2][j]);
      tt[2][j] = sub(dd[2][j], dd[1][j]);
      tt[3][j] = sub(dd[1][j], dd[3][j]);
    }
    const int chunk = q4 >> 1, half = q4 & 1;
    float* dst = V + ((static_cast<long>(chunk) * g.tile_blocks + (t >> 6)) * 16 * 64 + (t & 63)) * 8 + half * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // (.) B : columns
      st4(dst + (i * 4 + 0) * 64 * 8, sub(tt[i][0], tt[i][2]));
      st4(dst + (i * 4 + 1) * 64 * 8, add(tt[i][1], tt[i][2]));
      st4(dst + (i * 4 + 2) * 64 * 8, sub(tt[i][2], tt[i][1]));
      st4(dst + (i * 4 + 3) * 64 * 8, sub(tt[i][1], tt[i][3]));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// 2. the 16 products and the output transform
// ------------------------------------------------------------------------------------------------------------------------
struct WinoArgs {
  const float* V;
  const float* U;
  float* out;            // NHWC [N][H][W][Cout], or the partial slabs [splits][N*H*W][Cout] when splits > 1
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const float* residual;
  int act, rowvec_ld;
  int splits, chunks_per_split;
  // tail mode (splits == 1, more than one round of workgroups): blocks [0, n_full) are whole; each later block is cut into
  // tail_split pieces of tail_chunks chunks that write partial outputs to `tail_slabs` (wino_tail_reduce_kernel finishes them)
  int n_full, tail_split, tail_chunks;
  float* tail_slabs;
  WinoGeom g;
};

// float4 slot of (xi, row, q) in a stage of [16][64][2] float4: the two float4 of a row trade places on rows 8-15 (mod 16), so
// the 16 lanes of one ds_read_b128 pass (16 consecutive rows, same q) cover all 64 banks
__device__ __forceinline__ int wino_slot(int f) { return f ^ ((f >> 4) & 1); }

__device__ __forceinline__ float wino_act(float x, int act) {
  if (act == DIFFSAL_ACT_RELU) return fmaxf(x, 0.f);
  if (act == DIFFSAL_ACT_GELU_ERF) return gelu_erf(x);
  if (act == DIFFSAL_ACT_SIGMOID) return sigmoidf_(x);
  return x;
}

__global__ __launch_bounds__(256) void wino_gemm_kernel(WinoArgs p) {
  // two stages of [V 2048 | U 2048] float4 as TWO objects: hipcc orders an LDS read behind every LDS-DMA in flight (s_waitcnt
  // vmcnt(0) before each fragment read) unless it can see that they touch different variables
  __shared__ __attribute__((aligned(16))) float wino_st0[16384];
  __shared__ __attribute__((aligned(16))) float wino_st1[16384];
  const WinoGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the LDS-DMA destinations are wave-uniform
  const int wt = wave >> 1, wc = wave & 1;
  // workgroups are dealt round-robin over the 8 XCDs: walk the (tile block, channel block) list so that one XCD owns a
  // contiguous run of it -- the channel blocks of a tile block then read their shared V block from ONE L2
  int vb, tail_piece = -1;
  if (static_cast<int>(blockIdx.x) < p.n_full) {
    const int nwg = p.n_full, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    vb = xcd * q + (xcd < r ? xcd : r) + (b >> 3);
  } else {                                   // a piece of a tail block: dispatched after every whole block
    const int t = static_cast<int>(blockIdx.x) - p.n_full;
    vb = p.n_full + t / p.tail_split;
    tail_piece = t - (vb - p.n_full) * p.tail_split;
  }
  const int tb = vb / g.cout_blocks, cb = vb - tb * g.cout_blocks;
  const int c_begin = tail_piece >= 0 ? tail_piece * p.tail_chunks : blockIdx.y * p.chunks_per_split;
  const int c_end = min(g.chunks, c_begin + (tail_piece >= 0 ? p.tail_chunks : p.chunks_per_split));

  f32x16 acc[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

  // Operand staging: LDS-DMA (global_load_lds_dwordx4: global -> LDS, no registers -- the 256 accumulators leave none for a
  // second in-flight chunk).  An instruction fills 64 consecutive float4 slots; the bank swizzle (wino_slot) is applied on the
  // SOURCE address: the lane that fills slot s loads float4 wino_slot(s) of the block, and the fragment reads apply the same
  // involution.  The pieces of chunk c + 1 are issued between the MFMAs of the first half of chunk c (four per pair of positions)
  // into the other stage; the __syncthreads() at the end of the chunk retires them (vmcnt(0)) before anyone reads that stage.
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  const unsigned src_off = 16u * static_cast<unsigned>(tid ^ ((lane >> 4) & 1));   // bytes; the only per-lane address part
  auto block_v = [&](int c) { return reinterpret_cast<const char*>(p.V + (static_cast<long>(c) * g.tile_blocks + tb) * 8192); };
  auto block_u = [&](int c) { return reinterpret_cast<const char*>(p.U + (static_cast<long>(c) * g.cout_blocks + cb) * 8192); };
  // uniform base (scalar registers) + one 32-bit lane offset: the sixteen pieces of a chunk share ONE address register
  auto piece = [&](const char* vsrc, const char* usrc, float* dst_stage, int i) __attribute__((always_inline)) {
    float* dv = dst_stage + (i * 256 + wave * 64) * 4;                           // wave-uniform destination
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(vsrc + 4096 * i + src_off), (lds_ptr_t)dv, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(usrc + 4096 * i + src_off), (lds_ptr_t)(dv + 8192), 16, 0, 0);
  };

  const int hf = lane >> 5;
  // swizzled float4 slot of this lane's fragment of position 0; position xi is 128 slots further (the swizzle bit is bit 4)
  const int frag_b = wino_slot((wt * 32 + (lane & 31)) * 2 + hf);
  const int frag_a = wino_slot((wc * 32 + (lane & 31)) * 2 + hf);
  if (c_begin < c_end) {
    const char* v0 = block_v(c_begin);
    const char* u0 = block_u(c_begin);
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(v0, u0, wino_st0, i);
  }
  __syncthreads();
  // one chunk: 16 positions x 4 MFMAs on stage `cur`, the 16 DMA pieces of the next chunk into stage `nxt`
  auto chunk = [&](const float* cur, float* nxt, int c) __attribute__((always_inline)) {
    const float* vs = cur;
    const float* us = cur + 8192;
    const int cn = c + 1 < c_end ? c + 1 : c;                // last chunk: a harmless re-load instead of a branch
    const char* vn = block_v(cn);
    const char* un = block_u(cn);
    // two positions at a time, their MFMAs alternating (four back-to-back MFMAs on ONE accumulator wait for each other's
    // results); the fragments of the next pair are requested before this pair's MFMAs
    float4 a0 = ld4(us + 4 * frag_a), b0 = ld4(vs + 4 * frag_b);
    float4 a1 = ld4(us + 4 * (frag_a + 128)), b1 = ld4(vs + 4 * (frag_b + 128));
#pragma unroll
    for (int xi = 0; xi < 16; xi += 2) {
      float4 a0n = a0, b0n = b0, a1n = a1, b1n = b1;
      if (xi + 2 < 16) {
        a0n = ld4(us + 4 * (frag_a + (xi + 2) * 128));
        b0n = ld4(vs + 4 * (frag_b + (xi + 2) * 128));
        a1n = ld4(us + 4 * (frag_a + (xi + 3) * 128));
        b1n = ld4(vs + 4 * (frag_b + (xi + 3) * 128));
      }
      if (xi < 8) {                       // all sixteen pieces in the first half of the chunk: they have landed by its barrier
        piece(vn, un, nxt, xi);
        piece(vn, un, nxt, xi + 1);
      }
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[xi], 0, 0, 0);   // rows = output channels, cols = tiles
      acc[xi + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[xi + 1], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[xi], 0, 0, 0);
      acc[xi + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[xi + 1], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[xi], 0, 0, 0);
      acc[xi + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[xi + 1], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[xi], 0, 0, 0);
      acc[xi + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[xi + 1], 0, 0, 0);
      // issue order inside the pair: the four fragment reads, then MFMAs with the two DMA pieces slipped between them
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      if (xi < 8) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a0 = a0n; b0 = b0n; a1 = a1n; b1 = b1n;
    }
    __syncthreads();
  };
  for (int c = c_begin; c < c_end; c += 2) {      // the host keeps every split's chunk count even (no branch between the halves:
    chunk(wino_st0, wino_st1, c);                 // a conditional second half makes hipcc spill accumulators at the join)
    chunk(wino_st1, wino_st0, c + 1);
  }

  // ---- output transform + epilogue.  Lane: tile = column (lane & 31), channels co0 + 8 (r >> 2) + 4 hf + (r & 3).
  const int t = tb * 64 + wt * 32 + (lane & 31);
  if (t >= g.n_tiles) return;
  int n, y0, x0;
  wino_tile_coords(g, t, n, y0, x0);
  const long HW = static_cast<long>(g.H) * g.W;
  float* outp = tail_piece >= 0 ? p.tail_slabs + static_cast<long>(tail_piece) * g.N * HW * g.Cout
                                : p.out + (p.splits > 1 ? static_cast<long>(blockIdx.y) * g.N * HW * g.Cout : 0);
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int co = cb * 64 + wc * 32 + 8 * r4 + 4 * hf;
    if (co >= g.Cout) continue;                       // Cout % 4 == 0
    float y[2][2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = r4 * 4 + e;
      float s[2][4];                                  // A^T M : rows
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[0][j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
        s[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
      }
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) {                // (.) A : columns
        y[a2][0][e] = s[a2][0] + s[a2][1] + s[a2][2];
        y[a2][1][e] = s[a2][1] - s[a2][2] - s[a2][3];
      }
    }
    float4 bi = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bi, rw = bi;
    const bool epi = p.splits == 1 && tail_piece < 0;
    if (epi) {
      if (p.bias) bi = ld4(p.bias + co);
      if (p.scale) { sc = ld4(p.scale + co); sh = ld4(p.shift + co); }
      if (p.rowvec) rw = ld4(p.rowvec + static_cast<long>(n) * p.rowvec_ld + co);
    }
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
      const int oy = y0 + g.d * a2;
      if (oy >= g.H) continue;
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int ox = x0 + g.d * b2;
        if (ox >= g.W) continue;
        const long o = (static_cast<long>(n) * HW + static_cast<long>(oy) * g.W + ox) * g.Cout + co;
        float v[4] = {y[a2][b2][0], y[a2][b2][1], y[a2][b2][2], y[a2][b2][3]};
        if (epi) {
          const float bb[4] = {bi.x, bi.y, bi.z, bi.w}, ss[4] = {sc.x, sc.y, sc.z, sc.w}, hh[4] = {sh.x, sh.y, sh.z, sh.w},
                      rr[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float xv = v[e] + bb[e];
            if (p.scale) xv = xv * ss[e] + hh[e];
            xv += rr[e];
            v[e] = wino_act(xv, p.act);
          }
          if (p.residual) {
            const float4 rs = ld4(p.residual + o);
            v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
          }
        }
        st4(outp + o, make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  }
}

// 3. split sum (fixed order) + epilogue; float4 over Cout
__global__ __launch_bounds__(256) void wino_reduce_kernel(WinoArgs p, const float* __restrict__ slabs, float* __restrict__ out) {
  const WinoGeom& g = p.g;
  const int n4 = g.Cout >> 2;
  const long HW = static_cast<long>(g.H) * g.W;
  const long total = static_cast<long>(g.N) * HW * n4, slab = static_cast<long>(g.N) * HW * g.Cout;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int co = static_cast<int>(i % n4) * 4;
    const long m = i / n4;
    const long o = m * g.Cout + co;
    float4 a = ld4(slabs + o);
    for (int s = 1; s < p.splits; ++s) {
      const float4 b = ld4(slabs + s * slab + o);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xv = v[e];
      if (p.bias) xv += p.bias[co + e];
      if (p.scale) xv = xv * p.scale[co + e] + p.shift[co + e];
      if (p.rowvec) xv += p.rowvec[(m / HW) * p.rowvec_ld + co + e];
      xv = wino_act(xv, p.act);
      if (p.residual) xv += p.residual[o + e];
      v[e] = xv;
    }
    st4(out + o, make_float4(v[0], v[1], v[2], v[3]));
  }
}

// tail mode: sums the pieces of the tail blocks (tiles >= tile0, every channel) in a fixed order and applies the epilogue
__global__ __launch_bounds__(256) void wino_tail_reduce_kernel(WinoArgs p, float* __restrict__ out, int tile0) {
  const WinoGeom& g = p.g;
  const int n4 = g.Cout >> 2;
  const long HW = static_cast<long>(g.H) * g.W, slab = static_cast<long>(g.N) * HW * g.Cout;
  const long total = static_cast<long>(g.n_tiles - tile0) * 4 * n4;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long>(gridDim.x) * 256) {
    const int co = static_cast<int>(i % n4) * 4;
    const long r = i / n4;
    const int px = static_cast<int>(r & 3), t = tile0 + static_cast<int>(r >> 2);
    int n, y0, x0;
    wino_tile_coords(g, t, n, y0, x0);
    const int oy = y0 + g.d * (px >> 1), ox = x0 + g.d * (px & 1);
    if (oy >= g.H || ox >= g.W) continue;
    const long o = (static_cast<long>(n) * HW + static_cast<long>(oy) * g.W + ox) * g.Cout + co;
    float4 a = ld4(p.tail_slabs + o);
    for (int s = 1; s < p.tail_split; ++s) {
      const float4 b = ld4(p.tail_slabs + s * slab + o);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xv = v[e];
      if (p.bias) xv += p.bias[co + e];
      if (p.scale) xv = xv * p.scale[co + e] + p.shift[co + e];
      if (p.rowvec) xv += p.rowvec[static_cast<long>(n) * p.rowvec_ld + co + e];
      xv = wino_act(xv, p.act);
      if (p.residual) xv += p.residual[o + e];
      v[e] = xv;
    }
    st4(out + o, make_float4(v[0], v[1], v[2], v[3]));
  }
}

namespace {

struct WinoPlan { WinoGeom g; int splits, chunks_per_split, n_full, tail_split, tail_chunks; size_t v_bytes, slab_bytes; };

bool wino_shape_ok(const diffsal_conv_desc* d) {
  return d && d->KH == 3 && d->KW == 3 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == d->dil_w &&
         (d->dil_h == 1 || d->dil_h == 2) && d->pad_t == d->dil_h && d->pad_l == d->dil_w && d->Ho == d->H && d->Wo == d->W &&
         d->Cin > 0 && d->Cin % 32 == 0 && d->Cout > 0 && d->Cout % 4 == 0 && d->N > 0 && d->H > 0 && d->W > 0 &&
         d->dtype == DIFFSAL_F32 && d->precision == DIFFSAL_PREC_FP32;
}

WinoPlan wino_plan(const diffsal_conv_desc* d) {
  WinoPlan pl{};
  WinoGeom& g = pl.g;
  g.N = d->N; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout; g.d = d->dil_h;
  g.TY = ((g.H + g.d - 1) / g.d + 1) / 2;
  g.TX = ((g.W + g.d - 1) / g.d + 1) / 2;
  g.n_tiles = g.N * g.d * g.d * g.TY * g.TX;
  g.tile_blocks = (g.n_tiles + 63) / 64;
  g.chunks = g.Cin / 8;
  g.cout_blocks = (g.Cout + 63) / 64;
  // one workgroup per CU (128 KB of LDS): split the input channels when tiles x channel blocks do not fill the chip, keeping
  // at least 12 chunks (96 channels) per workgroup
  const int wgs = g.tile_blocks * g.cout_blocks;
  int s = 1;
  if (wgs < 128) {                               // never more than one round of 256 workgroups
    s = 256 / wgs;
    while (s > 1 && g.chunks / s < 12) --s;
    s = s > 8 ? 8 : s;
  }
  pl.chunks_per_split = ((g.chunks + s - 1) / s + 1) & ~1;       // even (the kernel walks two chunks per iteration); chunks is even
  pl.splits = (g.chunks + pl.chunks_per_split - 1) / pl.chunks_per_split;
  s = pl.splits;
  // More than one round (one workgroup per CU): the last, partly filled round costs as much as a full one.  Cut ITS blocks
  // (whole tile blocks, so the finishing pass owns complete rows of tiles) into pieces that fill the chip once more:
  // 324 blocks = 256 whole + 68 x 3 pieces of 16 chunks instead of a second round of 48 chunks on a quarter of the CUs.
  pl.n_full = wgs;
  pl.tail_split = 1;
  pl.tail_chunks = g.chunks;
  if (s == 1 && wgs > 256) {
    int n_full = (wgs / 256) * 256;
    n_full -= n_full % g.cout_blocks;
    const int rest = wgs - n_full;
    int ts = rest > 0 ? 256 / rest : 1;
    while (ts > 1 && (((g.chunks + ts - 1) / ts + 1) & ~1) < 12) --ts;
    if (rest > 0 && ts >= 2) {
      pl.n_full = n_full;
      pl.tail_chunks = ((g.chunks + ts - 1) / ts + 1) & ~1;
      pl.tail_split = (g.chunks + pl.tail_chunks - 1) / pl.tail_chunks;
    }
  }
  pl.v_bytes = static_cast<size_t>(g.chunks) * g.tile_blocks * 16 * 64 * 8 * sizeof(float);
  const int slabs = s > 1 ? s : (pl.tail_split > 1 ? pl.tail_split : 0);
  pl.slab_bytes = static_cast<size_t>(slabs) * g.N * g.H * g.W * g.Cout * sizeof(float);
  return pl;
}

}  // namespace
}  // namespace diffsal

using namespace diffsal;

// 1 when diffsal_conv_wino accepts the descriptor AND the planner expects it to beat the direct kernel (enough input and
// output channels to pay for the 4x-sized transformed input)
extern "C" int diffsal_conv_wino_supported(const diffsal_conv_desc* d) {
  if (!wino_shape_ok(d) || tune(TUNE_NO_WINOGRAD) == 1) return 0;
  if (tune(TUNE_FORCE_WINOGRAD) == 1) return 1;
  // Measured (tools/bench_wino.py stand-alone, bench.py in the step, B = 4): the products run at ~2.3 us per 8-channel
  // chunk and round of <= 256 workgroups (one per CU, one wave per SIMD).  Stand-alone against the direct kernel: the six
  // ResnetBlock convolutions 1.36-1.65x, UpEmbed's second convolutions of stages 1 / 2 (36 frames, dilation 2) 1.45x / 1.33x
  // once their last, partly filled round is cut into pieces (tail mode), stage 3 (Cout = 96: a quarter of the 64-channel blocks
  // is padding, 300 MB of transformed input) 0.91x.  In the step: class K4 0.66 -> 0.51 ms, K12 1.26 -> 1.14 ms.  The
  // transformed input is 4x the input bytes, written and read back: 160 MB bounds what still pays.  Fewer than 128 workgroups
  // (single clips at full resolution) leave half the chip idle at one workgroup per CU: direct.
  const WinoPlan pl = wino_plan(d);
  const int wgs = pl.g.tile_blocks * pl.g.cout_blocks * pl.splits;
  return (pl.v_bytes <= 160u * 1000u * 1000u && wgs >= 128 && d->Cout >= 128) ? 1 : 0;
}

extern "C" size_t diffsal_conv_wino_ws_bytes(const diffsal_conv_desc* d) {
  if (!wino_shape_ok(d)) return 0;
  const WinoPlan pl = wino_plan(d);
  return pl.v_bytes + pl.slab_bytes;
}

// U: the transformed weight G g G^T in blocked layout [Cin/8][ceil(Cout/64)][16][64][8] (ops.pack_wino_weight)
extern "C" int diffsal_conv_wino(const diffsal_conv_desc* d, const float* x, const float* U, const float* bias, const float* scale,
                                 const float* shift, const float* rowvec, const float* residual, float* out, void* ws,
                                 size_t ws_bytes, diffsal_stream_t stream) {
  DS_REQUIRE(d && x && U && out && ws, DIFFSAL_E_ARG, "conv_wino: null argument");
  DS_REQUIRE(wino_shape_ok(d), DIFFSAL_E_SHAPE,
             "conv_wino: fp32 3x3 stride-1 convolutions with padding = dilation in {1, 2}, Cin %% 32 == 0, Cout %% 4 == 0 only");
  DS_REQUIRE((scale == nullptr) == (shift == nullptr), DIFFSAL_E_ARG, "conv_wino: scale and shift come together");
  const WinoPlan pl = wino_plan(d);
  DS_REQUIRE(ws_bytes >= pl.v_bytes + pl.slab_bytes, DIFFSAL_E_ARG, "conv_wino: workspace of %zu bytes, need %zu", ws_bytes,
             pl.v_bytes + pl.slab_bytes);
  DS_REQUIRE(aligned16(x) && aligned16(U) && aligned16(out) && aligned16(ws) && (!bias || aligned16(bias)) &&
                 (!scale || (aligned16(scale) && aligned16(shift))) && (!rowvec || aligned16(rowvec)) &&
                 (!residual || aligned16(residual)) && (!rowvec || d->rowvec_ld % 4 == 0),
             DIFFSAL_E_ALIGN, "conv_wino: misaligned pointer");
  DS_REQUIRE(static_cast<long>(d->N) * d->H * d->W * d->Cin < (1L << 31), DIFFSAL_E_SHAPE, "conv_wino: input too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* V = static_cast<float*>(ws);
  float* slabs = reinterpret_cast<float*>(static_cast<char*>(ws) + pl.v_bytes);
  const long items = static_cast<long>(pl.g.tile_blocks) * 64 * (pl.g.Cin / 4);
  long gi = (items + 255) / 256;
  gi = gi > 16384 ? 16384 : gi;
  hipLaunchKernelGGL(wino_input_kernel, dim3(static_cast<unsigned>(gi)), dim3(256), 0, s, x, V, pl.g);
  int rc = check_launch("conv_wino(input transform)");
  if (rc) return rc;
  WinoArgs a{};
  a.V = V; a.U = U; a.out = pl.splits > 1 ? slabs : out;
  a.bias = bias; a.scale = scale; a.shift = shift; a.rowvec = rowvec; a.residual = residual;
  a.act = d->act; a.rowvec_ld = d->rowvec_ld; a.splits = pl.splits; a.chunks_per_split = pl.chunks_per_split; a.g = pl.g;
  a.n_full = pl.n_full; a.tail_split = pl.tail_split; a.tail_chunks = pl.tail_chunks; a.tail_slabs = slabs;
  const int n_blocks = pl.g.tile_blocks * pl.g.cout_blocks;
  const int grid_x = pl.n_full + (n_blocks - pl.n_full) * pl.tail_split;
  hipLaunchKernelGGL(wino_gemm_kernel, dim3(grid_x, pl.splits), dim3(256), 0, s, a);   // 128 KB of static LDS
  note_kernel("wino_gemm_kernel [%s]", pl.splits > 1 ? "input channels split" : (pl.tail_split > 1 ? "tail pieces" : "whole blocks"));
  rc = check_launch("conv_wino(products)");
  if (rc) return rc;
  if (pl.tail_split > 1) {
    const int tile0 = (pl.n_full / pl.g.cout_blocks) * 64;
    const long items = static_cast<long>(pl.g.n_tiles - tile0) * 4 * (d->Cout / 4);
    long gt = (items + 255) / 256;
    gt = gt > 2048 ? 2048 : gt;
    hipLaunchKernelGGL(wino_tail_reduce_kernel, dim3(static_cast<unsigned>(gt)), dim3(256), 0, s, a, out, tile0);
    return check_launch("conv_wino(tail sum)");
  }
  if (pl.splits == 1) return rc;
  const long total4 = static_cast<long>(d->N) * d->H * d->W * (d->Cout / 4);
  long gr = (total4 + 255) / 256;
  gr = gr > 2048 ? 2048 : gr;
  hipLaunchKernelGGL(wino_reduce_kernel, dim3(static_cast<unsigned>(gr)), dim3(256), 0, s, a, slabs, out);
  return check_launch("conv_wino(split sum)");
}
