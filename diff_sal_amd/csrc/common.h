// Shared helpers for the gfx950 kernels of libdiffsal_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/diffsal.h"

namespace diffsal {

void set_error(const char* fmt, ...);
// which kernel (and tile plan) the GEMM-family entry points launched last on this thread: diffsal_last_gemm_kernel()
void note_kernel(const char* fmt, ...);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return DIFFSAL_E_LAUNCH;
  }
  return DIFFSAL_OK;
}

#define DS_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      diffsal::set_error(__VA_ARGS__); \
      return (code);                 \
    }                                \
  } while (0)

constexpr int kWave = 64;  // CDNA wavefront

// Test / tuning switches ("Environment switches").  The environment is read ONCE, when the library is loaded; afterwards
// a switch changes only through diffsal_set_tuning() -- no getenv on a launch path, and no way for an inherited variable to
// reach a kernel argument.  tune(key) < 0 means "unset".
enum TuneKey {
  TUNE_NO_PERSIST = 0, TUNE_NO_XCD_ORDER, TUNE_NO_HALO, TUNE_FORCE_HALO, TUNE_IGEMM_CFG, TUNE_IGEMM16_CFG, TUNE_PLAN_DEBUG,
  TUNE_WGRAD_CFG, TUNE_WGRAD_SPLITS, TUNE_WGRAD_VERBOSE, TUNE_NO_FUSED_BLOCK, TUNE_NO_WINOGRAD, TUNE_FORCE_WINOGRAD, TUNE_GN_CHUNKS,
  TUNE_GN_APPLY_WGS, TUNE_GEMM_DMA, TUNE_CONV_DMA, TUNE_GROUP_GRID, TUNE_GEMM_DMA16, TUNE_WGRAD_DMA, TUNE_NO_GN_SLAB, TUNE_NO_WINOGRAD4, TUNE_BATCH_TILE, TUNE_BATCH_XCD, TUNE_NO_TAPSUM_ROWS, TUNE_TAPSUM_ROWS_FORM,
  TUNE_NO_ATTN_BWD_DS, TUNE_NO_POOL_RUNS, TUNE_NO_ATTN_SLOTS, TUNE_NO_STREAM16, TUNE_CONV16_TILE, TUNE_BLOCK16_WAVES, TUNE_NO_ATTN16_MFMA, TUNE_CONV16_HALF, TUNE_COUNT
};
int tune(int key);

// Butterfly all-reduce over `width` (power of two <= 64) consecutive lanes, entirely on the VALU: quad permutes for the
// strides 1 and 2, row_half_mirror / row_mirror for 4 and 8 (every quad / 8-lane group already holds its own total, so the
// mirrored partner carries the same value the xor partner would), v_permlane16_swap / v_permlane32_swap for 16 and 32.
// `__shfl_xor` compiles to ds_bpermute_b32 -- an LDS round trip of ~100 cycles per step on the critical path of every
// LayerNorm-style reduction.  Bit-identical to the xor butterfly (each step adds the same two operands).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_xor16(float v) {   // value of lane ^ 16
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // odd rows of r[0] <-> even rows of r[1]
  return __builtin_bit_cast(float, (__lane_id() & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {   // value of lane ^ 32
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // lanes 32-63 of r[0] <-> lanes 0-31 of r[1]
  return __builtin_bit_cast(float, (__lane_id() & 32) ? r[0] : r[1]);
}
template <int WIDTH, typename OP>
__device__ __forceinline__ float group_reduce(float v, OP op) {
  static_assert(WIDTH >= 1 && WIDTH <= 64 && (WIDTH & (WIDTH - 1)) == 0, "power of two <= 64");
  if constexpr (WIDTH >= 2) v = op(v, dpp_move<0xB1>(v));     // quad_perm [1,0,3,2]
  if constexpr (WIDTH >= 4) v = op(v, dpp_move<0x4E>(v));     // quad_perm [2,3,0,1]
  if constexpr (WIDTH >= 8) v = op(v, dpp_move<0x141>(v));    // row_half_mirror
  if constexpr (WIDTH >= 16) v = op(v, dpp_move<0x140>(v));   // row_mirror
  if constexpr (WIDTH >= 32) v = op(v, lane_xor16(v));
  if constexpr (WIDTH >= 64) v = op(v, lane_xor32(v));
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
  return group_reduce<WIDTH>(v, [](float a, float b) { return a + b; });
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
  return group_reduce<WIDTH>(v, [](float a, float b) { return fmaxf(a, b); });
}

__device__ __forceinline__ float swishf(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// x sigmoid(x) on the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp each, five instructions instead of the ~30 of
// libm's expf + an IEEE division): for kernels that evaluate it per LOADED element (the F(4x4) input transform reads every
// pixel 2.25 times).  exp2 of a large positive argument overflows to +inf, rcp(+inf) = 0: the limit x -> -inf is exact.
__device__ __forceinline__ float swishf_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26): one v_exp, one v_rcp, six FMAs -- a third of the
// instructions of libm's erff; the difference is far below the fp32 noise of the surrounding GEMMs.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// d/dx of erf-GELU: Phi(x) + x * phi(x), with erf_as's polynomial; its exp(-x^2 / 2) is the one phi needs (one v_exp, one v_rcp
// and ten FMAs per element -- cheap enough for a GEMM epilogue).  act_bwd_kernel (mode 2) uses the same function.
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float erf_x = copysignf(1.0f - p * t * e, x);
  return fmaf(x * 0.3989422804014327f, e, 0.5f * (1.0f + erf_x));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// N groups of MFMAs whose A fragment is one 16-byte LDS read: the read of group i + 1 is issued BEFORE the MFMAs of group i
// (pinned with sched_barrier: left alone, hipcc reads a fragment, waits for it, then issues its four dependent MFMAs, and a
// quarter of the matrix time is LDS latency).  addr(i) -> const float*, body(i, fragment).
template <int N, typename AddrF, typename BodyF>
__device__ __forceinline__ void mfma_groups_f32(AddrF addr, BodyF body) {
  float4 a = ld4(addr(0));
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float4 an = a;
    if (i + 1 < N) an = ld4(addr(i + 1));
    __builtin_amdgcn_sched_barrier(0);
    body(i, a);
    __builtin_amdgcn_sched_barrier(0);
    a = an;
  }
}
// The same for products whose A operand is ONE scalar LDS read per MFMA (the LDS tile is stored [contraction row][feature], the
// operand wants a feature column): group i = the PER scalars of PER MFMAs, read before the MFMAs of group i - 1.
// addr(i, j) -> const float* of scalar j of group i; body(i, fragment array).
template <int N, int PER, typename AddrF, typename BodyF>
__device__ __forceinline__ void mfma_groups_scalar_f32(AddrF addr, BodyF body) {
  float a[2][PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) a[0][j] = *addr(0, j);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i + 1 < N) {
#pragma unroll
      for (int j = 0; j < PER; ++j) a[(i + 1) & 1][j] = *addr(i + 1, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    body(i, a[i & 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
#define DS_MFMA4(ACC, A, B0, B1, B2, B3)                                         \
  do {                                                                           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A).x, (B0), ACC, 0, 0, 0);       \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A).y, (B1), ACC, 0, 0, 0);       \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A).z, (B2), ACC, 0, 0, 0);       \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A).w, (B3), ACC, 0, 0, 0);       \
  } while (0)

// 16-bit storage (DIFFSAL_BF16 / DIFFSAL_F16): the same 4-channel accessors on 8-byte runs; arithmetic stays fp32,
// one round-to-nearest-even on the way out.
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xFFFF0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xFFFF0000u));
}
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  const f32x4_t f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<bf16x4_t*>(p) = __builtin_convertvector(f, bf16x4_t);
}
__device__ __forceinline__ float4 ld4(const f16_t* p) {
  const f16x4_t h = *reinterpret_cast<const f16x4_t*>(p);
  const f32x4_t f = __builtin_convertvector(h, f32x4_t);
  return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ void st4(f16_t* p, float4 v) {
  const f32x4_t f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<f16x4_t*>(p) = __builtin_convertvector(f, f16x4_t);
}
// 8 elements = one 16-byte access of 16-bit storage: what the HBM-bound kernels on 16-bit storage move per lane (the 4-element
// accessors above are 8-byte accesses: half the bytes in flight per instruction, and a 32-channel slab is only half a 128-byte line)
struct f8v { float v[8]; };
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f8v ld8(const bf16_t* p) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  f8v r;
  r.v[0] = __builtin_bit_cast(float, u.x << 16); r.v[1] = __builtin_bit_cast(float, u.x & 0xFFFF0000u);
  r.v[2] = __builtin_bit_cast(float, u.y << 16); r.v[3] = __builtin_bit_cast(float, u.y & 0xFFFF0000u);
  r.v[4] = __builtin_bit_cast(float, u.z << 16); r.v[5] = __builtin_bit_cast(float, u.z & 0xFFFF0000u);
  r.v[6] = __builtin_bit_cast(float, u.w << 16); r.v[7] = __builtin_bit_cast(float, u.w & 0xFFFF0000u);
  return r;
}
__device__ __forceinline__ f8v ld8(const f16_t* p) {
  const f16x8_t h = *reinterpret_cast<const f16x8_t*>(p);
  const f32x8_t f = __builtin_convertvector(h, f32x8_t);
  f8v r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r.v[e] = f[e];
  return r;
}
__device__ __forceinline__ void st8(bf16_t* p, const f8v& x) {
  const f32x8_t f = {x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7]};
  *reinterpret_cast<bf16x8_t*>(p) = __builtin_convertvector(f, bf16x8_t);
}
__device__ __forceinline__ void st8(f16_t* p, const f8v& x) {
  const f32x8_t f = {x.v[0], x.v[1], x.v[2], x.v[3], x.v[4], x.v[5], x.v[6], x.v[7]};
  *reinterpret_cast<f16x8_t*>(p) = __builtin_convertvector(f, f16x8_t);
}
__device__ __forceinline__ float to_f32(bf16_t x) { return static_cast<float>(x); }
__device__ __forceinline__ float to_f32(f16_t x) { return static_cast<float>(x); }
// 4-element accesses of T need 4 * sizeof(T) alignment
template <typename T> inline bool aligned_vec4(const T* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(T) - 1)) == 0; }

// Opt a kernel in to more than 64 KiB of dynamic LDS once PER DEVICE (the attribute is per device; a process that drives
// several GPUs must not rely on a per-process flag).  A benign race at worst sets the attribute twice.
#define DS_RAISE_DYNAMIC_LDS(fn, bytes)                                                                       \
  do {                                                                                                        \
    static unsigned long long ds_raised_mask_ = 0;                                                            \
    int ds_dev_ = 0;                                                                                          \
    (void)hipGetDevice(&ds_dev_);                                                                             \
    const unsigned long long ds_bit_ = 1ull << (ds_dev_ & 63);                                                \
    if (!(ds_raised_mask_ & ds_bit_)) {                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
      ds_raised_mask_ |= ds_bit_;                                                                             \
    }                                                                                                         \
  } while (0)

// Run `CALL(T)` with T = the storage type named by a DIFFSAL_F32 / BF16 / F16 code.
#define DS_DTYPE_DISPATCH(dtype, what, CALL)                                                      \
  do {                                                                                            \
    if ((dtype) == DIFFSAL_F32) { CALL(float); }                                                  \
    else if ((dtype) == DIFFSAL_BF16) { CALL(diffsal::bf16_t); }                                  \
    else if ((dtype) == DIFFSAL_F16) { CALL(diffsal::f16_t); }                                    \
    else { diffsal::set_error("%s: dtype %d (DIFFSAL_F32, DIFFSAL_BF16 or DIFFSAL_F16)", what, (dtype)); return DIFFSAL_E_ARG; } \
  } while (0)

// Source coordinate of a bilinear resize with align_corners=False (PyTorch area_pixel_compute_source_index).
__device__ __forceinline__ void bilin_coord(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = (static_cast<float>(dst) + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = static_cast<int>(s);
  i0 = i0 > in_size - 1 ? in_size - 1 : i0;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - static_cast<float>(i0);
}

}  // namespace diffsal
