// Spike (not part of the product): how fast is an fp32-in / fp32-out GEMM on gfx950 when every fp32 operand is split into
// 2 or 3 bf16 terms on the fly and multiplied on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate)?
//   mode 0: exact fp32, v_mfma_f32_32x32x2_f32                                  (what diffsal::igemm_kernel uses)
//   mode 1: x = hi + lo,          hi*hi + hi*lo + lo*hi             (3 MFMAs, ~2^-16 relative per product)
//   mode 2: x = hi + mid + lo,    6 cross terms down to 2^-24       (6 MFMAs, ~fp32 accuracy)
// C[M][N] = A[M][K] * B[N][K]^T, both K-contiguous (the operand form of the implicit-GEMM kernel, without the im2col
// gather).  128x128 tile per workgroup, 4 waves (2x2), 32-wide K slices through LDS (pitch 36), one stage -- the same
// simple loop for all modes; `resident` = 1 skips the global loads after the first slice (LDS/convert/MFMA ceiling).
// Build: hipcc --offload-arch=gfx950 -O3 -o bf16x3_gemm bf16x3_gemm.hip ; run: ./bf16x3_gemm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32, PITCH = 36;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int NT>
__device__ __forceinline__ void split(const float (&x)[8], bf16x8 (&t)[NT]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float r = x[i];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const __bf16 h = static_cast<__bf16>(r);
      t[q][i] = h;
      r -= static_cast<float>(h);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ C, int M, int N, int K, int resident) {
  __shared__ __attribute__((aligned(16))) float As[BM * PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[BN * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int row = lane & 31, half = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int k0 = 0; k0 < K; k0 += BK) {
    if (!resident || k0 == 0) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {               // 128 rows x 8 float4 = 1024 float4 per operand
        const int idx = tid + 256 * q, r = idx >> 3, c4 = (idx & 7) * 4;
        *reinterpret_cast<float4*>(&As[r * PITCH + c4]) = ld4(A + static_cast<long>(m0 + r) * K + k0 + c4);
        *reinterpret_cast<float4*>(&Bs[r * PITCH + c4]) = ld4(B + static_cast<long>(n0 + r) * K + k0 + c4);
      }
      __syncthreads();
    }
    if constexpr (MODE == 0) {
      // lane reads 4 consecutive k (b128) per quarter slice; MFMA k pairs {j, j+4} within each group of 8 (order is free)
#pragma unroll
      for (int g = 0; g < 4; ++g) {               // 4 groups of 8 k
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(&As[((wm * 2 + i) * 32 + row) * PITCH + g * 8 + half * 4]);
          a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
          const float4 w = *reinterpret_cast<const float4*>(&Bs[((wn * 2 + i) * 32 + row) * PITCH + g * 8 + half * 4]);
          b[i][0] = w.x; b[i][1] = w.y; b[i][2] = w.z; b[i][3] = w.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
      }
    } else {
      constexpr int NT = MODE == 1 ? 2 : 3;
#pragma unroll
      for (int g = 0; g < 2; ++g) {               // 2 MFMA k-steps of 16: lane half h owns k = g*16 + h*8 .. +7
        bf16x8 at[2][NT], bt[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float xa[8], xb[8];
          const float* pa = &As[((wm * 2 + i) * 32 + row) * PITCH + g * 16 + half * 8];
          const float* pb = &Bs[((wn * 2 + i) * 32 + row) * PITCH + g * 16 + half * 8];
          const float4 a0 = *reinterpret_cast<const float4*>(pa), a1 = *reinterpret_cast<const float4*>(pa + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(pb), b1 = *reinterpret_cast<const float4*>(pb + 4);
          xa[0] = a0.x; xa[1] = a0.y; xa[2] = a0.z; xa[3] = a0.w; xa[4] = a1.x; xa[5] = a1.y; xa[6] = a1.z; xa[7] = a1.w;
          xb[0] = b0.x; xb[1] = b0.y; xb[2] = b0.z; xb[3] = b0.w; xb[4] = b1.x; xb[5] = b1.y; xb[6] = b1.z; xb[7] = b1.w;
          split<NT>(xa, at[i]);
          split<NT>(xb, bt[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            // smallest terms first
            if constexpr (NT == 3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[i][0], bt[j][2], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[i][2], bt[j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[i][1], bt[j][1], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[i][0], bt[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[i][1], bt[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[i][0], bt[j][0], acc[i][j], 0, 0, 0);
          }
      }
    }
  }
  // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int n = n0 + (wn * 2 + j) * 32 + row;
        C[static_cast<long>(m) * N + n] = acc[i][j][r];
      }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
float run(const float* dA, const float* dB, float* dC, int M, int N, int K, int resident, int iters) {
  dim3 grid(M / BM, N / BN);
  hipLaunchKernelGGL(gemm_kernel<MODE>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K, resident);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_kernel<MODE>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K, resident);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int M = 193536, N = 128, K = 1728;      // stage-3 UpEmbed conv as a plain GEMM (N padded to the tile)
  std::vector<float> hA(static_cast<size_t>(M) * K), hB(static_cast<size_t>(N) * K);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (static_cast<int>(s >> 9) - (1 << 22)) / static_cast<float>(1 << 22); };
  for (auto& v : hA) v = rnd() * (rnd() > 0 ? 1.f : 0.f);    // ~half zeros, like post-ReLU activations
  for (auto& v : hB) v = rnd() * 0.05f;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, static_cast<size_t>(M) * N * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  // reference: a few rows in fp64
  const int rows[4] = {0, 777, 100001, M - 1};
  std::vector<double> ref(4 * N);
  for (int q = 0; q < 4; ++q)
    for (int n = 0; n < N; ++n) {
      double a = 0;
      for (int k = 0; k < K; ++k) a += static_cast<double>(hA[static_cast<size_t>(rows[q]) * K + k]) * hB[static_cast<size_t>(n) * K + k];
      ref[q * N + n] = a;
    }
  const double gf = 2.0 * M * N * K / 1e9;
  std::vector<float> hC(N);
  const char* names[3] = {"fp32 mfma 32x32x2   ", "bf16x3 (hi,lo)      ", "bf16x6 (hi,mid,lo)  "};
  for (int resident = 0; resident < 2; ++resident)
    for (int mode = 0; mode < 3; ++mode) {
      float ms = mode == 0 ? run<0>(dA, dB, dC, M, N, K, resident, 10) : mode == 1 ? run<1>(dA, dB, dC, M, N, K, resident, 10)
                                                                                    : run<2>(dA, dB, dC, M, N, K, resident, 10);
      double err = 0, mag = 0;
      if (!resident)
        for (int q = 0; q < 4; ++q) {
          CK(hipMemcpy(hC.data(), dC + static_cast<size_t>(rows[q]) * N, N * 4, hipMemcpyDeviceToHost));
          for (int n = 0; n < N; ++n) { err = fmax(err, fabs(hC[n] - ref[q * N + n])); mag = fmax(mag, fabs(ref[q * N + n])); }
        }
      printf("%s %s  %8.3f ms  %7.1f TFLOP/s(fp32-equivalent)  max err / max |ref| = %.2e\n", names[mode],
             resident ? "LDS-resident operands" : "streaming from HBM    ", ms, gf / ms, resident ? 0.0 : err / mag);
    }
  return 0;
}
