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
      else if (mode == 2) {
        const float x = rv[c];
        d = 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
      } else d = rv[c] * (1.0f - rv[c]);
      gv[c] *= d;
    }
    st4(dx + i * 4, make_float4(gv[0], gv[1], gv[2], gv[3]));
  }
}

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
