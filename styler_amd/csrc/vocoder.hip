// Elementwise glue of the HiFi-GAN generator (hifigan/models.py:96-100,155-167).  Every convolution of the vocoder --
// Conv1d, dilated Conv1d, ConvTranspose1d -- runs on the implicit-GEMM engine (styler_conv_gemm_pad, gemm_conv.hip);
// what is left is the pre-activation  leaky_relu(x)  in front of each conv (the un-activated x is the residual, so it
// cannot live in a producer epilogue) and the resblock average  (r0 + r1 + r2) / num_kernels,  which is always followed
// by a leaky_relu and is fused with it here.  HBM-bound: 4 B read per input + 4 B written per element.
#include "common.h"

__global__ __launch_bounds__(256) void leaky_sum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ c, float* __restrict__ y,
                                                        int64_t count, float scale, float slope) {
  const int64_t nq = count >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
    float4 v = reinterpret_cast<const float4*>(a)[i];
    if (b) { const float4 w = reinterpret_cast<const float4*>(b)[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    if (c) { const float4 w = reinterpret_cast<const float4*>(c)[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    v.x = v.x > 0.f ? v.x : slope * v.x; v.y = v.y > 0.f ? v.y : slope * v.y;
    v.z = v.z > 0.f ? v.z : slope * v.z; v.w = v.w > 0.f ? v.w : slope * v.w;
    reinterpret_cast<float4*>(y)[i] = v;
  }
  // tail (count not a multiple of 4)
  for (int64_t i = (nq << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    float v = a[i];
    if (b) v += b[i];
    if (c) v += c[i];
    v *= scale;
    y[i] = v > 0.f ? v : slope * v;
  }
}

extern "C" int styler_leaky_sum(const float* a, const float* b, const float* c, float* y, int64_t count, float scale,
                                float slope, void* stream) {
  if (!a || !y || count <= 0) return STYLER_EINVAL;
  if (((uintptr_t)a | (uintptr_t)y | (uintptr_t)b | (uintptr_t)c) & 15) return STYLER_EALIGN;
  int64_t blocks = ((count >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
  hipLaunchKernelGGL(leaky_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, y, count, scale,
                     slope);
  return launch_status();
}
