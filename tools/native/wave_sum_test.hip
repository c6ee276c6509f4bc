// Bitwise check of wave_sum (common.h) against the __shfl_xor butterfly on random data.  Build: hipcc --offload-arch=gfx950
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../styler_amd/csrc/common.h"
const uint64_t* g_styler_drop_epoch = nullptr;
__global__ void k(const float* in, float* a, float* b) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  float v = in[i], w = v;
  a[i] = wave_sum(v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
  b[i] = w;
}
int main() {
  const int n = 64 * 1024;
  std::vector<float> h(n);
  srand(1);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *d, *a, *b;
  hipMalloc(&d, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, d, a, b);
  std::vector<float> ha(n), hb(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (memcmp(&ha[i], &hb[i], 4)) { if (bad < 5) printf("lane %d wave %d: %.9g vs %.9g\n", i % 64, i / 64, ha[i], hb[i]); ++bad; }
  printf("wave_sum mismatches: %d of %d\n", bad, n);
  return bad != 0;
}
