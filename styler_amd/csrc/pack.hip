// Packed-rows layout for the decoder.
//
// Every FFT block of the decoder (transformer/Models.py:111-135) zeroes its padded rows (Layers.py:29,32), masks padded
// keys in attention and convolves over zeros there: on the padded rectangle [B, T] a third of the rows (VCTK shapes:
// 36 %) is arithmetic on zeros.  The decoder therefore runs on the valid frames only, stored back to back:
//   packed[cu[b] + t] = padded[b, t]   for t < len[b]
// with capacity B*T rows (the number of valid rows stays on the device: no host sync, hipGraph-safe).
// styler_pack_plan derives, from the int64 lengths, everything the packed kernels index with:
//   cu       int32 [B+1]   first packed row of every item
//   rowinfo  int2  [B*T]   (t, len-1-t) of every packed row, (0,-1) behind the data -> conv tap validity
//   chunktab int4  [..]    (first row, t0, len, b) of every 64-row K chunk of the weight-gradient GEMM (chunks never
//                          straddle items)
//   counts   int64 [2]     number of packed rows (doubles as the `len` of the single packed "item"), number of chunks
#include "common.h"

static inline unsigned pack_grid(int64_t work) {
  int64_t b = (work + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// Round 6: 32 blocks instead of one (the kernel sits on the step's serial chain in front of the decoder: 22 us for 340 KB of index
// tables -- one thread walking len[] with a global load per item, then one block writing every row).  Every block loads the
// lengths into LDS with all its threads, scans them there (B <= 4096) and writes its slice of the rows; block 0 also writes cu / counts.
__global__ __launch_bounds__(1024) void pack_plan_kernel(const int64_t* __restrict__ len, int B, int T,
                                                         int* __restrict__ cu, int2* __restrict__ rowinfo,
                                                         int4* __restrict__ chunktab, int64_t* __restrict__ counts) {
  __shared__ int s_cu[4097], s_cc[4097];
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    int l = (int)len[b];
    s_cu[b] = l < 0 ? 0 : (l > T ? T : l);               // (the clamped length for now)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int r = 0, c = 0;
    for (int b = 0; b < B; ++b) {
      const int l = s_cu[b];
      s_cu[b] = r; s_cc[b] = c;
      r += l; c += (l + 63) / 64;
    }
    s_cu[B] = r; s_cc[B] = c;
    if (blockIdx.x == 0) { counts[0] = r; counts[1] = c; }
  }
  __syncthreads();
  if (blockIdx.x == 0)
    for (int b = threadIdx.x; b <= B; b += blockDim.x) cu[b] = s_cu[b];
  const int total = B * T;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / T, t = i - b * T;
    const int l = s_cu[b + 1] - s_cu[b];
    if (t < l) rowinfo[s_cu[b] + t] = make_int2(t, l - 1 - t);
    if (i >= s_cu[B]) rowinfo[i] = make_int2(0, -1);
    if ((t & 63) == 0 && t < l) chunktab[s_cc[b] + (t >> 6)] = make_int4(s_cu[b] + t, t, l, b);
  }
}

extern "C" int styler_pack_plan(const int64_t* len, int B, int T, int32_t* cu, int32_t* rowinfo, int32_t* chunktab,
                                int64_t* counts, void* stream) {
  if (!len || !cu || !rowinfo || !chunktab || !counts || B <= 0 || B > 4096 || T <= 0) return STYLER_EINVAL;
  const int64_t total = (int64_t)B * T;
  const unsigned blocks = (unsigned)(total >= 32 * 1024 ? 32 : (total + 1023) / 1024);
  hipLaunchKernelGGL(pack_plan_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, len, B, T, cu,
                     reinterpret_cast<int2*>(rowinfo), reinterpret_cast<int4*>(chunktab), counts);
  return launch_status();
}

// dir 0: packed[cu[b] + t] = padded[b, t] (+ add[t], the positional table)     for t < len[b]
// dir 1: padded[b, t] = t < len[b] ? packed[cu[b] + t] : 0
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst,
                                                        int64_t ldd, const float* __restrict__ add,
                                                        const int* __restrict__ cu, int B, int T, int C, int dir, int io) {
  const bool pad16 = io & STYLER_IO_X_BF16, pk16 = io & STYLER_IO_Y_BF16;      // storage of the padded / the packed tensor
  const int nq = C / 4;
  const int64_t total = (int64_t)B * T * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    const int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
    const int r0 = cu[b], l = cu[b + 1] - r0;
    if (dir == 0) {
      if (t >= l) continue;
      float4 v = ldg4(src, row * lds + q * 4, pad16);
      if (add) {
        const float4 p = *reinterpret_cast<const float4*>(add + (int64_t)t * C + q * 4);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
      }
      stg4(dst, (int64_t)(r0 + t) * ldd + q * 4, v, pk16);
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < l) v = ldg4(src, (int64_t)(r0 + t) * lds + q * 4, pk16);
      stg4(dst, row * ldd + q * 4, v, pad16);
    }
  }
}

extern "C" int styler_pack_rows(const float* padded, int64_t ldp, float* packed, int64_t ldk, const float* add,
                                const int32_t* cu, int B, int T, int C, int io_flags, void* stream) {
  if (!padded || !packed || !cu || B <= 0 || T <= 0 || C <= 0 || (C & 3) || (ldp & 3) || (ldk & 3)) return STYLER_EINVAL;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(pack_grid((int64_t)B * T * (C / 4))), dim3(256), 0, (hipStream_t)stream, padded,
                     ldp, packed, ldk, add, cu, B, T, C, 0, io_flags);
  return launch_status();
}

extern "C" int styler_unpack_rows(const float* packed, int64_t ldk, float* padded, int64_t ldp, const int32_t* cu, int B,
                                  int T, int C, int io_flags, void* stream) {
  if (!padded || !packed || !cu || B <= 0 || T <= 0 || C <= 0 || (C & 3) || (ldp & 3) || (ldk & 3)) return STYLER_EINVAL;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(pack_grid((int64_t)B * T * (C / 4))), dim3(256), 0, (hipStream_t)stream, packed,
                     ldk, padded, ldp, nullptr, cu, B, T, C, 1, io_flags);
  return launch_status();
}
