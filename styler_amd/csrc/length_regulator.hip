// LengthRegulator (modules.py:396-423 + utils.pad, utils.py:332-348) as two HBM-bound kernels.
//
//  scan  : one block per item; durations -> inclusive prefix sums with a wave64 shuffle scan
//          (+ cross-wave carry through LDS).  Fuses the free-running rounding
//          d = max(rint(exp(log_d) - 1) * d_control, 0) (modules.py:357-358) and the int() truncation
//          of LengthRegulator.expand (modules.py:415-416), so frame indices are bit-exact.
//  expand: grid (frame tiles, item); the item's prefix sums are staged in LDS, each wave owns frames,
//          does a uniform binary search (upper bound of t) and streams the C-wide row with float4:
//          write 4*C B per output frame (incl. zero fill of padded frames), read 4*C B per phoneme
//          (L2-resident re-reads).
#include "common.h"

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}

__global__ __launch_bounds__(256) void duration_scan_kernel(const void* __restrict__ dur, int dur_is_float,
                                                            const float* __restrict__ log_d, float d_control,
                                                            float* __restrict__ dur_out, int32_t* __restrict__ csum,
                                                            int64_t* __restrict__ mel_len, int S) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int s0 = 0; s0 < S; s0 += 256) {
    const int s = s0 + tid;
    int d = 0;
    if (s < S) {
      const int64_t i = (int64_t)b * S + s;
      if (log_d) {
        float f = fmaxf(rintf(expf(log_d[i]) - 1.0f) * d_control, 0.f);
        if (dur_out) dur_out[i] = f;
        d = (int)f;                                   // int() truncation (modules.py:416)
      } else if (dur_is_float) {
        d = (int)reinterpret_cast<const float*>(dur)[i];
      } else {
        d = (int)reinterpret_cast<const int64_t*>(dur)[i];
      }
      if (d < 0) d = 0;
    }
    int inc = wave_incl_scan(d, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int off = carry_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    inc += off;
    if (s < S) csum[(int64_t)b * S + s] = inc;
    __syncthreads();
    if (tid == 255) carry_s = inc;
    __syncthreads();
  }
  if (tid == 0) mel_len[b] = (int64_t)carry_s;
}

extern "C" int styler_duration_scan(const void* dur, int dur_is_float, const float* log_d, float d_control,
                                    float* dur_out, int32_t* csum, int64_t* mel_len, int B, int S, void* stream) {
  if ((!dur && !log_d) || !csum || !mel_len || B <= 0 || S <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(duration_scan_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dur, dur_is_float, log_d,
                     d_control, dur_out, csum, mel_len, S);
  return launch_status();
}

#define LR_TT 32          // frames per block
#define LR_SMAX 4096      // prefix sums staged in LDS (16 KiB); longer items search global memory

__global__ __launch_bounds__(256) void length_regulate_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const int32_t* __restrict__ csum,
                                                              float* __restrict__ out, int64_t ldo,
                                                              int32_t* __restrict__ frame_idx, int S, int T, int C) {
  __shared__ int32_t cs[LR_SMAX];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* gcs = csum + (int64_t)b * S;
  const bool in_lds = S <= LR_SMAX;
  if (in_lds) for (int i = tid; i < S; i += 256) cs[i] = gcs[i];
  __syncthreads();
  const int32_t* c = in_lds ? cs : gcs;
  const int total = c[S - 1];
  const int t0 = blockIdx.x * LR_TT;
  // A wave owns 8 consecutive frames.  Their source rows are found by 8 lanes in parallel (one binary search each, not
  // the same search repeated by all 64 lanes frame after frame) and broadcast; the copy then keeps the 8 frames' loads of
  // a 1 KB column chunk in flight together before the 8 stores (frame by frame, every 16-byte load was a dependent round
  // trip in front of its store: 3.2 TB/s).
  constexpr int FPW = LR_TT / 4;                      // frames per wave
  const int tw = t0 + wave * FPW;
  int my_idx = -1;
  if (lane < FPW) {
    const int t = tw + lane;
    if (t < T && t < total) {                         // upper bound: first i with csum[i] > t
      int lo = 0, hi = S - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (c[mid] > t) hi = mid; else lo = mid + 1; }
      my_idx = lo;
    }
    if (frame_idx && t < T) frame_idx[(int64_t)b * T + t] = my_idx;
  }
  int idx[FPW];
#pragma unroll
  for (int k = 0; k < FPW; ++k) idx[k] = __shfl(my_idx, k, 64);
  const float* xb = x + (int64_t)b * S * ldx;
  float* ob = out + ((int64_t)b * T + tw) * ldo;
  for (int q = lane * 4; q < C; q += 256) {
    float4 v[FPW];
#pragma unroll
    for (int k = 0; k < FPW; ++k)
      v[k] = idx[k] >= 0 ? *reinterpret_cast<const float4*>(xb + (int64_t)idx[k] * ldx + q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < FPW; ++k)
      if (tw + k < T) *reinterpret_cast<float4*>(ob + (int64_t)k * ldo + q) = v[k];
  }
}

extern "C" int styler_length_regulate(const float* x, int64_t ldx, const int32_t* csum, float* out, int64_t ldo,
                                      int32_t* frame_idx, int B, int S, int T, int C, void* stream) {
  if (!x || !csum || !out || B <= 0 || S <= 0 || T <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  if ((ldx & 3) || (ldo & 3)) return STYLER_EALIGN;
  hipLaunchKernelGGL(length_regulate_kernel, dim3((T + LR_TT - 1) / LR_TT, B), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, csum, out, ldo, frame_idx, S, T, C);
  return launch_status();
}
