// Backward companions of the GEMM / implicit-GEMM conv engine.
//
//   act_bwd    : dz = mask(dy) * act'(y)                                  (HBM-bound, one pass)
//   wgrad      : dw[n, c, j] += sum_{b,t} dz[b,t,n] * x[b, t+j-pad, c], db[n] += sum dz   (MFMA "TN" GEMM, K = B*L)
//   colsum     : out[n]   += sum_{b,t} dz[b,t,n]
//   repack_bwd : conv weight [n, cin, kw] -> [cin, kw*n] with taps flipped, the weight of the dX conv:
//                dx = conv_same(dz, w_flipped^T), which runs on the forward engine (styler_conv_gemm).
//
// wgrad uses the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): both operands are consumed straight from
// their row-major [k = time][feature] LDS images (A[i][k]: lane (i, h) reads row k = 2*kk + h), so no
// transposition of dz or x is ever materialised.  dw is addressed with explicit strides so the result
// lands directly in the PARAMETER layout ([n, cin, kw] for conv taps, [n, cin] for Linear).
#include "common.h"

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                      const float* __restrict__ y, int64_t ldy,
                                                      float* __restrict__ dz, int64_t lddz, int64_t rows, int L, int C,
                                                      int act, const int64_t* __restrict__ len) {
  const int nq = C / 4;
  const int64_t total = rows * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    float4 g = *reinterpret_cast<const float4*>(dy + row * lddy + q * 4);
    if (len) {
      const int64_t b = row / L;
      if ((row - b * L) >= len[b]) g = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (act != STYLER_ACT_NONE) {
      const float4 v = *reinterpret_cast<const float4*>(y + row * ldy + q * 4);
      if (act == STYLER_ACT_RELU) {
        g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
      } else if (act == STYLER_ACT_TANH) {
        g.x *= 1.f - v.x * v.x; g.y *= 1.f - v.y * v.y; g.z *= 1.f - v.z * v.z; g.w *= 1.f - v.w * v.w;
      }
    }
    *reinterpret_cast<float4*>(dz + row * lddz + q * 4) = g;
  }
}

extern "C" int styler_act_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dz, int64_t lddz,
                              int B, int L, int C, int act, const int64_t* len, void* stream) {
  if (!dy || !dz || B <= 0 || L <= 0 || C <= 0 || (C & 3) || (act != STYLER_ACT_NONE && !y)) return STYLER_EINVAL;
  if ((lddy & 3) || (lddz & 3) || (y && (ldy & 3))) return STYLER_EALIGN;
  const int64_t rows = (int64_t)B * L;
  int64_t blocks = (rows * (C / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, dz,
                     lddz, rows, L, C, act, len);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// wgrad: all KW taps of a conv weight (or a Linear, KW = 1) in ONE launch.
//   dw[nn*sn + c*sc + j*sj] += sum_{b,t} dz[b,t,nn] * x[b, t + j - pad_left, c]      (x = 0 outside the item)
//   db[nn]                  += sum_{b,t} dz[b,t,nn]                                   (optional, fused)
// Block tile 64 (n) x 64 (c), 4 waves as 2x2 (one 32x32 MFMA tile each) with KW accumulators per wave;
// K chunk = 32 rows of the flattened [B*L] axis.  The dz chunk [32][64] and the haloed x chunk
// [32 + KW - 1][64] are staged once per chunk; tap j reads x rows shifted by j.  Rows whose shifted
// time index leaves the item are zeroed by a per-row tap mask (LDS).  Both operands are consumed from their
// row-major images (A[i][k]: lane (i, h) reads row k = 2*kk + h), no transposition.  The row axis is split over
// blockIdx.y; partial tiles are combined with fp32 atomics straight into the parameter-layout gradient.
#define WG_BK 32
#define WG_LD 68            // LDS row stride (floats)

template <int KW>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ dz, int64_t lddz,
                                                    const float* __restrict__ x, int64_t ldx, float* __restrict__ dw,
                                                    float* __restrict__ db, int64_t sn, int64_t sc, int64_t sj, int B,
                                                    int L, int n, int cin, int pad_left, int ct, int chunks_per_split) {
  constexpr int XR = WG_BK + KW - 1;                 // x rows per chunk (with halo)
  __shared__ __attribute__((aligned(16))) float sA[2][WG_BK * WG_LD];
  __shared__ __attribute__((aligned(16))) float sB[2][XR * WG_LD];
  __shared__ uint32_t sMask[2][WG_BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x;
  const int n0 = (tile / ct) * 64, c0 = (tile % ct) * 64;
  const int64_t M = (int64_t)B * L;
  const int64_t nchunks = (M + WG_BK - 1) / WG_BK;
  const int64_t ch0 = (int64_t)blockIdx.y * chunks_per_split;
  int64_t ch1 = ch0 + chunks_per_split; if (ch1 > nchunks) ch1 = nchunks;
  if (ch0 >= ch1) return;

  // staging: a row is 64 floats = 16 float4; 256 threads cover 16 rows per pass
  const int sr = tid >> 4, sq = (tid & 15) * 4;
  constexpr int XP = (XR + 15) / 16;
  float4 ra[2], rb[XP];
  uint32_t rmask = 0;
  auto load = [&](int64_t ch) {
    const int64_t mb = ch * WG_BK;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int64_t m = mb + sr + p * 16;
      ra[p] = (m < M && n0 + sq < n) ? *reinterpret_cast<const float4*>(dz + m * lddz + n0 + sq)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int r = sr + p * 16;
      const int64_t m = mb - pad_left + r;
      rb[p] = (r < XR && m >= 0 && m < M && c0 + sq < ((cin + 3) & ~3)) ? *reinterpret_cast<const float4*>(x + m * ldx + c0 + sq)
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < WG_BK) {                               // tap validity of output row m = mb + tid
      const int64_t m = mb + tid;
      uint32_t bits = 0;
      if (m < M) {
        const int t = (int)(m % L);
#pragma unroll
        for (int j = 0; j < KW; ++j) {
          const int tt = t + j - pad_left;
          if (tt >= 0 && tt < L) bits |= 1u << j;
        }
      }
      rmask = bits;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<float4*>(&sA[buf][(sr + p * 16) * WG_LD + sq]) = ra[p];
#pragma unroll
    for (int p = 0; p < XP; ++p)
      if (sr + p * 16 < XR) *reinterpret_cast<float4*>(&sB[buf][(sr + p * 16) * WG_LD + sq]) = rb[p];
    if (tid < WG_BK) sMask[buf][tid] = rmask;
  };

  f32x16 acc[KW];
#pragma unroll
  for (int j = 0; j < KW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const bool use_mask = KW > 1 || pad_left != 0;     // a shifted Linear (LSTM W_hh) also crosses item boundaries
  float bsum = 0.f;                                  // bias partial (threads < 64 of c-tile 0)
  const bool do_bias = db && (tile % ct) == 0 && tid < 64;

  load(ch0);
  store(0);
  __syncthreads();
  int buf = 0;
  for (int64_t ch = ch0; ch < ch1; ++ch) {
    const bool more = ch + 1 < ch1;
    if (more) load(ch + 1);
    const float* pa = &sA[buf][lh * WG_LD + wm * 32 + li];
    const float* pb = &sB[buf][lh * WG_LD + wn * 32 + li];
#pragma unroll
    for (int kk = 0; kk < WG_BK / 2; ++kk) {
      const float a = pa[kk * 2 * WG_LD];
      const uint32_t mk = use_mask ? sMask[buf][kk * 2 + lh] : 0xffffffffu;
#pragma unroll
      for (int j = 0; j < KW; ++j) {
        float bv = pb[(kk * 2 + j) * WG_LD];
        bv = ((mk >> j) & 1u) ? bv : 0.f;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
      }
    }
    if (do_bias) {
#pragma unroll 8
      for (int k = 0; k < WG_BK; ++k) bsum += sA[buf][k * WG_LD + tid];
    }
    if (more) store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // C layout: col (= c) = lane&31, row (= n) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int c = c0 + wn * 32 + li;
  if (c < cin) {
#pragma unroll
    for (int j = 0; j < KW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (nn < n) atomicAdd(dw + nn * sn + c * sc + j * sj, acc[j][r]);
      }
  }
  if (do_bias && n0 + tid < n) atomicAdd(db + n0 + tid, bsum);
}

extern "C" int styler_wgrad(const float* dz, int64_t lddz, const float* x, int64_t ldx, float* dw, float* db,
                            int64_t stride_n, int64_t stride_c, int64_t stride_j, int B, int L, int n, int cin, int kw,
                            int pad_left, void* stream) {
  if (!dz || !x || !dw || B <= 0 || L <= 0 || n <= 0 || cin <= 0) return STYLER_EINVAL;
  if (kw != 1 && kw != 3 && kw != 5 && kw != 9) return STYLER_EINVAL;
  if ((lddz & 3) || (ldx & 3) || (n & 3) || ldx < ((cin + 3) & ~3) || ((uintptr_t)dz & 15) || ((uintptr_t)x & 15)) return STYLER_EALIGN;
  const int nt = (n + 63) / 64, ct = (cin + 63) / 64;
  const int64_t M = (int64_t)B * L;
  const int64_t nchunks = (M + WG_BK - 1) / WG_BK;
  int64_t splits = (512 + nt * ct - 1) / (nt * ct);
  if (splits > nchunks / 8) splits = nchunks / 8;
  if (splits < 1) splits = 1;
  const int cps = (int)((nchunks + splits - 1) / splits);
  splits = (nchunks + cps - 1) / cps;
  const dim3 grid(nt * ct, (unsigned)splits);
  hipStream_t st = (hipStream_t)stream;
#define WG_LAUNCH(K) hipLaunchKernelGGL(wgrad_kernel<K>, grid, dim3(256), 0, st, dz, lddz, x, ldx, dw, db, stride_n, \
                                        stride_c, stride_j, B, L, n, cin, pad_left, ct, cps)
  if (kw == 1) WG_LAUNCH(1); else if (kw == 3) WG_LAUNCH(3); else if (kw == 5) WG_LAUNCH(5); else WG_LAUNCH(9);
#undef WG_LAUNCH
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// colsum: block = 64 rows-chunk x all columns (thread per float4 column group, loops rows), atomics at the end
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dz, int64_t lddz,
                                                     float* __restrict__ out, float* __restrict__ out2, int64_t rows,
                                                     int C, int rows_per_block) {
  const int nq = C / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
  for (int q = threadIdx.x; q < nq; q += blockDim.x) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = r0; r < r1; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(dz + r * lddz + q * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    atomicAdd(out + q * 4 + 0, s.x); atomicAdd(out + q * 4 + 1, s.y);
    atomicAdd(out + q * 4 + 2, s.z); atomicAdd(out + q * 4 + 3, s.w);
    if (out2) {
      atomicAdd(out2 + q * 4 + 0, s.x); atomicAdd(out2 + q * 4 + 1, s.y);
      atomicAdd(out2 + q * 4 + 2, s.z); atomicAdd(out2 + q * 4 + 3, s.w);
    }
  }
}

extern "C" int styler_colsum(const float* dz, int64_t lddz, float* out, float* out2, int64_t rows, int C, void* stream) {
  if (!dz || !out || rows <= 0 || C <= 0 || (C & 3) || (lddz & 3)) return STYLER_EINVAL;
  int rpb = 128;
  if (rows / rpb > 2048) rpb = (int)((rows + 2047) / 2048);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(C / 4 >= 256 ? 256 : 64), 0,
                     (hipStream_t)stream, dz, lddz, out, out2, rows, C, rpb);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// dst[c, j', nn] = src[nn, c, kw-1-j']     (src = parameter layout [n, cin, kw]; kw = 1: plain transpose)
__global__ void repack_bwd_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int cin, int kw) {
  const int64_t total = (int64_t)n * cin * kw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int nn = (int)(i % n); const int j = (int)((i / n) % kw); const int64_t c = i / ((int64_t)n * kw);
    dst[i] = src[((int64_t)nn * cin + c) * kw + (kw - 1 - j)];
  }
}

extern "C" int styler_repack_weight_bwd(const float* src, float* dst, int n, int cin, int kw, void* stream) {
  if (!src || !dst || n <= 0 || cin <= 0 || kw <= 0) return STYLER_EINVAL;
  const int64_t total = (int64_t)n * cin * kw;
  int64_t blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(repack_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n, cin, kw);
  return launch_status();
}
