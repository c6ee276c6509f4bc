// Normalisation kernels: fused residual + LayerNorm(256) + pad mask (+ optional Linear(256,1) tail),
// GroupNorm(16 ch / group, stats over the padded time axis) + ReLU, BatchNorm1d eval fold and train.
// All HBM-bound: one read + one write of the activation (GroupNorm re-reads once from L2).
#include <cstdlib>
#include "common.h"

// one wave per row; lane holds 4 consecutive channels (64 x 4 = 256)
__global__ __launch_bounds__(256) void add_layernorm_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ res, int64_t ldres,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, int64_t ldy,
    const float* __restrict__ dot_w, const float* __restrict__ dot_b, float* __restrict__ dot_out, int64_t rows,
    int L, const int64_t* __restrict__ len, float drop_p, uint64_t drop_seed_host, const uint64_t* __restrict__ epoch,
    float in_drop_p, uint64_t in_drop_seed_host, float* __restrict__ sum_out, int64_t ldsum, uint16_t* __restrict__ y16,
    int64_t ldy16, int io, uint16_t* __restrict__ y3, int y3parts) {
  const bool res16 = io & STYLER_LN_RES_BF16, yb16 = io & STYLER_LN_Y_BF16, sum16 = io & STYLER_LN_SUM_BF16;
  const uint64_t drop_seed = mix_drop_epoch(drop_seed_host, epoch);
  const int lane = threadIdx.x & 63;
  // the row is a wave-uniform scalar (see layernorm_bwd_kernel): row * ld, the item / time split and the length lookup run on
  // the scalar unit; one-item launches (packed rows: L >= rows) need no division, the others a 32-bit one (rows < 2^31)
  const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (row >= rows) return;
  bool masked = false;
  if (len) {
    uint32_t b = 0, t = (uint32_t)row;
    if ((int64_t)L < rows) { b = (uint32_t)row / (uint32_t)L; t = (uint32_t)row - b * (uint32_t)L; }
    masked = (int64_t)t >= len[b];
  }
  if (masked) {
    if (y) stg4(y, row * ldy + lane * 4, make_float4(0.f, 0.f, 0.f, 0.f), yb16);
    if (y16) *reinterpret_cast<uint2*>(y16 + row * ldy16 + lane * 4) = make_uint2(0u, 0u);
    if (y3) x3_store4(y3, row, lane * 4, 256, y3parts, make_float4(0.f, 0.f, 0.f, 0.f));
    if (dot_out && lane == 0) dot_out[row] = 0.f;
    return;                                              // (sum_out is only read back on unmasked rows)
  }
  float4 v = *reinterpret_cast<const float4*>(x + row * ldx + lane * 4);
  if (in_drop_p > 0.f) {                                 // dropout(x) before the residual (SubLayers.py:58,86), same
    const uint2 kd = dropout_key(mix_drop_epoch(in_drop_seed_host, epoch));     // stream as styler_dropout on [rows, 256]
    v = dropout_apply4(v, kd, (uint64_t)row * 256 + lane * 4, dropout_thr16(in_drop_p), 1.f / (1.f - in_drop_p));
  }
  if (res) {
    const float4 r = ldg4(res, row * ldres + lane * 4, res16);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (sum_out) {                                         // the pre-norm sum, for the backward
    stg4(sum_out, row * ldsum + lane * 4, v, sum16);
    // a bf16 sum is what the backward will see: normalise THAT value, so that forward and backward agree on the statistics
    if (sum16) v = raw4_f32(make_uint2(cvt_pk_bf16_rne(v.x, v.y), cvt_pk_bf16_rne(v.z, v.w)));
  }
  const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float4 bt = *reinterpret_cast<const float4*>(beta + lane * 4);
  float4 o = make_float4(dx * rstd * g.x + bt.x, dy * rstd * g.y + bt.y, dz * rstd * g.z + bt.z,
                         dw * rstd * g.w + bt.w);
  if (drop_p > 0.f) {                                    // dropout behind the LayerNorm (train mode): what y and the
    // ... Linear(256,1) tail see (modules.py:436-447)
    o = dropout_apply4(o, dropout_key(drop_seed), (uint64_t)row * 256 + lane * 4, dropout_thr16(drop_p), 1.f / (1.f - drop_p));
  }
  if (y) stg4(y, row * ldy + lane * 4, o, yb16);
  // y16: a second copy of the output as bf16 (round to nearest even) for the GEMMs that consume it -- they round their
  // activation operand to bf16 anyway, so results do not change; the 256 x 256 engine (gemm256.hip) DMAs it straight into LDS
  if (y16) *reinterpret_cast<uint2*>(y16 + row * ldy16 + lane * 4) = make_uint2(cvt_pk_bf16_rne(o.x, o.y), cvt_pk_bf16_rne(o.z, o.w));
  if (y3) x3_store4(y3, row, lane * 4, 256, y3parts, o);     // round 5, bf16x3: the split of the fp32 output (styler_set_x3_out)
  if (dot_out) {
    const float4 w = *reinterpret_cast<const float4*>(dot_w + lane * 4);
    const float d = wave_sum(o.x * w.x + o.y * w.y + o.z * w.z + o.w * w.w);
    if (lane == 0) dot_out[row] = d + dot_b[0];
  }
}

extern "C" int styler_add_layernorm(const float* x, int64_t ldx, const float* res, int64_t ldres,
                                    const float* gamma, const float* beta, float* y, int64_t ldy,
                                    const float* dot_w, const float* dot_b, float* dot_out, int B, int L, int C,
                                    const int64_t* len, float drop_p, uint64_t drop_seed, float in_drop_p,
                                    uint64_t in_drop_seed, float* sum_out, int64_t ldsum, uint16_t* y16, int64_t ldy16,
                                    int io_flags, void* stream) {
  uint16_t* y3 = nullptr;
  int y3parts = 0;
  styler_take_x3_out(&y3, &y3parts);               // (bf16x3: the split of the fp32 output rows, filed by the caller)
  if (y3 && (!y || (io_flags & STYLER_LN_Y_BF16) || ldy != C)) return STYLER_EINVAL;
  if (!x || !gamma || !beta || (!y && !dot_out) || B <= 0 || L <= 0) return STYLER_EINVAL;
  if (y16 && ((ldy16 & 3) || ((uintptr_t)y16 & 7))) return STYLER_EALIGN;
  if (C != 256) return STYLER_EINVAL;
  if (dot_out && (!dot_w || !dot_b)) return STYLER_EINVAL;
  if ((ldx & 3) || (res && (ldres & 3)) || (y && (ldy & 3)) || (sum_out && (ldsum & 3))) return STYLER_EALIGN;
  if (in_drop_p < 0.f || in_drop_p >= 1.f) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  if (rows >= ((int64_t)1 << 31)) return STYLER_EINVAL;
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, res, ldres, gamma, beta, y, ldy, dot_w, dot_b, dot_out, rows, L, len, drop_p, drop_seed,
                     g_styler_drop_epoch, in_drop_p, in_drop_seed, sum_out, ldsum, y16, ldy16, io_flags, y3, y3parts);
  return launch_status();
}

// ---------------------------------------------------------------------------------------
// GroupNorm + ReLU, two launches so that the whole chip streams: (item b, 64-channel chunk = 4 groups) alone is
// only B*C/64 blocks (240 at the model's shapes, 12 % of the wave slots: 22 us against an 8 us stream).  The time
// axis is therefore cut into segments -- grid (C/64, B, nseg):
//   gn_stats : per-segment sum / sumsq of each group in fp64 -> fp64 atomics into ws[B][C/16][2];
//   gn_apply : mean / rstd from ws, normalise, ReLU (the second read of x comes from L2 / MALL).
// 256 threads = 16 row-lanes x 16 channel-quads: thread (rl, cq) streams rows rl, rl+16, ... of its segment reading
// float4 = 4 channels of group cq/4.
int gn_segments_host(int B, int L, int C) {
  int nseg = (1024 + (C / 64) * B - 1) / ((C / 64) * B);
  const int cap = (L + 31) / 32;
  if (nseg > cap) nseg = cap;
  return nseg < 1 ? 1 : nseg;
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int64_t ldx, double* __restrict__ ws,
                                                       int L, int C, int seg_rows) {
  __shared__ double red[2][16][16];
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int t0 = blockIdx.z * seg_rows;
  const int t1 = min(L, t0 + seg_rows);
  const float* xp = x + (int64_t)b * L * ldx + c0 + cq * 4;
  double s = 0.0, ss = 0.0;
  for (int t = t0 + rl; t < t1; t += 16) {
    const float4 v = *reinterpret_cast<const float4*>(xp + (int64_t)t * ldx);
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  red[0][rl][cq] = s; red[1][rl][cq] = ss;
  __syncthreads();
  if (threadIdx.x < 8) {
    const int g = threadIdx.x & 3, which = threadIdx.x >> 2;
    double t = 0.0;
    for (int r = 0; r < 16; ++r)
      for (int q = 0; q < 4; ++q) t += red[which][r][g * 4 + q];
    atomicAdd(&ws[((int64_t)b * (C / 16) + c0 / 16 + g) * 2 + which], t);
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int64_t ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const double* __restrict__ ws, void* __restrict__ y, int64_t ldy,
                                                       float* __restrict__ stats, int L, int C, int seg_rows, int y16,
                                                       uint16_t* __restrict__ y3, int y3parts) {
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int t0 = blockIdx.z * seg_rows;
  const int t1 = min(L, t0 + seg_rows);
  const int64_t gi = ((int64_t)b * (C / 16) + c0 / 16 + (cq >> 2)) * 2;
  const double n = 16.0 * L;
  const double dmean = ws[gi] / n;
  double var = ws[gi + 1] / n - dmean * dmean;
  if (var < 0.0) var = 0.0;
  const float mean = (float)dmean, rstd = (float)(1.0 / sqrt(var + 1e-5));
  if (stats && blockIdx.z == 0 && rl == 0 && (cq & 3) == 0) { stats[gi] = mean; stats[gi + 1] = rstd; }
  const float4 g = *reinterpret_cast<const float4*>(gamma + c0 + cq * 4);
  const float4 bt = *reinterpret_cast<const float4*>(beta + c0 + cq * 4);
  const float* xp = x + (int64_t)b * L * ldx + c0 + cq * 4;
  // y16: the output lives as bf16 (throughput mode: its only consumers, the next convolution and that convolution's
  // weight gradient, round it to bf16 anyway -- same results, half the bytes written here and read there)
  float* yp = reinterpret_cast<float*>(y) + (int64_t)b * L * ldy + c0 + cq * 4;
  uint16_t* yp16 = reinterpret_cast<uint16_t*>(y) + (int64_t)b * L * ldy + c0 + cq * 4;
  for (int t = t0 + rl; t < t1; t += 16) {
    const float4 v = *reinterpret_cast<const float4*>(xp + (int64_t)t * ldx);
    float4 o;
    o.x = fmaxf((v.x - mean) * rstd * g.x + bt.x, 0.f);
    o.y = fmaxf((v.y - mean) * rstd * g.y + bt.y, 0.f);
    o.z = fmaxf((v.z - mean) * rstd * g.z + bt.z, 0.f);
    o.w = fmaxf((v.w - mean) * rstd * g.w + bt.w, 0.f);
    if (y16) *reinterpret_cast<uint2*>(yp16 + (int64_t)t * ldy) = make_uint2(cvt_pk_bf16_rne(o.x, o.y), cvt_pk_bf16_rne(o.z, o.w));
    else *reinterpret_cast<float4*>(yp + (int64_t)t * ldy) = o;
    if (y3) x3_store4(y3, (int64_t)b * L + t, c0 + cq * 4, C, y3parts, o);
  }
}

// Single-pass GroupNorm + ReLU for items of up to 64 * IT rows (IT = 8 or 16 float4 rows per thread): block = (64 channels = 4 groups, item), 1024 threads
// = 64 row-lanes x 16 channel-quads.  Every thread loads ALL its rows up front (GNF_IT float4 in flight per thread, 16
// waves per CU) and keeps them in registers across the statistics -- mean, then the centred second moment, each one
// wave-shuffle step + one LDS exchange between the block's 16 waves -- so x is read from HBM once (the two-kernel form
// reads it twice: 9.3 + 14.2 us per layer at the C2 shape).  Statistics in fp32 from centred values (the two-kernel form
// needs fp64 sums because it forms E[x^2] - mean^2).
#define GNF_IT 8                                    // rows per thread of the short variant; the long one holds 16
__device__ __forceinline__ float gn_group_sum_wave(float s) {    // over the lanes of one group: bits 0,1 (channel quad
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);          // within the group) and bits 4,5 (row-lane) of the lane id
  s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
  return s;
}
// X16: x (the convolution output, kept for the backward) is stored as bf16 (STYLER_IO_Z_BF16).
// Y3 (round 5, bf16x3): the split of the fp32 output is stored too (styler_set_x3_out).  A template argument, not a runtime
// test: the two extra kernel arguments cost the bf16 instantiations registers they do not have (gn_bwd_fused spilled, bn_apply
// went from 98 to 146 VGPRs when y3 was a plain argument of every instantiation).
template <int IT, bool X16 = false, bool Y3 = false>
__global__ __launch_bounds__(1024) void gn_fused_kernel(const void* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        void* __restrict__ y, int64_t ldy, float* __restrict__ stats, int L,
                                                        int C, int y16, uint16_t* __restrict__ y3, int y3parts) {
  __shared__ float red[2][16][4];
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4, wave = threadIdx.x >> 6, grp = cq >> 2;
  const bool writer = (threadIdx.x & 0x33) == 0;                  // one lane per (wave, group)
  const int64_t xoff = (int64_t)b * L * ldx + c0 + cq * 4;
  float4 v[IT];
  {
    typename Raw4<X16>::T rv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int t = rl + 64 * i < L ? rl + 64 * i : L - 1;        // clamped, masked below: no load sits behind a branch
      rv[i] = raw4_load<X16>(x, xoff + (int64_t)t * ldx);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) v[i] = raw4_f32(rv[i]);
  }
  const float inv_n = 1.f / (16.f * (float)L);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i)
    if (rl + 64 * i < L) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  s = gn_group_sum_wave(s);
  if (writer) red[0][wave][grp] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += red[0][w][grp];
  const float mean = tot * inv_n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    if (rl + 64 * i < L) q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  q = gn_group_sum_wave(q);
  if (writer) red[1][wave][grp] = q;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += red[1][w][grp];
  const float rstd = (float)(1.0 / sqrt((double)(tot * inv_n) + 1e-5));
  if (stats && rl == 0 && (cq & 3) == 0) {
    const int64_t gi = ((int64_t)b * (C / 16) + c0 / 16 + grp) * 2;
    stats[gi] = mean; stats[gi + 1] = rstd;
  }
  const float4 g = *reinterpret_cast<const float4*>(gamma + c0 + cq * 4);
  const float4 bt = *reinterpret_cast<const float4*>(beta + c0 + cq * 4);
  const float4 a = make_float4(rstd * g.x, rstd * g.y, rstd * g.z, rstd * g.w);
  float* yp = reinterpret_cast<float*>(y) + (int64_t)b * L * ldy + c0 + cq * 4;
  uint16_t* yp16 = reinterpret_cast<uint16_t*>(y) + (int64_t)b * L * ldy + c0 + cq * 4;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int t = rl + 64 * i;
    if (t >= L) break;
    float4 o;
    o.x = fmaxf(v[i].x * a.x + bt.x, 0.f); o.y = fmaxf(v[i].y * a.y + bt.y, 0.f);
    o.z = fmaxf(v[i].z * a.z + bt.z, 0.f); o.w = fmaxf(v[i].w * a.w + bt.w, 0.f);
    if (y16) *reinterpret_cast<uint2*>(yp16 + (int64_t)t * ldy) = make_uint2(cvt_pk_bf16_rne(o.x, o.y), cvt_pk_bf16_rne(o.z, o.w));
    else *reinterpret_cast<float4*>(yp + (int64_t)t * ldy) = o;
    if (Y3) x3_store4(y3, (int64_t)b * L + t, c0 + cq * 4, C, y3parts, o);    // round 5, bf16x3 (styler_set_x3_out)
  }
}
// 0: two-kernel form; otherwise the rows per thread of the single-pass variant that holds an item of L rows.  The backward
// keeps x AND dy in registers: it has no 16-row variant (85 registers spilled at the 128 a 1024-thread block may use).
int gn_fused_iters(int L, bool bwd) {
  static const int on = [] { const char* e = getenv("STYLER_GN_FUSED"); return e ? atoi(e) : 1; }();
  if (!on) return 0;
  if (L <= 64 * GNF_IT) return GNF_IT;
  if (L <= 128 * GNF_IT && !bwd) return 2 * GNF_IT;
  return 0;
}

// Longest item (rows) the single-pass GroupNorm kernels take (forward / backward); 0 when they are switched off.  A bf16
// x (STYLER_IO_Z_BF16) is accepted up to this length only.
extern "C" int styler_groupnorm_fused_rows(int bwd) {
  if (!gn_fused_iters(1, bwd != 0)) return 0;
  return bwd ? 64 * GNF_IT : 128 * GNF_IT;
}

extern "C" int styler_groupnorm_relu(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                                     int64_t ldy, float* stats, double* workspace, int ws_zeroed, int B, int L, int C,
                                     int io_flags, void* stream) {
  uint16_t* y3 = nullptr;
  int y3parts = 0;
  styler_take_x3_out(&y3, &y3parts);               // (bf16x3: the split of the fp32 output rows, filed by the caller)
  if (y3 && ((io_flags & STYLER_IO_Y_BF16) || ldy != C)) return STYLER_EINVAL;
  if (!x || !y || !gamma || !beta || !workspace || B <= 0 || L <= 0 || C <= 0 || (C & 63)) return STYLER_EINVAL;
  if ((ldx & 3) || (ldy & 3)) return STYLER_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  const bool x16 = (io_flags & STYLER_IO_Z_BF16) != 0;
  if (const int it = gn_fused_iters(L, false)) {
    const int y16 = (io_flags & STYLER_IO_Y_BF16) ? 1 : 0;
#define GNF_LAUNCH(IT_, X_, Y_) hipLaunchKernelGGL((gn_fused_kernel<IT_, X_, Y_>), dim3(C / 64, B), dim3(1024), 0, st, x, ldx, gamma, beta, y, \
                                                   ldy, stats, L, C, y16, y3, y3parts)
    if (y3 && x16) return STYLER_EINVAL;             // (the split goes with fp32 tensors: the bf16x3 arithmetic stores fp32)
    if (it == GNF_IT) { if (x16) GNF_LAUNCH(GNF_IT, true, false); else if (y3) GNF_LAUNCH(GNF_IT, false, true); else GNF_LAUNCH(GNF_IT, false, false); }
    else { if (x16) GNF_LAUNCH(2 * GNF_IT, true, false); else if (y3) GNF_LAUNCH(2 * GNF_IT, false, true); else GNF_LAUNCH(2 * GNF_IT, false, false); }
#undef GNF_LAUNCH
    return launch_status();
  }
  if (x16) return STYLER_EINVAL;                     // bf16 x: single-pass variant only (styler_groupnorm_fused_rows)
  if (!ws_zeroed) {
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double) * 2 * B * (C / 16), st);
    if (e != hipSuccess) return (int)e;
  }
  const int nseg = gn_segments_host(B, L, C);
  const int seg_rows = ((L + nseg - 1) / nseg + 15) & ~15;
  const dim3 grid(C / 64, B, (L + seg_rows - 1) / seg_rows);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, st, x, ldx, workspace, L, C, seg_rows);
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), 0, st, x, ldx, gamma, beta, workspace, y, ldy, stats, L, C,
                     seg_rows, (io_flags & STYLER_IO_Y_BF16) ? 1 : 0, y3, y3parts);
  return launch_status();
}

// ---------------------------------------------------------------------------------------
__global__ void bn_fold_kernel(const float* g, const float* b, const float* rm, const float* rv, const float* cb,
                               float* scale, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = g[c] / sqrtf(rv[c] + 1e-5f);
  scale[c] = sc;
  shift[c] = ((cb ? cb[c] : 0.f) - rm[c]) * sc + b[c];
}

extern "C" int styler_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                              const float* running_var, const float* conv_bias, float* scale, float* shift, int C,
                              void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, conv_bias, scale, shift, C);
  return launch_status();
}

// Train-mode BatchNorm: pass 1 column sums, pass 2 finalize + normalise.
// Column sums: block = (segment, chunk of 128 rows, tile of 128 channels), 256 threads as (row-lane, float4 column); per-thread
// fp64 partials are folded across the row-lanes in LDS.  Round 5: they leave as plain STORES into the block's own slot
// ws[segment][chunk][2C] (until round 4: fp64 atomics into one of 16 replicas -- 41 us with them, 14 us without at
// [42 336, 512], and the last source of run-to-run order noise in a training step); bn_fold_slots_kernel / bn_finalize_kernel
// add the chunks of a column in a FIXED order (the 64 lanes of a wave take every 64th chunk, then a fixed shuffle tree).
// Workspace: styler_bn_workspace_doubles(rows, C, segs) = segs * ceil(rows / segs / 128) * 2C doubles, no zeroing needed.

#define BN_RPB 32                                  // rows per block of the apply kernels
#define BN_STAT_RPB 128                            // rows per block of the column statistics
#define BN_CT 32                                   // float4 columns per block of the column statistics

template <bool BWD, bool DY16 = false, bool X16 = false>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const void* __restrict__ x, const float* __restrict__ y,
                                                          const void* __restrict__ dy, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, double* __restrict__ ws,
                                                          int64_t rows, int C, int act, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float drop_p,
                                                          uint64_t drop_seed_host, const uint64_t* __restrict__ epoch,
                                                          int rpb, int bps, int64_t rps, int dbg) {
  // Segments: rows [seg * rps, (seg + 1) * rps) carry their own statistics (the clean and the noisy decode of
  // styler.py:52,55 run through the PostNet as ONE batch, but each call of the reference normalises with its own batch
  // statistics, Layers.py:126).  Block = (segment, chunk of rpb rows); per-segment arrays are [segs][...].
  // The column axis is cut into tiles of BN_CT float4 columns (128 channels): a block is (segment, chunk of rpb rows, column
  // tile) = 8 row-lanes x 32 column quads.  The fp64 atomics that carry the block's sums away are what this kernel's time is
  // made of (41 us with them, 14 us without, at [42 336, 512]); their number is rows / rpb * 2C whatever the column split, so
  // tall narrow blocks (128 rows) cut them 4x against full-width blocks of 32 rows at the same block count.
  const int nq = C / 4;
  const int nqt = nq < BN_CT ? nq : BN_CT;           // threads across a row
  const int ct = (nq + nqt - 1) / nqt;               // column tiles
  const int ctile = blockIdx.x % ct, bc = blockIdx.x / ct;
  const int seg = bc / bps, chunk = bc - seg * bps;
  ws += ((int64_t)seg * bps + chunk) * 2 * C;         // this block's slot: [2C] doubles (its column tile of them)
  if (BWD) { mean += (int64_t)seg * C; rstd += (int64_t)seg * C; }
  const uint2 dkey = dropout_key(mix_drop_epoch(drop_seed_host, epoch));
  const uint32_t dthr = dropout_thr16(drop_p);
  const float dsc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  __shared__ double red[256][8];
  const int lanes = 256 / nqt;                       // row-lanes
  const int rl = threadIdx.x / nqt, ql = threadIdx.x - rl * nqt;
  const bool live = rl < lanes;
  const int64_t r0 = (int64_t)seg * rps + (int64_t)chunk * rpb;
  int64_t r1 = r0 + rpb; if (r1 > (seg + 1) * rps) r1 = (seg + 1) * rps;
  (void)rows;
  double* const wsc = ws;
  {
    const int q = ctile * nqt + ql;
    double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
    if (live && q < nq) {
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f), rs = m, ga = m, be = m;
      if (BWD) {
        m = *reinterpret_cast<const float4*>(mean + q * 4); rs = *reinterpret_cast<const float4*>(rstd + q * 4);
        ga = *reinterpret_cast<const float4*>(gamma + q * 4);
        if (beta) be = *reinterpret_cast<const float4*>(beta + q * 4);
      }
      const bool has_y = BWD && y && act == STYLER_ACT_TANH;
      // U rows' loads in flight per thread and batch (the accumulation chain is fp64 and serial, the loads are not): a
      // row-at-a-time loop is one memory round trip per row -- 16 in a row at 32 rows per block
      constexpr int U = BWD ? 4 : 8;
      for (int64_t r = r0 + rl; r < r1; r += (int64_t)U * lanes) {
        float4 o4[U];
        typename Raw4<X16>::T v4[U];
        typename Raw4<DY16>::T g4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int64_t ru = r + (int64_t)u * lanes;
          ru = ru < r1 ? ru : r1 - 1;                      // clamped, discarded below: no lane branches around loads
          v4[u] = raw4_load<X16>(x, ru * C + q * 4);
          if (BWD) {
            g4[u] = raw4_load<DY16>(dy, ru * C + q * 4);
            if (has_y) o4[u] = *reinterpret_cast<const float4*>(y + ru * C + q * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t ru = r + (int64_t)u * lanes;
          const bool okr = ru < r1;                        // a select, not a branch: a branch lets the loads sink to it
          float4 v = raw4_f32(v4[u]);
          if (!okr) v = make_float4(BWD ? m.x : 0.f, BWD ? m.y : 0.f, BWD ? m.z : 0.f, BWD ? m.w : 0.f);
          if (BWD) {
            float4 g = okr ? raw4_f32(g4[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 o = has_y ? o4[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float hx = (v.x - m.x) * rs.x, hy = (v.y - m.y) * rs.y, hz = (v.z - m.z) * rs.z, hw = (v.w - m.w) * rs.w;
            const uint64_t e = (uint64_t)(ru * C + q * 4);
            const float4 ks = drop_p > 0.f ? dropout_scale4(dropout_word4(dkey, (uint32_t)e, (uint32_t)(e >> 32)), dthr, dsc)
                                           : make_float4(1.f, 1.f, 1.f, 1.f);
            g.x = bn_dz_elem(g.x, hx, ga.x, be.x, act, has_y, o.x, ks.x);
            g.y = bn_dz_elem(g.y, hy, ga.y, be.y, act, has_y, o.y, ks.y);
            g.z = bn_dz_elem(g.z, hz, ga.z, be.z, act, has_y, o.z, ks.z);
            g.w = bn_dz_elem(g.w, hw, ga.w, be.w, act, has_y, o.w, ks.w);
            s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
            t[0] += (double)g.x * hx; t[1] += (double)g.y * hy; t[2] += (double)g.z * hz; t[3] += (double)g.w * hw;
          } else {
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            t[0] += (double)v.x * v.x; t[1] += (double)v.y * v.y; t[2] += (double)v.z * v.z; t[3] += (double)v.w * v.w;
          }
        }
      }
    }
    if (lanes > 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { red[threadIdx.x][k] = s[k]; red[threadIdx.x][4 + k] = t[k]; }
      __syncthreads();
      if (rl == 0)
        for (int l = 1; l < lanes; ++l)
#pragma unroll
          for (int k = 0; k < 4; ++k) { s[k] += red[l * nqt + ql][k]; t[k] += red[l * nqt + ql][4 + k]; }
      __syncthreads();
    }
    if (rl == 0 && q < nq && !(dbg & 1)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { wsc[q * 4 + k] = s[k]; wsc[C + q * 4 + k] = t[k]; }
    }
  }
}

// Sums over the chunk slots of TWO columns (a, b) at once, fixed order: lane l of the pair's 64 lanes adds chunks l, l + 64, ...
// (both columns' loads in flight together), a wave shuffle tree in a fixed pattern adds the 64 lane sums -- every lane returns
// the totals.  (331 chunks at the PostNet shape: five or six dependent loads per lane; the 16-lane form of the first build took
// 21 per lane and the forward's finalize pass went from 5 to ~12 us.)
__device__ __forceinline__ void bn_fold_pair(const double* __restrict__ wa, const double* __restrict__ wb, int64_t slot_stride,
                                             int nslots, int lane, double& sa, double& sb) {
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  int k = lane;
  for (; k + 64 < nslots; k += 128) {
    a0 += wa[(int64_t)k * slot_stride]; b0 += wb[(int64_t)k * slot_stride];
    a1 += wa[(int64_t)(k + 64) * slot_stride]; b1 += wb[(int64_t)(k + 64) * slot_stride];
  }
  if (k < nslots) { a0 += wa[(int64_t)k * slot_stride]; b0 += wb[(int64_t)k * slot_stride]; }
  sa = wave_sum_d(a0 + a1);                          // xor butterfly: the same association on every lane and every launch
  sb = wave_sum_d(b0 + b1);
}

// backward: ws[seg][0][i] = sum over the chunk slots of column i (i < 2C: the dbeta / dgamma partial sums), in place.
// One wave per pair of columns (i, i + C): 4 pairs per block.
__global__ __launch_bounds__(256) void bn_fold_slots_kernel(double* __restrict__ ws, int C, int nslots) {
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6), seg = blockIdx.y;
  if (c >= C) return;
  double* w = ws + (int64_t)seg * nslots * 2 * C;
  double s1, s2;
  bn_fold_pair(w + c, w + C + c, 2 * C, nslots, lane, s1, s2);
  if (lane == 0) { w[c] = s1; w[C + c] = s2; }       // slot 0 of the two columns: read above by THIS lane only (k = lane = 0)
}

extern "C" int64_t styler_bn_workspace_doubles(int64_t rows, int C, int segs) {
  if (rows <= 0 || C <= 0 || segs < 1) return 0;
  static const int rpb_env = [] { const char* e = getenv("STYLER_BN_RPB"); return e ? atoi(e) : BN_STAT_RPB; }();
  const int rpb = rpb_env > 0 ? rpb_env : BN_STAT_RPB;
  const int64_t rps = rows / segs;
  return (int64_t)segs * ((rps + rpb - 1) / rpb) * 2 * C;
}

int styler_bn_colstats(bool bwd, const void* x, const float* y, const void* dy, const float* mean, const float* rstd,
                       double* ws, int ws_zeroed, int64_t rows, int C, int act, const float* gamma, const float* beta,
                       float drop_p, uint64_t drop_seed, int segs, int dy16, int x16, hipStream_t st, bool fold) {
  (void)ws_zeroed;                                   // (slots are stored, not accumulated: nothing to clear)
  static const int rpb_env = [] { const char* e = getenv("STYLER_BN_RPB"); return e ? atoi(e) : BN_STAT_RPB; }();
  const int rpb = rpb_env > 0 ? rpb_env : BN_STAT_RPB;
  static const int dbg = [] { const char* e = getenv("STYLER_BN_DBG"); return e ? atoi(e) : 0; }();   // timing experiments
  const int64_t rps = rows / segs;                   // rows per segment
  const int bps = (int)((rps + rpb - 1) / rpb);      // blocks per segment
  const int nq = C / 4, nqt = nq < BN_CT ? nq : BN_CT;
  const dim3 grid((unsigned)(bps * segs * ((nq + nqt - 1) / nqt)));
#define BN_STATS_LAUNCH(...) hipLaunchKernelGGL((bn_colstats_kernel<__VA_ARGS__>), grid, dim3(256), 0, st, x, y, dy, mean, rstd, ws, \
                                                 rows, C, act, gamma, beta, drop_p, drop_seed, g_styler_drop_epoch, rpb, bps, rps, dbg)
  if (bwd && dy16 && x16) BN_STATS_LAUNCH(true, true, true);
  else if (bwd && dy16) BN_STATS_LAUNCH(true, true, false);
  else if (bwd && x16) BN_STATS_LAUNCH(true, false, true);
  else if (bwd) BN_STATS_LAUNCH(true, false, false);
  else if (x16) BN_STATS_LAUNCH(false, false, true);
  else BN_STATS_LAUNCH(false, false, false);
#undef BN_STATS_LAUNCH
  // (the forward's finalize kernel sums the replicas itself: one launch less per BatchNorm layer)
  if (fold) hipLaunchKernelGGL(bn_fold_slots_kernel, dim3((C + 3) / 4, segs), dim3(256), 0, st, ws, C, bps);
  return 0;
}

// one wave per channel (4 per block): its lanes fold the chunk slots of the channel's two column sums (fixed order), lane 0 finishes
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ ws, float* save_mean, float* save_rstd,
                                                          float* running_mean, float* running_var, int64_t rows, int C, int segs,
                                                          int nslots) {
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const double n = (double)(rows / segs);
  for (int seg = 0; seg < segs; ++seg) {             // the running statistics see the segments as consecutive calls
    const double* w = ws + (int64_t)seg * nslots * 2 * C;
    double s1, s2;
    bn_fold_pair(w + c, w + C + c, 2 * C, nslots, lane, s1, s2);
    if (lane != 0) continue;
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[seg * C + c] = (float)mean;
    save_rstd[seg * C + c] = (float)(1.0 / sqrt(var + 1e-5));
    if (running_mean) {
      const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
      running_mean[c] = 0.9f * running_mean[c] + 0.1f * (float)mean;
      running_var[c] = 0.9f * running_var[c] + 0.1f * (float)unbiased;
    }
  }
}

// Normalise + activation + dropout.  Same geometry as the column statistics: block = (segment, chunk of rpb rows), thread =
// (row-lane, float4 column): the per-channel constants are fetched once per thread, the rows in batches of four, and no
// index is ever divided.
template <bool X16, bool Y3 = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, void* __restrict__ yv,
                                                       int C, int act, float drop_p, uint64_t drop_seed_host,
                                                       const uint64_t* __restrict__ epoch, int rpb, int bps, int64_t rps,
                                                       int y16, uint16_t* __restrict__ y3, int y3parts) {
  float* const y = reinterpret_cast<float*>(yv);
  uint16_t* const yh = reinterpret_cast<uint16_t*>(yv);    // y16: bf16 output (see gn_apply_kernel)
  const int seg = blockIdx.x / bps, chunk = blockIdx.x - seg * bps;
  const int nq = C / 4;
  const int nqt = nq < 256 ? nq : 256;
  const int lanes = 256 / nqt;
  const int rl = threadIdx.x / nqt, ql = threadIdx.x - rl * nqt;
  if (rl >= lanes) return;
  const uint2 dkey = dropout_key(mix_drop_epoch(drop_seed_host, epoch));
  const uint32_t thr = dropout_thr16(drop_p);
  const float sc = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int64_t r0 = (int64_t)seg * rps + (int64_t)chunk * rpb;
  int64_t r1 = r0 + rpb; if (r1 > (seg + 1) * rps) r1 = (seg + 1) * rps;
  for (int q = ql; q < nq; q += nqt) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + q * 4);
    const float4 b = *reinterpret_cast<const float4*>(beta + q * 4);
    const float4 m = *reinterpret_cast<const float4*>(mean + (int64_t)seg * C + q * 4);
    const float4 r = *reinterpret_cast<const float4*>(rstd + (int64_t)seg * C + q * 4);
    // y = (x - m) * r * g + b = x * a + c
    const float4 a = make_float4(r.x * g.x, r.y * g.y, r.z * g.z, r.w * g.w);
    constexpr int U = 4;
    for (int64_t row = r0 + rl; row < r1; row += (int64_t)U * lanes) {
      typename Raw4<X16>::T v4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t ru = row + (int64_t)u * lanes;
        ru = ru < r1 ? ru : r1 - 1;
        v4[u] = raw4_load<X16>(x, ru * C + q * 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t ru = row + (int64_t)u * lanes;
        if (ru >= r1) break;
        const float4 v = raw4_f32(v4[u]);
        float4 o;
        o.x = apply_act((v.x - m.x) * a.x + b.x, act);
        o.y = apply_act((v.y - m.y) * a.y + b.y, act);
        o.z = apply_act((v.z - m.z) * a.z + b.z, act);
        o.w = apply_act((v.w - m.w) * a.w + b.w, act);
        if (drop_p > 0.f) {                              // F.dropout after the activation (Layers.py:126-128), the stream
          o = dropout_apply4(o, dkey, (uint64_t)(ru * C + q * 4), thr, sc);   // styler_dropout would draw on the [rows, C] tensor
        }
        if (y16) *reinterpret_cast<uint2*>(yh + ru * C + q * 4) = make_uint2(cvt_pk_bf16_rne(o.x, o.y), cvt_pk_bf16_rne(o.z, o.w));
        else *reinterpret_cast<float4*>(y + ru * C + q * 4) = o;
        if (Y3) x3_store4(y3, ru, q * 4, C, y3parts, o);     // round 5, bf16x3: the split of the fp32 output (styler_set_x3_out)
      }
    }
  }
}

extern "C" int styler_batchnorm_train(const float* x, const float* gamma, const float* beta, void* y,
                                      float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                                      double* workspace, int ws_zeroed, int64_t rows, int C, int act, float drop_p,
                                      uint64_t drop_seed, int segs, int io_flags, void* stream) {
  uint16_t* y3 = nullptr;
  int y3parts = 0;
  styler_take_x3_out(&y3, &y3parts);               // (bf16x3: the split of the fp32 output rows, filed by the caller)
  if (y3 && (io_flags & (STYLER_IO_Y_BF16 | STYLER_IO_Z_BF16))) return STYLER_EINVAL;
  if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !workspace || rows <= 0 || C <= 0 || (C & 3) ||
      drop_p < 0.f || drop_p >= 1.f || segs < 1 || rows % segs)
    return STYLER_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int x16 = (io_flags & STYLER_IO_Z_BF16) ? 1 : 0;
  const int rc = styler_bn_colstats(false, x, nullptr, nullptr, nullptr, nullptr, workspace, ws_zeroed, rows, C, act, nullptr,
                                    nullptr, 0.f, 0, segs, 0, x16, st, /*fold=*/false);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, workspace, save_mean, save_rstd,
                     running_mean, running_var, rows, C, segs, (int)(styler_bn_workspace_doubles(rows, C, segs) / ((int64_t)segs * 2 * C)));
  const int64_t rps = rows / segs;
  const int bps = (int)((rps + BN_RPB - 1) / BN_RPB);
  if (x16)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3((unsigned)(bps * segs)), dim3(256), 0, st, x, gamma, beta, save_mean, save_rstd,
                       y, C, act, drop_p, drop_seed, g_styler_drop_epoch, BN_RPB, bps, rps, (io_flags & STYLER_IO_Y_BF16) ? 1 : 0, y3, y3parts);
  else if (y3)
    hipLaunchKernelGGL((bn_apply_kernel<false, true>), dim3((unsigned)(bps * segs)), dim3(256), 0, st, x, gamma, beta, save_mean, save_rstd,
                       y, C, act, drop_p, drop_seed, g_styler_drop_epoch, BN_RPB, bps, rps, 0, y3, y3parts);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3((unsigned)(bps * segs)), dim3(256), 0, st, x, gamma, beta, save_mean, save_rstd,
                       y, C, act, drop_p, drop_seed, g_styler_drop_epoch, BN_RPB, bps, rps, (io_flags & STYLER_IO_Y_BF16) ? 1 : 0, y3, y3parts);
  return launch_status();
}
