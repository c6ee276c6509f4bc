// Recurrent half of nn.LSTM(.., bidirectional=True) for one layer (modules.py:117,132,147,162,179-182).
//
// The input projections x W_ih^T + b_ih + b_hh for both directions are one MFMA GEMM (styler_conv_gemm,
// N = 2*4H).  This kernel runs the sequential part: one block per (item, direction), 4H threads; thread j
// keeps row j of W_hh in REGISTERS (H floats) for the whole sequence, h_{t-1} is broadcast from LDS
// (ds_read_b128), gates cross threads through LDS, so a step costs H FMAs + two barriers and touches
// HBM only for the 4H-float gx row and the H-float output row.  Latency-bound by construction
// (S <= ~300 steps); all items/directions/LSTMs run concurrently on separate CUs.
#include "common.h"

template <int H>
__device__ __forceinline__ void lstm_fwd_body(const float* __restrict__ gx, const float* __restrict__ w_hh,
                                              float* __restrict__ out, float* __restrict__ cell_out,
                                              float* __restrict__ gates_out, int S, float* hbuf, float* gbuf) {
  const int j = threadIdx.x;                 // gate row 0..4H-1 (i | f | g | o)
  const int b = blockIdx.x, dir = blockIdx.y;
  if (j >= 4 * H) {                          // surplus threads of a shared launch only take part in the barriers
    __syncthreads();
    for (int step = 0; step < S; ++step) { __syncthreads(); __syncthreads(); }
    return;
  }
  float w[H];
  {
    const float* wp = w_hh + ((int64_t)dir * 4 * H + j) * H;
#pragma unroll
    for (int k = 0; k < H; k += 4) {
      const float4 v = *reinterpret_cast<const float4*>(wp + k);
      w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
    }
  }
  if (j < H) hbuf[j] = 0.f;
  float c = 0.f;
  __syncthreads();
  const int64_t gx_ld = 2 * 4 * H, out_ld = 2 * H;
  const float* gxp = gx + (int64_t)b * S * gx_ld + dir * 4 * H + j;
  int t = dir ? S - 1 : 0;
  const int dt = dir ? -1 : 1;
  float gnext = gxp[(int64_t)t * gx_ld];
  for (int step = 0; step < S; ++step, t += dt) {
    // two packed accumulators (v_pk_fma_f32: H / 2 VALU instructions for the H-term dot product instead of H)
    f32x2_t a01 = {gnext, 0.f}, a23 = {0.f, 0.f};
    if (step + 1 < S) gnext = gxp[(int64_t)(t + dt) * gx_ld];        // prefetch next step's row
#pragma unroll
    for (int k = 0; k < H; k += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(&hbuf[k]);
      const f32x2_t w01 = {w[k], w[k + 1]}, w23 = {w[k + 2], w[k + 3]}, h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
      a01 = __builtin_elementwise_fma(w01, h01, a01);
      a23 = __builtin_elementwise_fma(w23, h23, a23);
    }
    const float pre = (a01.x + a01.y) + (a23.x + a23.y);
    // one exp2 + one rcp per gate, no divergence where a wave straddles two gates: sigmoid(x) = 0.5 + 0.5 tanh(x / 2).
    // (The library tanhf / expf + division were ~60 of the ~100 VALU instructions of a step, and with three blocks per CU
    // the step time IS the VALU instruction count.)
    const bool is_g = (j >= 2 * H) && (j < 3 * H);
    const float th = fast_tanh(is_g ? pre : 0.5f * pre);
    const float act = is_g ? th : fmaf(0.5f, th, 0.5f);
    gbuf[j] = act;
    if (gates_out) gates_out[((int64_t)b * S + t) * gx_ld + dir * 4 * H + j] = act;
    __syncthreads();
    if (j < H) {
      c = gbuf[H + j] * c + gbuf[j] * gbuf[2 * H + j];
      const float h = gbuf[3 * H + j] * fast_tanh(c);
      hbuf[j] = h;
      out[((int64_t)b * S + t) * out_ld + dir * H + j] = h;
      if (cell_out) cell_out[((int64_t)b * S + t) * out_ld + dir * H + j] = c;
    }
    __syncthreads();
  }
}

template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bidir_kernel(const float* __restrict__ gx,
                                                           const float* __restrict__ w_hh, float* __restrict__ out,
                                                           float* __restrict__ cell_out,
                                                           float* __restrict__ gates_out, int S) {
  __shared__ __attribute__((aligned(16))) float hbuf[H];
  __shared__ float gbuf[4 * H];
  lstm_fwd_body<H>(gx, w_hh, out, cell_out, gates_out, S, hbuf, gbuf);
}

// Up to 4 independent BiLSTM layers (the duration / f0 / energy / residual streams, modules.py:179-182) in ONE launch:
// grid (B, 2, count), 320 threads; H is 80 or 64 per descriptor.  The four recurrences are latency-bound and use
// 96 blocks each, so running them side by side costs the time of the slowest instead of the sum.
struct LstmMultiArgs { StylerLstmDesc d[4]; };

__global__ __launch_bounds__(320) void lstm_bidir_multi_kernel(LstmMultiArgs a, int S) {
  __shared__ __attribute__((aligned(16))) float hbuf[80];
  __shared__ float gbuf[320];
  const StylerLstmDesc d = a.d[blockIdx.z];
  if (d.H == 80)
    lstm_fwd_body<80>(d.gx, d.w_hh, d.out, d.cell_out, d.gates_out, S, hbuf, gbuf);
  else
    lstm_fwd_body<64>(d.gx, d.w_hh, d.out, d.cell_out, d.gates_out, S, hbuf, gbuf);
}

extern "C" int styler_lstm_bidir_multi(const StylerLstmDesc* descs, int count, int B, int S, void* stream) {
  if (!descs || count <= 0 || count > 4 || B <= 0 || S <= 0) return STYLER_EINVAL;
  LstmMultiArgs a;
  for (int i = 0; i < count; ++i) {
    a.d[i] = descs[i];
    if (!descs[i].gx || !descs[i].w_hh || !descs[i].out || (descs[i].H != 64 && descs[i].H != 80)) return STYLER_EINVAL;
  }
  hipLaunchKernelGGL(lstm_bidir_multi_kernel, dim3(B, 2, count), dim3(320), 0, (hipStream_t)stream, a, S);
  return launch_status();
}

extern "C" int styler_lstm_bidir(const float* gx, const float* w_hh, float* out, float* cell_out, float* gates_out,
                                 int B, int S, int H, void* stream) {
  if (!gx || !w_hh || !out || B <= 0 || S <= 0) return STYLER_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (H == 64)
    hipLaunchKernelGGL(lstm_bidir_kernel<64>, dim3(B, 2), dim3(256), 0, st, gx, w_hh, out, cell_out, gates_out, S);
  else if (H == 80)
    hipLaunchKernelGGL(lstm_bidir_kernel<80>, dim3(B, 2), dim3(320), 0, st, gx, w_hh, out, cell_out, gates_out, S);
  else
    return STYLER_EINVAL;
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Backward of the recurrent half (BPTT).  One block per (item, direction), 4H threads.
// Saved by forward: gates (post-activation i|f|g|o) [B,S,2*4H], cell states c [B,S,2H], hidden states out [B,S,2H].
// Produces dgp = dL/d(pre-activation gates) [B,S,2*4H]; every weight / input gradient is then a GEMM:
//   dx = dgp W_ih (styler_conv_gemm), dW_ih = dgp^T x, dW_hh = dgp^T h_{prev} (styler_wgrad, shift -1 / +1),
//   db_ih = db_hh = colsum(dgp).
// Per step: threads u < H form the four gate gradients of unit u; then all 4H threads do the transposed
// mat-vec dh_prev[k] = sum_j W_hh[j][k] dgp[j] with thread (q = tid / H, k = tid % H) holding the H
// weights W_hh[q*H + j'][k] in registers and the 4 partial sums combined through LDS.
template <int H>
__device__ __forceinline__ void lstm_bwd_body(const float* __restrict__ dout, const float* __restrict__ gates,
                                              const float* __restrict__ cell, const float* __restrict__ w_hh,
                                              float* __restrict__ dgp, int S, float* sdg, float (*part)[80]) {
  const int tid = threadIdx.x, q = tid / H, k = tid % H;
  const int b = blockIdx.x, dir = blockIdx.y;
  if (tid >= 4 * H) {                        // surplus threads of a shared launch only take part in the barriers
    for (int step = 0; step < S; ++step) { __syncthreads(); __syncthreads(); }
    return;
  }
  float w[H];
  {
    const float* wp = w_hh + ((int64_t)dir * 4 * H + q * H) * H + k;     // column k of gate block q
#pragma unroll
    for (int j = 0; j < H; ++j) w[j] = wp[(int64_t)j * H];
  }
  float dh_rec = 0.f, dc_next = 0.f;
  const int64_t g_ld = 2 * 4 * H, o_ld = 2 * H;
  // reverse of the forward processing order: forward dir processed t = 0..S-1, so walk S-1..0 (and vice versa)
  int t = dir ? 0 : S - 1;
  const int dt = dir ? 1 : -1;
  // The step's inputs (four gates, dout, cell) do not depend on the recurrence: they are fetched one step ahead -- the cell
  // state two steps ahead, because c_prev of step t is the cell of the step the walk visits next -- so the recurrence never
  // waits for memory (it did, seven dependent loads at the top of every step: ~40 % of the 0.47 us step).
  const bool unit = tid < H;
  const int u = unit ? tid : 0;
  auto goff = [&](int tt) { return ((int64_t)b * S + tt) * g_ld + dir * 4 * H + u; };
  auto ooff = [&](int tt) { return ((int64_t)b * S + tt) * o_ld + dir * H + u; };
  float gi = 0.f, gf = 0.f, gg = 0.f, gout = 0.f, dov = 0.f, c = 0.f, c_n1 = 0.f;
  float n_gi = 0.f, n_gf = 0.f, n_gg = 0.f, n_gout = 0.f, n_dov = 0.f, c_n2 = 0.f;
  if (unit) {
    const int64_t go = goff(t);
    gi = gates[go]; gf = gates[go + H]; gg = gates[go + 2 * H]; gout = gates[go + 3 * H];
    dov = dout[ooff(t)]; c = cell[ooff(t)];
    if (S > 1) c_n1 = cell[ooff(t + dt)];
  }
  for (int step = 0; step < S; ++step, t += dt) {
    if (unit) {
      if (step + 1 < S) {                                // next step's gates / dout, the cell two steps ahead
        const int64_t gn = goff(t + dt);
        n_gi = gates[gn]; n_gf = gates[gn + H]; n_gg = gates[gn + 2 * H]; n_gout = gates[gn + 3 * H];
        n_dov = dout[ooff(t + dt)];
      }
      c_n2 = step + 2 < S ? cell[ooff(t + 2 * dt)] : 0.f;
      const int64_t go = goff(t);
      const float c_prev = step + 1 < S ? c_n1 : 0.f;    // previous step in forward processing order = the walk's next
      const float dh = dov + dh_rec;
      const float tc = fast_tanh(c);
      const float d_o = dh * tc;
      const float dc = dc_next + dh * gout * (1.f - tc * tc);
      const float di = dc * gg, dg = dc * gi, df = dc * c_prev;
      dc_next = dc * gf;
      const float pi = di * gi * (1.f - gi), pf = df * gf * (1.f - gf), pg = dg * (1.f - gg * gg),
                  po = d_o * gout * (1.f - gout);
      sdg[u] = pi; sdg[H + u] = pf; sdg[2 * H + u] = pg; sdg[3 * H + u] = po;
      dgp[go] = pi; dgp[go + H] = pf; dgp[go + 2 * H] = pg; dgp[go + 3 * H] = po;
    }
    __syncthreads();
    f32x2_t a01 = {0.f, 0.f}, a23 = {0.f, 0.f};        // packed FMAs: H / 2 VALU instructions for the H-term product
#pragma unroll
    for (int j = 0; j < H; j += 4) {
      const float4 d = *reinterpret_cast<const float4*>(&sdg[q * H + j]);
      const f32x2_t w01 = {w[j], w[j + 1]}, w23 = {w[j + 2], w[j + 3]}, d01 = {d.x, d.y}, d23 = {d.z, d.w};
      a01 = __builtin_elementwise_fma(w01, d01, a01);
      a23 = __builtin_elementwise_fma(w23, d23, a23);
    }
    part[q][k] = (a01.x + a01.y) + (a23.x + a23.y);
    __syncthreads();
    // (two barriers per step are enough: sdg is rewritten only after the second one, which every reader of sdg has
    // passed; part is rewritten only after the next step's first one, which its readers -- the unit threads, right
    // below -- reach after reading it)
    if (unit) {
      dh_rec = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      gi = n_gi; gf = n_gf; gg = n_gg; gout = n_gout; dov = n_dov; c = c_n1; c_n1 = c_n2;
    }
  }
}

template <int H>
__global__ __launch_bounds__(4 * H) void lstm_bidir_bwd_kernel(const float* __restrict__ dout,
                                                               const float* __restrict__ gates,
                                                               const float* __restrict__ cell,
                                                               const float* __restrict__ w_hh,
                                                               float* __restrict__ dgp, int S) {
  __shared__ __attribute__((aligned(16))) float sdg[4 * 80];
  __shared__ float part[4][80];
  lstm_bwd_body<H>(dout, gates, cell, w_hh, dgp, S, sdg, part);
}

struct LstmBwdMultiArgs { StylerLstmBwdDesc d[4]; };

// 4 waves per SIMD stated (<= 128 registers): 768 blocks of 5 waves are then ONE round of three blocks per CU; left to itself
// hipcc takes 130 registers, i.e. two blocks per CU and 1.5 rounds of a kernel that is 441 sequential steps long (198 -> 157 us)
__global__ __launch_bounds__(320, 4) void lstm_bidir_bwd_multi_kernel(LstmBwdMultiArgs a, int S) {
  __shared__ __attribute__((aligned(16))) float sdg[4 * 80];
  __shared__ float part[4][80];
  const StylerLstmBwdDesc d = a.d[blockIdx.z];
  if (d.H == 80)
    lstm_bwd_body<80>(d.dout, d.gates, d.cell, d.w_hh, d.dgp, S, sdg, part);
  else
    lstm_bwd_body<64>(d.dout, d.gates, d.cell, d.w_hh, d.dgp, S, sdg, part);
}

extern "C" int styler_lstm_bidir_bwd_multi(const StylerLstmBwdDesc* descs, int count, int B, int S, void* stream) {
  if (!descs || count <= 0 || count > 4 || B <= 0 || S <= 0) return STYLER_EINVAL;
  LstmBwdMultiArgs a;
  for (int i = 0; i < count; ++i) {
    a.d[i] = descs[i];
    if (!descs[i].dout || !descs[i].gates || !descs[i].cell || !descs[i].w_hh || !descs[i].dgp ||
        (descs[i].H != 64 && descs[i].H != 80))
      return STYLER_EINVAL;
  }
  hipLaunchKernelGGL(lstm_bidir_bwd_multi_kernel, dim3(B, 2, count), dim3(320), 0, (hipStream_t)stream, a, S);
  return launch_status();
}

extern "C" int styler_lstm_bidir_bwd(const float* dout, const float* gates, const float* cell, const float* w_hh,
                                     float* dgp, int B, int S, int H, void* stream) {
  if (!dout || !gates || !cell || !w_hh || !dgp || B <= 0 || S <= 0) return STYLER_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (H == 64)
    hipLaunchKernelGGL(lstm_bidir_bwd_kernel<64>, dim3(B, 2), dim3(256), 0, st, dout, gates, cell, w_hh, dgp, S);
  else if (H == 80)
    hipLaunchKernelGGL(lstm_bidir_bwd_kernel<80>, dim3(B, 2), dim3(320), 0, st, dout, gates, cell, w_hh, dgp, S);
  else
    return STYLER_EINVAL;
  return launch_status();
}
