// Memory-bound stitching kernels of the STYLER path: embeddings + position tables, one-hot-aware first
// conv of the f0/energy streams, mel calibrator, augmentation-classifier tail, bucketise+embed+add,
// strided adds, masked error sums.  All are single-pass, float4-vectorised, channels-last.
#include "common.h"

// ---- embedding + position ---------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_pos_kernel(const int64_t* __restrict__ text,
                                                        const float* __restrict__ emb,
                                                        const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ pe, float* __restrict__ out,
                                                        int64_t rows, int L, int C) {
  const int nq = C / 4;
  const int64_t total = rows * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    const int t = (int)(row % L);
    const float* src = text ? emb + text[row] * C : x + row * ldx;
    const float4 a = *reinterpret_cast<const float4*>(src + q * 4);
    const float4 p = *reinterpret_cast<const float4*>(pe + (int64_t)t * C + q * 4);
    *reinterpret_cast<float4*>(out + row * C + q * 4) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}

static inline unsigned grid_for(int64_t work, int per_block = 256, int cap = 4096) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" int styler_embed_pos(const int64_t* text, const float* emb, const float* pe, float* out, int B, int L,
                                int C, void* stream) {
  if (!text || !emb || !pe || !out || B <= 0 || L <= 0 || (C & 3)) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  hipLaunchKernelGGL(embed_pos_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, text, emb,
                     (const float*)nullptr, (int64_t)0, pe, out, rows, L, C);
  return launch_status();
}

extern "C" int styler_add_pos(const float* x, int64_t ldx, const float* pe, float* out, int B, int L, int C,
                              void* stream) {
  if (!x || !pe || !out || B <= 0 || L <= 0 || (C & 3) || (ldx & 3)) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  hipLaunchKernelGGL(embed_pos_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const int64_t*)nullptr, (const float*)nullptr, x, ldx, pe, out, rows, L, C);
  return launch_status();
}

// out[r, :] = table[ids[r], :] (the un-summed pitch / energy embeddings predict_inference returns, modules.py:300-303)
__global__ __launch_bounds__(256) void gather_rows_kernel(const int32_t* __restrict__ ids, const float* __restrict__ table,
                                                          float* __restrict__ out, int64_t rows, int C) {
  const int nq = C / 4;
  const int64_t total = rows * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    *reinterpret_cast<float4*>(out + row * C + q * 4) =
        *reinterpret_cast<const float4*>(table + (int64_t)ids[row] * C + q * 4);
  }
}

extern "C" int styler_gather_rows(const int32_t* ids, const float* table, float* out, int64_t rows, int C, void* stream) {
  if (!ids || !table || !out || rows <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, ids, table,
                     out, rows, C);
  return launch_status();
}

__global__ void sinusoid_kernel(float* pe, int L, int C) {
  const int64_t total = (int64_t)L * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i / C), j = (int)(i % C);
    const double angle = (double)pos / pow(10000.0, 2.0 * (double)(j / 2) / (double)C);
    pe[i] = (float)((j & 1) ? cos(angle) : sin(angle));
  }
}

extern "C" int styler_sinusoid_table(float* pe, int L, int C, void* stream) {
  if (!pe || L <= 0 || C <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(sinusoid_kernel, dim3(grid_for((int64_t)L * C)), dim3(256), 0, (hipStream_t)stream, pe, L, C);
  return launch_status();
}

// ---- quantise + one-hot-aware Conv1d(257 -> C, k = 5) -------------------------------------
__device__ __forceinline__ int quant_index(float v, int* bad) {
  if (v <= 0.f) return 0;
  if (v > 1.f) { *bad = 1; v = 1.f; }
  return (int)rintf(v * 255.f) + 1;        // rintf = round-half-to-even, as torch.round
}

// one wave per frame; lanes stride the output channels (coalesced 256-B rows of wt).  Nothing is loaded under a condition:
// the five neighbour values come from clamped addresses and the five weight rows from a clamped index, invalid taps are
// dropped by a select -- so the loads of a frame are in flight together (under `if (valid)` each one was its own memory
// round trip: 45 us for 42 k frames).
__global__ __launch_bounds__(256) void onehot_conv5_kernel(const float* __restrict__ v, const float* __restrict__ wt,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int64_t ldy, int32_t* __restrict__ idx_out,
                                                           int32_t* __restrict__ err_flag, int64_t rows, int L,
                                                           int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int t = (int)(row % L);
  int bad = 0;
  float vv[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tt = t + j - 2;
    vv[j] = v[row + ((tt >= 0 && tt < L) ? j - 2 : 0)];
  }
  int idx[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int tt = t + j - 2;
    int bj = 0;
    const int q = quant_index(vv[j], &bj);
    const bool ok = tt >= 0 && tt < L;
    idx[j] = ok ? q : -1;
    bad |= ok ? bj : 0;
  }
  if (bad && lane == 0 && err_flag) atomicOr(err_flag, 1);
  if (idx_out && lane == 0) idx_out[row] = idx[2];
  for (int c = lane; c < C; c += 64) {
    float w[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) w[j] = wt[((int64_t)j * 257 + (idx[j] >= 0 ? idx[j] : 0)) * C + c];
    float acc = bias[c];
#pragma unroll
    for (int j = 0; j < 5; ++j) acc += idx[j] >= 0 ? w[j] : 0.f;
    y[row * ldy + c] = acc;
  }
}

extern "C" int styler_onehot_conv5(const float* v, const float* wt, const float* bias, float* y, int64_t ldy,
                                   int32_t* idx_out, int32_t* err_flag, int B, int L, int C, void* stream) {
  if (!v || !wt || !bias || !y || B <= 0 || L <= 0 || C <= 0) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  hipLaunchKernelGGL(onehot_conv5_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, v, wt,
                     bias, y, ldy, idx_out, err_flag, rows, L, C);
  return launch_status();
}

// ---- mel calibrator ------------------------------------------------------------------------
// grid (S, B); output row s of item b:
//   ml > sl : mean of frames [start, start+n), n = ml/sl + (s < ml%sl), start = s*(ml/sl) + min(s, ml%sl)
//   ml < sl : frame f with f*(q) + min(f, r) <= s, q = sl/ml, r = sl%ml
//   ml == sl: copy;   s >= sl: zeros
// Block = 256 / (C / 4) output rows of one item (C = 80: 12 rows, 240 threads busy; a block per row kept 20 of 256 lanes
// busy and paid a block launch per 320 bytes), thread = (row, float4 column); the frames of a mean are fetched four at a time.
template <bool X16>                                  // x (the concatenated AudioEncoder streams) stored as bf16
__global__ __launch_bounds__(256) void mel_calibrate_kernel(const void* __restrict__ x, int64_t ldx,
                                                            float* __restrict__ y, int64_t ldy,
                                                            const int64_t* __restrict__ mel_len,
                                                            const int64_t* __restrict__ src_len, int T, int S,
                                                            int C) {
  const int nq = C >> 2, cpr = nq < 256 ? nq : 256, rpb = 256 / cpr;
  const int rl = threadIdx.x / cpr, ql = threadIdx.x - rl * cpr;
  const int s = blockIdx.x * rpb + rl, b = blockIdx.y;
  if (rl >= rpb || s >= S) return;
  const int ml = (int)mel_len[b], sl = (int)src_len[b];
  float* yp = y + ((int64_t)b * S + s) * ldy;
  const int64_t xb = (int64_t)b * T * ldx;
  int start = 0, n = 0;
  if (s < sl && ml > 0) {
    if (ml >= sl) {
      const int q = ml / sl, r = ml % sl;
      n = q + (s < r ? 1 : 0);
      start = s * q + (s < r ? s : r);
    } else {
      const int q = sl / ml, r = sl % ml;
      const int f = (s < r * (q + 1)) ? s / (q + 1) : r + (s - r * (q + 1)) / q;
      start = f; n = 1;
    }
  }
  for (int q4 = ql; q4 < nq; q4 += cpr) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int f0 = 0; f0 < n; f0 += 4) {
      typename Raw4<X16>::T rv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) rv[u] = raw4_load<X16>(x, xb + (int64_t)(start + (f0 + u < n ? f0 + u : n - 1)) * ldx + q4 * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = raw4_f32(rv[u]);
        if (f0 + u < n) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
      }
    }
    if (n > 1) { const float d = (float)n; acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d; }
    *reinterpret_cast<float4*>(yp + q4 * 4) = acc;
  }
}

static inline int mel_cal_rows_per_block(int C) { const int nq = C >> 2; return 256 / (nq < 256 ? nq : 256); }

// io_flags & STYLER_IO_X_BF16: x is stored as bf16 (ldx in elements) -- throughput mode keeps the concatenated output of
// the AudioEncoder's conv stacks that way (its only reader is this kernel, which averages it in fp32).
extern "C" int styler_mel_calibrate_io(const void* x, int64_t ldx, float* y, int64_t ldy, const int64_t* mel_len,
                                       const int64_t* src_len, int B, int T, int S, int C, int io_flags, void* stream) {
  if (!x || !y || !mel_len || !src_len || B <= 0 || T <= 0 || S <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  if ((ldx & 3) || (ldy & 3)) return STYLER_EALIGN;
  const int rpb = mel_cal_rows_per_block(C);
  if (io_flags & STYLER_IO_X_BF16)
    hipLaunchKernelGGL(mel_calibrate_kernel<true>, dim3((S + rpb - 1) / rpb, B), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                       mel_len, src_len, T, S, C);
  else
    hipLaunchKernelGGL(mel_calibrate_kernel<false>, dim3((S + rpb - 1) / rpb, B), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                       mel_len, src_len, T, S, C);
  return launch_status();
}

extern "C" int styler_mel_calibrate(const float* x, int64_t ldx, float* y, int64_t ldy, const int64_t* mel_len,
                                    const int64_t* src_len, int B, int T, int S, int C, void* stream) {
  return styler_mel_calibrate_io(x, ldx, y, ldy, mel_len, src_len, B, T, S, C, 0, stream);
}

// ---- augmentation classifier tail ------------------------------------------------------------
// block per item; wave per row (strided); LN(256) -> ReLU -> Linear(256,2) -> log_softmax; mean over S
__global__ __launch_bounds__(256) void aug_tail_kernel(const float* __restrict__ h, const float* __restrict__ g,
                                                       const float* __restrict__ bt, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ out,
                                                       float* __restrict__ rows_out, int S) {
  __shared__ float part[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
  const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4);
  const float4 bb = *reinterpret_cast<const float4*>(bt + lane * 4);
  const float4 w0 = *reinterpret_cast<const float4*>(w2 + lane * 4);
  const float4 w1 = *reinterpret_cast<const float4*>(w2 + 256 + lane * 4);
  float a0 = 0.f, a1 = 0.f;
  for (int s = wave; s < S; s += 4) {
    const float4 v = *reinterpret_cast<const float4*>(h + ((int64_t)b * S + s) * 256 + lane * 4);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float ox = fmaxf(dx * rstd * gg.x + bb.x, 0.f), oy = fmaxf(dy * rstd * gg.y + bb.y, 0.f);
    const float oz = fmaxf(dz * rstd * gg.z + bb.z, 0.f), ow = fmaxf(dw * rstd * gg.w + bb.w, 0.f);
    const float z0 = wave_sum(ox * w0.x + oy * w0.y + oz * w0.z + ow * w0.w) + b2[0];
    const float z1 = wave_sum(ox * w1.x + oy * w1.y + oz * w1.z + ow * w1.w) + b2[1];
    const float m = fmaxf(z0, z1);
    const float lse = m + logf(expf(z0 - m) + expf(z1 - m));
    a0 += z0 - lse; a1 += z1 - lse;
    if (rows_out && lane == 0) {
      rows_out[((int64_t)b * S + s) * 2 + 0] = z0 - lse;
      rows_out[((int64_t)b * S + s) * 2 + 1] = z1 - lse;
    }
  }
  if (lane == 0) { part[wave][0] = a0; part[wave][1] = a1; }
  __syncthreads();
  if (threadIdx.x < 2)
    out[b * 2 + threadIdx.x] =
        (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / (float)S;
}

extern "C" int styler_aug_classifier_tail(const float* h, const float* ln_g, const float* ln_b, const float* w2,
                                          const float* b2, float* out, int B, int S, void* stream) {
  if (!h || !ln_g || !ln_b || !w2 || !b2 || !out || B <= 0 || S <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(aug_tail_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, h, ln_g, ln_b, w2, b2, out,
                     (float*)nullptr, S);
  return launch_status();
}

// ---- bucketise + embeddings + 4-way add --------------------------------------------------------
// wave per frame.  bucketize(v, bins, right=False) = #{bins < v}: 255 bins, lane holds 4 (256th = +inf),
// count via ballots.
__device__ __forceinline__ int bucketize255(float v, const float* __restrict__ bins, int lane) {
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = k * 64 + lane;
    const float bv = (i < 255) ? bins[i] : INFINITY;
    cnt += __popcll(__ballot(bv < v));
  }
  return cnt;
}

__global__ __launch_bounds__(256) void bucket_embed_add_kernel(
    const float* __restrict__ text, int64_t ldt, const float* __restrict__ speaker, int64_t lds,
    const float* __restrict__ p, float p_scale, const float* __restrict__ e, float e_scale,
    const float* __restrict__ pitch_bins, const float* __restrict__ energy_bins,
    const float* __restrict__ pitch_emb, const float* __restrict__ energy_emb, float* __restrict__ out,
    const float* __restrict__ noise, int64_t ldn, float* __restrict__ out2, int32_t* __restrict__ p_ids,
    int32_t* __restrict__ e_ids, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int pi = bucketize255(p[row] * p_scale, pitch_bins, lane);
  const int ei = bucketize255(e[row] * e_scale, energy_bins, lane);
  if (lane == 0) { if (p_ids) p_ids[row] = pi; if (e_ids) e_ids[row] = ei; }
  const float4 a = *reinterpret_cast<const float4*>(text + row * ldt + lane * 4);
  const float4 s = *reinterpret_cast<const float4*>(speaker + row * lds + lane * 4);
  const float4 pe = *reinterpret_cast<const float4*>(pitch_emb + (int64_t)pi * 256 + lane * 4);
  const float4 ee = *reinterpret_cast<const float4*>(energy_emb + (int64_t)ei * 256 + lane * 4);
  // reference order: text + pitch_embedding + speaker + energy_embedding (modules.py:385)
  float4 o = make_float4(((a.x + pe.x) + s.x) + ee.x, ((a.y + pe.y) + s.y) + ee.y, ((a.z + pe.z) + s.z) + ee.z,
                         ((a.w + pe.w) + s.w) + ee.w);
  *reinterpret_cast<float4*>(out + row * 256 + lane * 4) = o;
  if (out2) {
    const float4 nz = *reinterpret_cast<const float4*>(noise + row * ldn + lane * 4);
    *reinterpret_cast<float4*>(out2 + row * 256 + lane * 4) = make_float4(o.x + nz.x, o.y + nz.y, o.z + nz.z, o.w + nz.w);
  }
}

extern "C" int styler_bucket_embed_add(const float* text, int64_t ldt, const float* speaker, int64_t lds,
                                       const float* p, float p_scale, const float* e, float e_scale,
                                       const float* pitch_bins, const float* energy_bins, const float* pitch_emb,
                                       const float* energy_emb, float* out, const float* noise, int64_t ldn,
                                       float* out2, int32_t* p_ids, int32_t* e_ids, int B, int T, void* stream) {
  if (!text || !speaker || !p || !e || !pitch_bins || !energy_bins || !pitch_emb || !energy_emb || !out || B <= 0 ||
      T <= 0)
    return STYLER_EINVAL;
  if (out2 && !noise) return STYLER_EINVAL;
  if ((ldt & 3) || (lds & 3) || (out2 && (ldn & 3))) return STYLER_EALIGN;
  const int64_t rows = (int64_t)B * T;
  hipLaunchKernelGGL(bucket_embed_add_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     text, ldt, speaker, lds, p, p_scale, e, e_scale, pitch_bins, energy_bins, pitch_emb, energy_emb,
                     out, noise, ldn, out2, p_ids, e_ids, rows);
  return launch_status();
}

// ---- strided adds ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add2_kernel(const float* __restrict__ a, int64_t lda,
                                                   const float* __restrict__ b, int64_t ldb, int64_t b_row_div,
                                                   float* __restrict__ y, int64_t ldy, int64_t rows, int C) {
  const int nq = C / 4;
  const int64_t total = rows * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a) v = *reinterpret_cast<const float4*>(a + row * lda + q * 4);
    if (b) {
      const float4 w = *reinterpret_cast<const float4*>(b + (row / b_row_div) * ldb + q * 4);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    *reinterpret_cast<float4*>(y + row * ldy + q * 4) = v;
  }
}

extern "C" int styler_add2(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy,
                           int64_t rows, int C, void* stream) {
  if (!a || !y || rows <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  if ((lda & 3) || (ldy & 3) || (b && (ldb & 3))) return STYLER_EALIGN;
  hipLaunchKernelGGL(add2_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                     (int64_t)1, y, ldy, rows, C);
  return launch_status();
}

// ---- the decoder-input concatenation of StyleModeling.forward (modules.py:335-350) in ONE launch (round 6) ----------------------------
//   enc[row] = [ text | pitch_up + neck_up | speaker[row / S] | neck_up + energy_up | residual_up ]   (5 x 256 channels)
//   dp[row]  = neck_up + duration_up                                                              (the duration predictor's input)
// Was: three add2, two add_rowvec and the concatenation's copy -- six nodes of >= 4.7 us each on the step's serial chain.
struct StyleCatArgs {
  const float* te; const float* pu; const float* tnu; const float* spk; const float* eu; const float* ru; const float* du;
  float* enc; float* dp;
  int64_t rows; int32_t S;
};
__global__ __launch_bounds__(256) void style_cat_kernel(const StyleCatArgs a) {
  const int64_t total = a.rows * 384;                  // 320 float4 of enc + 64 of dp per row
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / 384;
    const int q = (int)(i - row * 384), part = q >> 6, c = (q & 63) * 4;
    const int64_t o = row * 256 + c;
    float4 v;
    if (part == 0) v = *reinterpret_cast<const float4*>(a.te + o);
    else if (part == 2) v = *reinterpret_cast<const float4*>(a.spk + (row / a.S) * 256 + c);
    else if (part == 4) v = *reinterpret_cast<const float4*>(a.ru + o);
    else {
      const float4 n = *reinterpret_cast<const float4*>(a.tnu + o);
      const float4 w = *reinterpret_cast<const float4*>((part == 1 ? a.pu : part == 3 ? a.eu : a.du) + o);
      v = part == 5 ? make_float4(n.x + w.x, n.y + w.y, n.z + w.z, n.w + w.w)          // dp_in = neck_up + duration_up
                    : part == 1 ? make_float4(w.x + n.x, w.y + n.y, w.z + n.z, w.w + n.w)   // pitch_up + neck_up
                                : make_float4(n.x + w.x, n.y + w.y, n.z + w.z, n.w + w.w);  // neck_up + energy_up
    }
    if (part == 5) *reinterpret_cast<float4*>(a.dp + o) = v;
    else *reinterpret_cast<float4*>(a.enc + row * 1280 + part * 256 + c) = v;
  }
}
extern "C" int styler_style_cat(const float* te, const float* pu, const float* tnu, const float* spk, const float* eu,
                                const float* ru, const float* du, float* enc, float* dp, int B, int S, void* stream) {
  if (!te || !pu || !tnu || !spk || !eu || !ru || !du || !enc || !dp || B <= 0 || S <= 0) return STYLER_EINVAL;
  StyleCatArgs a{te, pu, tnu, spk, eu, ru, du, enc, dp, (int64_t)B * S, S};
  hipLaunchKernelGGL(style_cat_kernel, dim3(grid_for(a.rows * 384)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}
// out = a + b + c over [rows, C] views (row strides in elements): the neck's gradient of the node above
__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                   const float* __restrict__ c, int64_t ldc, float* __restrict__ y, int64_t ldy,
                                                   int64_t rows, int C) {
  const int nq = C / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * nq; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    const float4 u = *reinterpret_cast<const float4*>(a + row * lda + q * 4), v = *reinterpret_cast<const float4*>(b + row * ldb + q * 4);
    const float4 w = *reinterpret_cast<const float4*>(c + row * ldc + q * 4);
    *reinterpret_cast<float4*>(y + row * ldy + q * 4) = make_float4((u.x + v.x) + w.x, (u.y + v.y) + w.y, (u.z + v.z) + w.z, (u.w + v.w) + w.w);
  }
}
extern "C" int styler_add3(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c, int64_t ldc, float* y, int64_t ldy,
                           int64_t rows, int C, void* stream) {
  if (!a || !b || !c || !y || rows <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  if ((lda & 3) || (ldb & 3) || (ldc & 3) || (ldy & 3)) return STYLER_EALIGN;
  hipLaunchKernelGGL(add3_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, c, ldc, y, ldy, rows, C);
  return launch_status();
}

extern "C" int styler_add_rowvec(const float* a, int64_t lda, const float* v, int64_t ldv, float* y, int64_t ldy,
                                 int B, int L, int C, void* stream) {
  if (!v || !y || B <= 0 || L <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  if ((a && (lda & 3)) || (ldy & 3) || (ldv & 3)) return STYLER_EALIGN;
  const int64_t rows = (int64_t)B * L;
  hipLaunchKernelGGL(add2_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, v, ldv,
                     (int64_t)L, y, ldy, rows, C);
  return launch_status();
}

// ---- up to 8 strided row copies in ONE launch (round 4: the launch tail) -------------------------------------------
// Segment k copies rows_k rows of C_k floats from src_k (row stride lds_k; NULL = zeros) to dst_k (row stride ldd_k).  The
// segments travel BY VALUE in the kernel arguments (no descriptor table to upload -- inside a captured hipGraph a table
// upload is a memcpy node with 6-8 us of idle time in front of it).  Replaces the per-slice copy launches of the
// concatenations / gathered slice gradients of autograd.py (SplitBatchFn: two aten copy_ = two memcpy nodes per call).
struct CopySegs {
  const float* src[8];
  float* dst[8];
  int64_t lds[8], ldd[8], rows[8];
  int32_t C[8];
  int32_t n;
};
__global__ __launch_bounds__(256) void copy_rows_multi_kernel(const CopySegs g) {
  const int k = blockIdx.y;
  if (k >= g.n) return;
  const float* __restrict__ src = g.src[k];
  float* __restrict__ dst = g.dst[k];
  const int nq = g.C[k] / 4;
  const int64_t total = g.rows[k] * nq, lds = g.lds[k], ldd = g.ldd[k];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    const float4 v = src ? *reinterpret_cast<const float4*>(src + row * lds + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(dst + row * ldd + q * 4) = v;
  }
}

extern "C" int styler_copy_rows_multi(const StylerCopySeg* segs, int count, void* stream) {
  if (!segs || count <= 0 || count > 8) return STYLER_EINVAL;
  CopySegs g;
  int64_t most = 0;
  for (int k = 0; k < count; ++k) {
    const StylerCopySeg& s = segs[k];
    if (!s.dst || s.rows <= 0 || s.C <= 0 || (s.C & 3)) return STYLER_EINVAL;
    if ((s.ld_src & 3) || (s.ld_dst & 3) || ((uintptr_t)s.src & 15) || ((uintptr_t)s.dst & 15)) return STYLER_EALIGN;
    g.src[k] = reinterpret_cast<const float*>(s.src); g.dst[k] = reinterpret_cast<float*>(s.dst);
    g.lds[k] = s.ld_src; g.ldd[k] = s.ld_dst; g.rows[k] = s.rows; g.C[k] = s.C;
    const int64_t t = s.rows * (s.C / 4);
    most = t > most ? t : most;
  }
  g.n = count;
  int64_t bx = (most + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(copy_rows_multi_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, g);
  return launch_status();
}

// ---- bf16x3 operand split (round 4) ------------------------------------------------------------------------------------
// The parity-grade arithmetic on the bf16 matrix cores: a value is carried as hi + lo (hi = bf16(v), lo = bf16(v - hi): 16
// mantissa bits) and a product a * w as a_hi w_hi + a_hi w_lo + a_lo w_hi (fp32 accumulate; the dropped a_lo w_lo term is
// 2^-18 relative).  The three products are ONE bf16 GEMM over a contraction axis three times as long: the activation
// blocks (hi, lo, hi) against the weight blocks [w_hi | w_hi | w_lo] (runtime.x3), so every bf16 GEMM engine of the library --
// 128 x 128, 256 x 256 LDS-DMA, the grouped kernel -- computes it unchanged.  Two storage forms of the activation:
//   compact (C % 64 == 0; STYLER_IO_X3A at the GEMM, which reads the hi block a second time for the third product):
//     y[row, 0:C] = bf16(x[row]);  y[row, C:2C] = bf16(x[row] - float(bf16(x[row])))                 (4 bytes per element)
//   triple (any C % 4 == 0):  y[row, 0:C] = y[row, 2C:3C] = hi,  y[row, C:2C] = lo                      (6 bytes per element)
// `count` (optional, device): rows at or past count[0] are skipped (packed rows: only the valid prefix is ever read).
__global__ __launch_bounds__(256) void split3_bf16_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y,
                                                          int64_t rows, int C, const int64_t* __restrict__ count, int parts) {
  const int nq = C / 4;
  if (count && count[0] < rows) rows = count[0];
  const int64_t total = rows * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + q * 4);
    const uint32_t h01 = cvt_pk_bf16_rne(v.x, v.y), h23 = cvt_pk_bf16_rne(v.z, v.w);
    const float l0 = v.x - __uint_as_float(h01 << 16), l1 = v.y - __uint_as_float(h01 & 0xffff0000u);
    const float l2 = v.z - __uint_as_float(h23 << 16), l3 = v.w - __uint_as_float(h23 & 0xffff0000u);
    uint16_t* yr = y + row * (int64_t)(parts * C) + q * 4;
    const uint2 hi = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(yr) = hi;
    *reinterpret_cast<uint2*>(yr + C) = make_uint2(cvt_pk_bf16_rne(l0, l1), cvt_pk_bf16_rne(l2, l3));
    if (parts == 3) *reinterpret_cast<uint2*>(yr + 2 * C) = hi;
  }
}

// Round 5: producers write the split themselves.  The next PRODUCER call of this host thread -- styler_conv_gemm[_packed] (fp32
// output, bf16 MFMA mode; any engine, incl. the split-K combine pass), styler_add_layernorm, styler_layernorm_bwd (dx),
// styler_groupnorm_relu[_bwd], styler_batchnorm_train / _bwd, styler_attention_fwd_x3 / _bwd_x3 (fp32 outputs, contiguous rows)
// -- ALSO stores the [hi | lo (| hi)] bf16 split of its fp32 output
// rows into y3: rows of parts * C bf16 (parts = 2 | 3), the layout and the values of styler_split3_bf16 bit for bit, so that
// the GEMM consuming the output as a bf16x3 operand needs no split pass (116 passes = 1.5 ms of the 21 ms bf16x3 step in round
// 4).  The registration is consumed by that call (taken first thing, whatever path it then runs); a producer that cannot honour
// it (bf16 output, strided rows) returns STYLER_EINVAL.  y3 = NULL clears a registration.
static thread_local uint16_t* t_y3 = nullptr;
static thread_local int t_y3_parts = 0;
extern "C" int styler_set_x3_out(void* y3, int parts) {
  if (y3 && ((parts != 2 && parts != 3) || ((uintptr_t)y3 & 7))) return STYLER_EINVAL;
  t_y3 = reinterpret_cast<uint16_t*>(y3); t_y3_parts = y3 ? parts : 0;
  return 0;
}
void styler_take_x3_out(uint16_t** y3, int* parts) {
  *y3 = t_y3; *parts = t_y3_parts;
  t_y3 = nullptr; t_y3_parts = 0;
}

// parts: 3 = the triple form [hi | lo | hi], 2 = the compact form [hi | lo] (see above)
extern "C" int styler_split3_bf16(const float* x, int64_t ldx, void* y, int64_t rows, int C, const int64_t* count, int parts,
                                  void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (C & 3) || (parts != 2 && parts != 3)) return STYLER_EINVAL;
  if ((ldx & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return STYLER_EALIGN;
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     reinterpret_cast<uint16_t*>(y), rows, C, count, parts);
  return launch_status();
}

// Up to 8 of those splits in ONE launch (segments by value in the kernel arguments, as styler_copy_rows_multi): the S-domain
// of the bf16x3 step splits dozens of [B S, 256..1280] tensors of a few microseconds each.  seg.src fp32 rows (ld_src),
// seg.dst the bf16 split (contiguous rows of parts * C), seg._pad = parts (2 or 3).
struct SplitSegs {
  const float* src[8];
  uint16_t* dst[8];
  int64_t lds[8], rows[8];
  int32_t C[8], parts[8];
  int32_t n;
};
__global__ __launch_bounds__(256) void split3_multi_kernel(const SplitSegs g) {
  const int k = blockIdx.y;
  if (k >= g.n) return;
  const float* __restrict__ x = g.src[k];
  uint16_t* __restrict__ y = g.dst[k];
  const int C = g.C[k], parts = g.parts[k], nq = C / 4;
  const int64_t total = g.rows[k] * nq, ldx = g.lds[k];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + q * 4);
    const uint32_t h01 = cvt_pk_bf16_rne(v.x, v.y), h23 = cvt_pk_bf16_rne(v.z, v.w);
    const float l0 = v.x - __uint_as_float(h01 << 16), l1 = v.y - __uint_as_float(h01 & 0xffff0000u);
    const float l2 = v.z - __uint_as_float(h23 << 16), l3 = v.w - __uint_as_float(h23 & 0xffff0000u);
    uint16_t* yr = y + row * (int64_t)(parts * C) + q * 4;
    const uint2 hi = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(yr) = hi;
    *reinterpret_cast<uint2*>(yr + C) = make_uint2(cvt_pk_bf16_rne(l0, l1), cvt_pk_bf16_rne(l2, l3));
    if (parts == 3) *reinterpret_cast<uint2*>(yr + 2 * C) = hi;
  }
}

extern "C" int styler_split3_multi(const StylerCopySeg* segs, int count, void* stream) {
  if (!segs || count <= 0 || count > 8) return STYLER_EINVAL;
  SplitSegs g;
  int64_t most = 0;
  for (int k = 0; k < count; ++k) {
    const StylerCopySeg& s = segs[k];
    if (!s.src || !s.dst || s.rows <= 0 || s.C <= 0 || (s.C & 3) || (s._pad != 2 && s._pad != 3)) return STYLER_EINVAL;
    if ((s.ld_src & 3) || ((uintptr_t)s.src & 15) || ((uintptr_t)s.dst & 7)) return STYLER_EALIGN;
    g.src[k] = reinterpret_cast<const float*>(s.src); g.dst[k] = reinterpret_cast<uint16_t*>(s.dst);
    g.lds[k] = s.ld_src; g.rows[k] = s.rows; g.C[k] = s.C; g.parts[k] = s._pad;
    const int64_t t = s.rows * (s.C / 4);
    most = t > most ? t : most;
  }
  g.n = count;
  int64_t bx = (most + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(split3_multi_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, g);
  return launch_status();
}

// y = x - float(bf16(x)), rounded to bf16 and stored as fp32 (exactly representable): the low part of an operand of the
// weight-gradient GEMMs in the bf16x3 arithmetic, in the format every wgrad kernel variant accepts.
__global__ __launch_bounds__(256) void lo_part_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                      int64_t rows, int C, const int64_t* __restrict__ count) {
  const int nq = C / 4;
  if (count && count[0] < rows) rows = count[0];
  const int64_t total = rows * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + q * 4);
    float4 o;
    o.x = __uint_as_float(f32_to_bf16_bits(bf16_lo_part(v.x)) << 16); o.y = __uint_as_float(f32_to_bf16_bits(bf16_lo_part(v.y)) << 16);
    o.z = __uint_as_float(f32_to_bf16_bits(bf16_lo_part(v.z)) << 16); o.w = __uint_as_float(f32_to_bf16_bits(bf16_lo_part(v.w)) << 16);
    *reinterpret_cast<float4*>(y + row * (int64_t)C + q * 4) = o;
  }
}

extern "C" int styler_lo_part(const float* x, int64_t ldx, float* y, int64_t rows, int C, const int64_t* count, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  if ((ldx & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return STYLER_EALIGN;
  hipLaunchKernelGGL(lo_part_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, rows, C,
                     count);
  return launch_status();
}

// ---- masked error sums ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masked_err_kernel(const float* __restrict__ a, int64_t lda,
                                                         const float* __restrict__ b, int64_t ldb,
                                                         double* __restrict__ acc, int kind, int64_t rows, int L,
                                                         int C, const int64_t* __restrict__ len) {
  __shared__ double red[4][2];
  double s = 0.0, n = 0.0;
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / C; const int c = (int)(i - row * C);
    const int64_t bb = row / L;
    if (len && (row - bb * L) >= len[bb]) continue;
    const float d = a[row * lda + c] - b[row * ldb + c];
    s += kind == 0 ? (double)d * d : (double)fabsf(d);
    n += 1.0;
  }
  s = wave_sum_d(s); n = wave_sum_d(n);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s; red[wave][1] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], red[0][0] + red[1][0] + red[2][0] + red[3][0]);
    atomicAdd(&acc[1], red[0][1] + red[1][1] + red[2][1] + red[3][1]);
  }
}

extern "C" int styler_masked_err_sum(const float* a, int64_t lda, const float* b, int64_t ldb, double* acc, int kind,
                                     int B, int L, int C, const int64_t* len, void* stream) {
  if (!a || !b || !acc || B <= 0 || L <= 0 || C <= 0 || (kind != 0 && kind != 1)) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  hipLaunchKernelGGL(masked_err_kernel, dim3(grid_for(rows * C, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a, lda,
                     b, ldb, acc, kind, rows, L, C, len);
  return launch_status();
}

// ---- length mask (utils.py:223-232): mask[b,t] = t >= len[b] (True = padding), torch.bool storage ----
__global__ void length_mask_kernel(const int64_t* __restrict__ len, uint8_t* __restrict__ mask, int64_t total, int L) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / L;
    mask[i] = (uint8_t)((i - b * L) >= len[b]);
  }
}

extern "C" int styler_length_mask(const int64_t* len, uint8_t* mask, int B, int L, void* stream) {
  if (!len || !mask || B <= 0 || L <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(length_mask_kernel, dim3(grid_for((int64_t)B * L)), dim3(256), 0, (hipStream_t)stream, len, mask,
                     (int64_t)B * L, L);
  return launch_status();
}

// both masks of a forward (styler.py:42-43: src_mask [B, S], mel_mask [B, T]) in one launch
__global__ void length_mask2_kernel(const int64_t* __restrict__ len0, uint8_t* __restrict__ mask0, int64_t total0, int L0,
                                    const int64_t* __restrict__ len1, uint8_t* __restrict__ mask1, int64_t total1, int L1) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total0 + total1; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < total0) { const int64_t b = i / L0; mask0[i] = (uint8_t)((i - b * L0) >= len0[b]); }
    else { const int64_t j = i - total0, b = j / L1; mask1[j] = (uint8_t)((j - b * L1) >= len1[b]); }
  }
}
extern "C" int styler_length_mask2(const int64_t* len0, uint8_t* mask0, int B0, int L0, const int64_t* len1, uint8_t* mask1, int B1,
                                   int L1, void* stream) {
  if (!len0 || !mask0 || !len1 || !mask1 || B0 <= 0 || L0 <= 0 || B1 <= 0 || L1 <= 0) return STYLER_EINVAL;
  const int64_t t0 = (int64_t)B0 * L0, t1 = (int64_t)B1 * L1;
  hipLaunchKernelGGL(length_mask2_kernel, dim3(grid_for(t0 + t1)), dim3(256), 0, (hipStream_t)stream, len0, mask0, t0, L0, len1,
                     mask1, t1, L1);
  return launch_status();
}

// ---- start of a training step: clear the flat gradient and the norm kernels' statistics slab, advance the dropout step counter ----
// (round 6: three launches -- two torch fills and an int64 add_ -- were three 5-16 us nodes at the head of every step; every node of
// the step's graph costs >= 4.7 us whatever it does)
__global__ __launch_bounds__(256) void step_begin_kernel(float4* __restrict__ a, int64_t na, float4* __restrict__ b, int64_t nb,
                                                         uint64_t* __restrict__ counter) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) a[i] = z;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) b[i] = z;
  if (counter && blockIdx.x == 0 && threadIdx.x == 0) counter[0] += 1;
}

extern "C" int styler_step_begin(void* a, int64_t a_bytes, void* b, int64_t b_bytes, uint64_t* counter, void* stream) {
  if (a_bytes < 0 || b_bytes < 0 || (a_bytes & 15) || (b_bytes & 15) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15)) return STYLER_EALIGN;
  if ((a_bytes && !a) || (b_bytes && !b)) return STYLER_EINVAL;
  const int64_t na = a_bytes / 16, nb = b_bytes / 16, most = na > nb ? na : nb;
  int64_t blocks = (most + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(step_begin_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(a), na,
                     reinterpret_cast<float4*>(b), nb, counter);
  return launch_status();
}

// ---- derived weight layouts: many strided 3-D copies (+ fp32 -> bf16) in ONE launch -------------------------------
// Every kernel-side view of a parameter (bf16 shadow, [n, kw, cin] conv layout, tap-flipped transposed dX layout, fused
// QKV / BiLSTM matrices, summed LSTM biases) is an index permutation of parameter elements, optionally cast or
// summed with a second tensor of the same shape.  One descriptor = one source tensor walked as dims (d0, d1, d2) with element strides on both sides; the
// optimiser refreshes ALL derived layouts of the model with one launch after each update (runtime.Derived).
// flags bit3 (round 4, the bf16x3 arithmetic): the element written is the LOW part of the source value, v - float(bf16(v)) --
// together with the plain bf16 cast (the high part) it represents v to 16 mantissa bits (runtime.gemm_weight's x3 layouts).
__global__ __launch_bounds__(256) void strided_copy_multi_kernel(const StylerCopyDesc* __restrict__ desc, int count,
                                                                 const int32_t* __restrict__ blockmap) {
  int lo = 0, hi = count - 1;                        // last descriptor with block_start <= blockIdx.x
  const int64_t bid = blockIdx.x;
  if (blockmap) lo = blockmap[bid];                  // round 6: the owner of every block precomputed by the host -- the search is
                                                     // eight dependent loads in front of 2-5 KB of work (103 -> 91 us for the model's table)
  else while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].block_start <= bid) lo = mid; else hi = mid - 1; }
  const StylerCopyDesc d = desc[lo];
  if (d.flags & 2) {
    // Tiled transpose (flags bit1, set by the host when the pattern holds): dst is contiguous along a2, the source is
    // contiguous along the MERGED (a0, a1) index m (ss0 == d1, ss1 == +-1) and strided along a2 -- the tap-flipped
    // transposed dX weights [cin, kw, n] <- [n, cin, kw] and the transposed Linear / fused-QKV layouts.  Walked one
    // destination element per thread, a wave touched 64 different cache lines for 256 useful bytes (32x the bytes through
    // L1 / L2: this launch took 300 us for 59 M elements).  Here a block moves a [32 a2] x [64 m] tile through LDS: reads
    // are 256-byte runs along m, writes 32-element runs along a2.
    // Round 6: a block moves a [64 a2] x [64 m] tile (was 32 x 64): 16 independent 256-byte row reads per wave in flight
    // (8 before: the launch ran at 3 TB/s, latency-bound), written as 64-element runs along a2, two bf16 per store.
    __shared__ float tile[64][65];
    const uint32_t M = (uint32_t)d.d0 * (uint32_t)d.d1, N2 = (uint32_t)d.d2;
    const uint32_t mt = (M + 63u) / 64u;
    const uint32_t t = (uint32_t)(bid - d.block_start);
    const uint32_t n0 = (t / mt) * 64u, m0 = (t % mt) * 64u;
    const float* srcp = reinterpret_cast<const float*>(d.src) + (d.ss1 < 0 ? -(int64_t)(d.d1 - 1) : 0);
    const float* src2p = d.src2 ? reinterpret_cast<const float*>(d.src2) + (d.ss1 < 0 ? -(int64_t)(d.d1 - 1) : 0) : nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float vv[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const uint32_t nn = n0 + wave * 16 + rr, m = m0 + lane;
      float v = 0.f;
      if (nn < N2 && m < M) {
        const int64_t si = (int64_t)nn * d.ss2 + m;
        v = srcp[si];
        if (src2p) v += src2p[si];
      }
      vv[rr] = v;
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) tile[wave * 16 + rr][lane] = vv[rr];
    __syncthreads();
    const uint32_t ud1t = (uint32_t)d.d1;
    const bool pair2 = !(N2 & 1u) && !(d.ds0 & 1) && !(d.ds1 & 1) && !((uintptr_t)d.dst & ((d.flags & 1) ? 3 : 7));
    if (pair2) {
      const uint32_t nl = (threadIdx.x & 31) * 2u, ms = threadIdx.x >> 5;     // 32 a2 pairs x 8 m values per pass
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint32_t ml = p * 8 + ms, m = m0 + ml, nn = n0 + nl;
        if (m >= M || nn >= N2) continue;
        const uint32_t a0 = m / ud1t, tt = m % ud1t;
        const uint32_t a1 = d.ss1 < 0 ? ud1t - 1u - tt : tt;
        const int64_t o = (int64_t)a0 * d.ds0 + (int64_t)a1 * d.ds1 + nn;
        float v0 = tile[nl][ml], v1 = tile[nl + 1][ml];
        if (d.flags & 8) { v0 = bf16_lo_part(v0); v1 = bf16_lo_part(v1); }
        if (d.flags & 1) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(d.dst) + o) = pack_bf16x2(v0, v1);
        else *reinterpret_cast<float2*>(reinterpret_cast<float*>(d.dst) + o) = make_float2(v0, v1);
      }
    } else {
      const uint32_t nl = threadIdx.x & 63, ms = threadIdx.x >> 6;            // 64 a2 values x 4 m values per pass
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const uint32_t ml = p * 4 + ms, m = m0 + ml, nn = n0 + nl;
        if (m >= M || nn >= N2) continue;
        const uint32_t a0 = m / ud1t, tt = m % ud1t;
        const uint32_t a1 = d.ss1 < 0 ? ud1t - 1u - tt : tt;
        const int64_t o = (int64_t)a0 * d.ds0 + (int64_t)a1 * d.ds1 + nn;
        float v = tile[nl][ml];
        if (d.flags & 8) v = bf16_lo_part(v);
        if (d.flags & 1) reinterpret_cast<uint16_t*>(d.dst)[o] = (uint16_t)f32_to_bf16_bits(v);
        else reinterpret_cast<float*>(d.dst)[o] = v;
      }
    }
    return;
  }
  if (d.flags & 4) {
    // Tap interleave (flags bit2): dst [a0][a1][a2] <- src[a0 * ss0 + a2 * d1 + a1]: the [n, kw, cin] conv layout from the
    // parameter's [n, cin, kw].  Inside one a0 the source is ONE contiguous run of d1 * d2 elements; a block moves 128
    // a2 values x all d1 taps of one a0 through LDS (read as a contiguous run, written as d1 runs of 128).
    // Round 6: a block takes TWO consecutive a0 (was one: 384-1152 elements per block, one or two loads in flight per thread);
    // every load of the block is issued before the first LDS write.
    __shared__ float run[2][128 * 9 + 8];
    const uint32_t ud1k = (uint32_t)d.d1, ud2k = (uint32_t)d.d2;
    const uint32_t ct = (ud2k + 127u) / 128u;
    const uint32_t t = (uint32_t)(bid - d.block_start);
    const uint32_t a0p = (t / ct) * 2u, c0 = (t % ct) * 128u;
    const uint32_t cn = ud2k - c0 < 128u ? ud2k - c0 : 128u;
    const uint32_t tot = cn * ud1k;                                      // <= 1152: five passes of 256 threads
    float vv[2][5];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const uint32_t a0 = a0p + r;
      const bool okr = a0 < (uint32_t)d.d0;
      const float* sp = reinterpret_cast<const float*>(d.src) + (int64_t)(okr ? a0 : 0) * d.ss0 + (int64_t)c0 * ud1k;
      const float* sp2 = d.src2 ? reinterpret_cast<const float*>(d.src2) + (int64_t)(okr ? a0 : 0) * d.ss0 + (int64_t)c0 * ud1k : nullptr;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const uint32_t i = threadIdx.x + (uint32_t)k * 256u;
        vv[r][k] = (okr && i < tot) ? sp[i] + (sp2 ? sp2[i] : 0.f) : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const uint32_t i = threadIdx.x + (uint32_t)k * 256u;
        if (i < tot) run[r][i] = vv[r][k];
      }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const uint32_t a0 = a0p + r;
      if (a0 >= (uint32_t)d.d0) break;
      const bool pr = !(cn & 1u) && !(c0 & 1u) && !(d.ds0 & 1) && !(d.ds1 & 1) && !((uintptr_t)d.dst & ((d.flags & 1) ? 3 : 7));
      if (pr) {                                                            // two consecutive channels per store
        const uint32_t hn = cn >> 1;
        for (uint32_t i = threadIdx.x; i < hn * ud1k; i += 256u) {
          const uint32_t a1 = i / hn, c = (i - a1 * hn) * 2u;
          float v0 = run[r][c * ud1k + a1], v1 = run[r][(c + 1) * ud1k + a1];
          if (d.flags & 8) { v0 = bf16_lo_part(v0); v1 = bf16_lo_part(v1); }
          const int64_t o = (int64_t)a0 * d.ds0 + (int64_t)a1 * d.ds1 + c0 + c;
          if (d.flags & 1) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(d.dst) + o) = pack_bf16x2(v0, v1);
          else *reinterpret_cast<float2*>(reinterpret_cast<float*>(d.dst) + o) = make_float2(v0, v1);
        }
      } else {
        for (uint32_t i = threadIdx.x; i < tot; i += 256u) {
          const uint32_t a1 = i / cn, c = i - a1 * cn;
          float v = run[r][c * ud1k + a1];
          if (d.flags & 8) v = bf16_lo_part(v);
          const int64_t o = (int64_t)a0 * d.ds0 + (int64_t)a1 * d.ds1 + c0 + c;
          if (d.flags & 1) reinterpret_cast<uint16_t*>(d.dst)[o] = (uint16_t)f32_to_bf16_bits(v);
          else reinterpret_cast<float*>(d.dst)[o] = v;
        }
      }
    }
    return;
  }
  // 32-bit index arithmetic (a descriptor covers < 2^31 elements): the three 64-bit divisions per element that decoded
  // (a0, a1, a2) were what bound this kernel (300 us for the 59 M derived elements of the model)
  const uint32_t total = (uint32_t)d.d0 * (uint32_t)d.d1 * (uint32_t)d.d2;
  const float* src = reinterpret_cast<const float*>(d.src);
  const float* src2 = reinterpret_cast<const float*>(d.src2);
  const uint32_t i0 = (uint32_t)(bid - d.block_start) * 1024u;
  const uint32_t ud1 = (uint32_t)d.d1, ud2 = (uint32_t)d.d2;
  const bool bf = d.flags & 1;
  // destination-contiguous pairs leave as one 4-byte (bf16 x 2) or 8-byte (fp32 x 2) store
  const bool pair = d.ds2 == 1 && !(ud2 & 1u) && !(d.ds0 & 1) && !(d.ds1 & 1) && !((uintptr_t)d.dst & (bf ? 3 : 7));
  if (pair) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t i = i0 + (uint32_t)(k * 256 + threadIdx.x) * 2u;
      if (i >= total) return;
      const uint32_t a2 = i % ud2, r = i / ud2;
      const uint32_t a1 = r % ud1, a0 = r / ud1;
      const int64_t si = (int64_t)a0 * d.ss0 + (int64_t)a1 * d.ss1 + (int64_t)a2 * d.ss2;
      float v0 = src[si], v1 = src[si + d.ss2];
      if (src2) { v0 += src2[si]; v1 += src2[si + d.ss2]; }
      if (d.flags & 8) { v0 = bf16_lo_part(v0); v1 = bf16_lo_part(v1); }
      const int64_t o = (int64_t)a0 * d.ds0 + (int64_t)a1 * d.ds1 + a2;
      if (bf) *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(d.dst) + o) = pack_bf16x2(v0, v1);
      else *reinterpret_cast<float2*>(reinterpret_cast<float*>(d.dst) + o) = make_float2(v0, v1);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = i0 + (uint32_t)(k * 256) + threadIdx.x;
    if (i >= total) return;
    const uint32_t a2 = i % ud2, r = i / ud2;
    const uint32_t a1 = r % ud1, a0 = r / ud1;
    const int64_t si = (int64_t)a0 * d.ss0 + (int64_t)a1 * d.ss1 + (int64_t)a2 * d.ss2;
    float v = src[si];
    if (src2) v += src2[si];
    if (d.flags & 8) v = bf16_lo_part(v);
    const int64_t o = (int64_t)a0 * d.ds0 + (int64_t)a1 * d.ds1 + (int64_t)a2 * d.ds2;
    if (bf) reinterpret_cast<uint16_t*>(d.dst)[o] = (uint16_t)f32_to_bf16_bits(v);
    else reinterpret_cast<float*>(d.dst)[o] = v;
  }
}

extern "C" int styler_strided_copy_multi(const StylerCopyDesc* desc_dev, int count, int64_t total_blocks, void* stream) {
  if (!desc_dev || count <= 0 || total_blocks <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(strided_copy_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, desc_dev,
                     count, (const int32_t*)nullptr);
  return launch_status();
}
extern "C" int styler_strided_copy_multi_map(const StylerCopyDesc* desc_dev, int count, int64_t total_blocks, const int32_t* blockmap,
                                             void* stream) {
  if (!desc_dev || count <= 0 || total_blocks <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(strided_copy_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, desc_dev,
                     count, blockmap);
  return launch_status();
}
