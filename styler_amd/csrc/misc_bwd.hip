// Backward of the memory-bound stitching kernels (misc.hip, length_regulator.hip), the loss gradients,
// dropout and the fused clip + Adam update.
#include "common.h"

static inline unsigned grid_for(int64_t work, int per_block = 256, int cap = 4096) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- embedding backward: demb[text[b,t], :] += dy[b,t,:], padding_idx 0 receives nothing (Models.py:52-53) ----
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ text, const float* __restrict__ dy,
                                                        int64_t lddy, float* __restrict__ demb, int64_t rows, int C) {
  const int64_t total = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / C; const int c = (int)(i - row * C);
    const int64_t id = text[row];
    if (id != 0) atomicAdd(demb + id * C + c, dy[row * lddy + c]);
  }
}

// The same without atomics (round 4): block v OWNS row v of the table -- it scans the token ids (a few thousand, L2 / scalar
// cache), compacts the positions that hold v in token order, and adds their dy rows in that order: bit-reproducible.
__global__ __launch_bounds__(256) void embed_bwd_det_kernel(const int64_t* __restrict__ text, const float* __restrict__ dy,
                                                            int64_t lddy, float* __restrict__ demb, int rows, int C) {
  __shared__ int hits[256];
  __shared__ int wcount[4], nhit;
  const int v = blockIdx.x + 1;                      // (row 0 = padding_idx: no gradient)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};               // channels threadIdx.x + 256 k, C <= 1024
  for (int base = 0; base < rows; base += 256) {
    const int r = base + threadIdx.x;
    const bool hit = r < rows && text[r] == v;
    const uint64_t m = __ballot(hit);
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    if (hit) hits[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
    if (threadIdx.x == 0) nhit = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
    const int n = nhit;
    for (int i = 0; i < n; ++i) {
      const float* row = dy + (int64_t)hits[i] * lddy;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int c = threadIdx.x + 256 * k; if (c < C) acc[k] += row[c]; }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int c = threadIdx.x + 256 * k; if (c < C) demb[(int64_t)v * C + c] += acc[k]; }
}

extern "C" int styler_embed_bwd_det(const int64_t* text, const float* dy, int64_t lddy, float* demb, int B, int L, int C,
                                    int V, void* stream) {
  if (!text || !dy || !demb || B <= 0 || L <= 0 || C <= 0 || C > 1024 || V <= 1 || (int64_t)B * L >= ((int64_t)1 << 31))
    return STYLER_EINVAL;
  hipLaunchKernelGGL(embed_bwd_det_kernel, dim3((unsigned)(V - 1)), dim3(256), 0, (hipStream_t)stream, text, dy, lddy, demb,
                     B * L, C);
  return launch_status();
}

extern "C" int styler_embed_bwd(const int64_t* text, const float* dy, int64_t lddy, float* demb, int B, int L, int C,
                                void* stream) {
  if (!text || !dy || !demb || B <= 0 || L <= 0 || C <= 0) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, text, dy, lddy, demb,
                     rows, C);
  return launch_status();
}

// ---- one-hot conv backward: dw[c, idx[b,t+j-2], j] += dy[b,t,c] (parameter layout [C, 257, 5]); db += colsum ----
__device__ __forceinline__ int quant_index_b(float v) {
  if (v <= 0.f) return 0;
  if (v > 1.f) v = 1.f;
  return (int)rintf(v * 255.f) + 1;
}

// The scatter form contends on index 0 (unvoiced + padded frames); instead the one-hot [rows, 260] (zero padded
// from 257) is materialised once in backward and the gradient is a wgrad GEMM (styler_wgrad, kw = 5, strides of the
// parameter layout [C, 257, 5]) -- 17.6 GFLOP on the MFMA engine instead of ~34 M contended atomics.
__global__ __launch_bounds__(256) void onehot_expand_kernel(const float* __restrict__ v, float* __restrict__ oh,
                                                            int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int idx = quant_index_b(v[row]);
  float* o = oh + row * 260;
  for (int c = lane; c < 260; c += 64) o[c] = (c == idx) ? 1.f : 0.f;
}

extern "C" int styler_onehot_expand(const float* v, float* onehot, int64_t rows, void* stream) {
  if (!v || !onehot || rows <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(onehot_expand_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, v, onehot,
                     rows);
  return launch_status();
}

// ---- mel calibrator backward: grid (T, B) over INPUT frames -----------------------------------------------
// (block = 256 / (C / 4) input frames of one item, thread = (frame, float4 column): see mel_calibrate_kernel)
#define MCB_FR 8
template <bool DX16>                                 // dx (the gradient of the concatenated streams) written as bf16
__global__ __launch_bounds__(256) void mel_calibrate_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                void* __restrict__ dx, int64_t lddx,
                                                                const int64_t* __restrict__ mel_len,
                                                                const int64_t* __restrict__ src_len, int T, int S, int C) {
  // (round 4: a block walks MCB_FR consecutive groups of frames -- at C = 1024 a block was ONE frame, 42 k blocks of one
  //  16-byte load and one store per thread: 54 us for 87 MB)
  const int nq = C >> 2, cpr = nq < 256 ? nq : 256, rpb = 256 / cpr;
  const int rl = threadIdx.x / cpr, ql = threadIdx.x - rl * cpr;
  const int b = blockIdx.y;
  if (rl >= rpb) return;
  const int ml = (int)mel_len[b], sl = (int)src_len[b];
  const float* dyb = dy + (int64_t)b * S * lddy;
  for (int f = 0; f < MCB_FR; ++f) {
    const int t = (blockIdx.x * MCB_FR + f) * rpb + rl;
    if (t >= T) return;
    const int64_t dxo = ((int64_t)b * T + t) * lddx;
    // div > 0: frame t lies in exactly one segment (compression / copy): grad = dy[s0] / div
    // div == 0: frame t was repeated `cnt` times (expansion): grad = sum of those output rows
    int s0 = 0, cnt = 0, div = 0;
    if (t < ml && sl > 0) {
      if (ml >= sl) {
        const int q = ml / sl, r = ml % sl;
        s0 = (t < r * (q + 1)) ? t / (q + 1) : r + (t - r * (q + 1)) / q;
        div = q + (s0 < r ? 1 : 0);
      } else {
        const int q = sl / ml, r = sl % ml;
        cnt = q + (t < r ? 1 : 0);
        s0 = t * q + (t < r ? t : r);
      }
    }
    const int n = div > 0 ? 1 : cnt;                    // rows of dy that reach this frame
    for (int q4 = ql; q4 < nq; q4 += cpr) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k0 = 0; k0 < n; k0 += 4) {
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = *reinterpret_cast<const float4*>(dyb + (int64_t)(s0 + (k0 + u < n ? k0 + u : n - 1)) * lddy + q4 * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k0 + u < n) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
      }
      if (div > 1) { const float d = (float)div; acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d; }
      stg4(dx, dxo + q4 * 4, acc, DX16);
    }
  }
}

// io_flags & STYLER_IO_Y_BF16: dx is written as bf16 (lddx in elements; its reader, the GroupNorm backward, takes a bf16 dy).
extern "C" int styler_mel_calibrate_bwd_io(const float* dy, int64_t lddy, void* dx, int64_t lddx, const int64_t* mel_len,
                                           const int64_t* src_len, int B, int T, int S, int C, int io_flags, void* stream) {
  if (!dy || !dx || !mel_len || !src_len || B <= 0 || T <= 0 || S <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  if ((lddy & 3) || (lddx & 3)) return STYLER_EALIGN;
  const int nq = C >> 2, rpb = 256 / (nq < 256 ? nq : 256);
  if (io_flags & STYLER_IO_Y_BF16)
    hipLaunchKernelGGL(mel_calibrate_bwd_kernel<true>, dim3((T + rpb * MCB_FR - 1) / (rpb * MCB_FR), B), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx,
                       lddx, mel_len, src_len, T, S, C);
  else
    hipLaunchKernelGGL(mel_calibrate_bwd_kernel<false>, dim3((T + rpb * MCB_FR - 1) / (rpb * MCB_FR), B), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx,
                       lddx, mel_len, src_len, T, S, C);
  return launch_status();
}

extern "C" int styler_mel_calibrate_bwd(const float* dy, int64_t lddy, float* dx, int64_t lddx, const int64_t* mel_len,
                                        const int64_t* src_len, int B, int T, int S, int C, void* stream) {
  return styler_mel_calibrate_bwd_io(dy, lddy, dx, lddx, mel_len, src_len, B, T, S, C, 0, stream);
}

// ---- augmentation classifier tail backward ----------------------------------------------------------------
// forward: y = relu(LN(h)*g + b); z = W2 y + b2; out[b] = mean_s log_softmax(z).  Given dout [B,2].
__global__ __launch_bounds__(256) void aug_tail_bwd_kernel(const float* __restrict__ h, const float* __restrict__ g,
                                                           const float* __restrict__ bt, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, const float* __restrict__ dout,
                                                           float* __restrict__ dh, float* __restrict__ dg,
                                                           float* __restrict__ dbt, float* __restrict__ dw2,
                                                           float* __restrict__ db2, int S, int seg_rows, int pslots) {
  // grid (B, segments of the S axis): 48 blocks walking 60 rows each were a 45 us latency chain (6 wave reductions per row)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
  const int s_lo = blockIdx.y * seg_rows, s_hi = min(S, s_lo + seg_rows);
  const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4);
  const float4 bb = *reinterpret_cast<const float4*>(bt + lane * 4);
  const float4 w0 = *reinterpret_cast<const float4*>(w2 + lane * 4);
  const float4 w1 = *reinterpret_cast<const float4*>(w2 + 256 + lane * 4);
  const float go0 = dout[b * 2] / (float)S, go1 = dout[b * 2 + 1] / (float)S;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag, aw0 = ag, aw1 = ag;
  float ab0 = 0.f, ab1 = 0.f;
  for (int s = s_lo + wave; s < s_hi; s += 4) {
    const float4 v = *reinterpret_cast<const float4*>(h + ((int64_t)b * S + s) * 256 + lane * 4);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float cx = v.x - mean, cy = v.y - mean, cz = v.z - mean, cw = v.w - mean;
    const float var = wave_sum(cx * cx + cy * cy + cz * cz + cw * cw) * (1.f / 256.f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float hx = cx * rstd, hy = cy * rstd, hz = cz * rstd, hw = cw * rstd;
    const float px = hx * gg.x + bb.x, py = hy * gg.y + bb.y, pz = hz * gg.z + bb.z, pw = hw * gg.w + bb.w;
    const float ox = fmaxf(px, 0.f), oy = fmaxf(py, 0.f), oz = fmaxf(pz, 0.f), ow = fmaxf(pw, 0.f);
    const float z0 = wave_sum(ox * w0.x + oy * w0.y + oz * w0.z + ow * w0.w) + b2[0];
    const float z1 = wave_sum(ox * w1.x + oy * w1.y + oz * w1.z + ow * w1.w) + b2[1];
    const float m = fmaxf(z0, z1);
    const float e0 = expf(z0 - m), e1 = expf(z1 - m), inv = 1.f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    // d log_softmax: dz_k = go_k - p_k * (go_0 + go_1)
    const float dz0 = go0 - p0 * (go0 + go1), dz1 = go1 - p1 * (go0 + go1);
    ab0 += dz0; ab1 += dz1;
    aw0.x += dz0 * ox; aw0.y += dz0 * oy; aw0.z += dz0 * oz; aw0.w += dz0 * ow;
    aw1.x += dz1 * ox; aw1.y += dz1 * oy; aw1.z += dz1 * oz; aw1.w += dz1 * ow;
    float dx_ = px > 0.f ? dz0 * w0.x + dz1 * w1.x : 0.f, dy_ = py > 0.f ? dz0 * w0.y + dz1 * w1.y : 0.f;
    float dz_ = pz > 0.f ? dz0 * w0.z + dz1 * w1.z : 0.f, dw_ = pw > 0.f ? dz0 * w0.w + dz1 * w1.w : 0.f;
    ag.x += dx_ * hx; ag.y += dy_ * hy; ag.z += dz_ * hz; ag.w += dw_ * hw;
    ab.x += dx_; ab.y += dy_; ab.z += dz_; ab.w += dw_;
    const float ex = dx_ * gg.x, ey = dy_ * gg.y, ez = dz_ * gg.z, ew = dw_ * gg.w;
    const float m1 = wave_sum(ex + ey + ez + ew) * (1.f / 256.f);
    const float m2 = wave_sum(ex * hx + ey * hy + ez * hz + ew * hw) * (1.f / 256.f);
    *reinterpret_cast<float4*>(dh + ((int64_t)b * S + s) * 256 + lane * 4) =
        make_float4(rstd * (ex - m1 - hx * m2), rstd * (ey - m1 - hy * m2), rstd * (ez - m1 - hz * m2),
                    rstd * (ew - m1 - hw * m2));
  }
  // block-level reduction over the 4 waves, then one atomic per (vector, channel) and block
  __shared__ float red[4][4][256];
  __shared__ float redb[4][2];
  const float* srcs[4] = {&ag.x, &ab.x, &aw0.x, &aw1.x};
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[k][wave][lane * 4 + e] = srcs[k][e];
  if (lane == 0) { redb[wave][0] = ab0; redb[wave][1] = ab1; }
  __syncthreads();
  float* dsts[4] = {dg, dbt, dw2, dw2 + 256};
  const int c = threadIdx.x;
  if (pslots) {
    // STYLER_IO_PARAM_SLOTS: the four gradients are slot arrays [blocks][256 | 256 | 512 | 2]; this block STORES its sums into
    // slot blockIdx.y * gridDim.x + b, the caller's multi-tensor reduce folds the slots in order (no atomics)
    const int64_t blk = (int64_t)blockIdx.y * gridDim.x + b;
    dsts[0] = dg + blk * 256; dsts[1] = dbt + blk * 256; dsts[2] = dw2 + blk * 512; dsts[3] = dw2 + blk * 512 + 256;
#pragma unroll
    for (int k = 0; k < 4; ++k) dsts[k][c] = (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]);
    if (threadIdx.x < 2)
      db2[blk * 2 + threadIdx.x] = (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]);
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    atomicAdd(dsts[k] + c, (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]));
  if (threadIdx.x < 2) atomicAdd(db2 + threadIdx.x, (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]));
}

extern "C" int styler_aug_classifier_tail_bwd(const float* h, const float* ln_g, const float* ln_b, const float* w2,
                                              const float* b2, const float* dout, float* dh, float* dln_g,
                                              float* dln_b, float* dw2, float* db2, int B, int S, void* stream) {
  if (!h || !ln_g || !ln_b || !w2 || !b2 || !dout || !dh || !dln_g || !dln_b || !dw2 || !db2 || B <= 0 || S <= 0)
    return STYLER_EINVAL;
  const int seg_rows = 16;
  hipLaunchKernelGGL(aug_tail_bwd_kernel, dim3(B, (S + seg_rows - 1) / seg_rows), dim3(256), 0, (hipStream_t)stream, h, ln_g,
                     ln_b, w2, b2, dout, dh, dln_g, dln_b, dw2, db2, S, seg_rows, 0);
  return launch_status();
}

// The same with the four parameter gradients as SLOT arrays (io_flags & STYLER_IO_PARAM_SLOTS): [slots][256], [slots][256],
// [slots][512], [slots][2] with slots = styler_aug_classifier_tail_slots(B, S); every slot is stored.
extern "C" int styler_aug_classifier_tail_slots(int B, int S) { return B * ((S + 15) / 16); }
extern "C" int styler_aug_classifier_tail_bwd_io(const float* h, const float* ln_g, const float* ln_b, const float* w2,
                                                 const float* b2, const float* dout, float* dh, float* dln_g,
                                                 float* dln_b, float* dw2, float* db2, int B, int S, int io_flags,
                                                 void* stream) {
  if (!h || !ln_g || !ln_b || !w2 || !b2 || !dout || !dh || !dln_g || !dln_b || !dw2 || !db2 || B <= 0 || S <= 0)
    return STYLER_EINVAL;
  const int seg_rows = 16;
  hipLaunchKernelGGL(aug_tail_bwd_kernel, dim3(B, (S + seg_rows - 1) / seg_rows), dim3(256), 0, (hipStream_t)stream, h, ln_g,
                     ln_b, w2, b2, dout, dh, dln_g, dln_b, dw2, db2, S, seg_rows, (io_flags & STYLER_IO_PARAM_SLOTS) ? 1 : 0);
  return launch_status();
}

// ---- LengthRegulator backward: dx[b,i,:] = sum of dy over the frames of phoneme i (contiguous segment) ----
__global__ __launch_bounds__(256) void length_regulate_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                  const int32_t* __restrict__ csum,
                                                                  float* __restrict__ dx, int64_t lddx, int S, int T,
                                                                  int C) {
  const int i = blockIdx.x, b = blockIdx.y;
  const int32_t* cs = csum + (int64_t)b * S;
  int t0 = i ? cs[i - 1] : 0, t1 = cs[i];
  if (t0 > T) t0 = T;
  if (t1 > T) t1 = T;
  float* dxp = dx + ((int64_t)b * S + i) * lddx;
  const float* dyb = dy + (int64_t)b * T * lddy;
  for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = t0; t < t1; ++t) {
      const float4 g = *reinterpret_cast<const float4*>(dyb + (int64_t)t * lddy + c);
      acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
    }
    *reinterpret_cast<float4*>(dxp + c) = acc;
  }
}

extern "C" int styler_length_regulate_bwd(const float* dy, int64_t lddy, const int32_t* csum, float* dx, int64_t lddx,
                                          int B, int S, int T, int C, void* stream) {
  if (!dy || !csum || !dx || B <= 0 || S <= 0 || T <= 0 || C <= 0 || (C & 3)) return STYLER_EINVAL;
  if ((lddy & 3) || (lddx & 3)) return STYLER_EALIGN;
  hipLaunchKernelGGL(length_regulate_bwd_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, dy, lddy, csum, dx, lddx,
                     S, T, C);
  return launch_status();
}

// ---- bucketise/embed/add backward: demb[table][bucket] += sum of the dy rows that looked the bucket up ----------
// A scatter with atomics serialises on the popular buckets (bucket 0 collects every unvoiced frame: ~4000 atomics per
// address, 119 us).  Instead one block OWNS a (table, bucket, row-slice): it scans the slice's ids (L2-resident) and sums
// the matching dy rows in registers -- each row is read once per table, a handful of atomics per block remain.
#define BEB_SLICES 16
__global__ __launch_bounds__(256) void bucket_embed_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ pid,
                                                               const int32_t* __restrict__ eid, float* __restrict__ dpe,
                                                               float* __restrict__ dee, int64_t rows, int nbuckets,
                                                               int pslots) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bucket = blockIdx.x % nbuckets, table = blockIdx.x / nbuckets, slice = blockIdx.y;
  const int32_t* ids = table ? eid : pid;
  const int64_t per = (rows + BEB_SLICES - 1) / BEB_SLICES;
  const int64_t r0 = slice * per, r1 = min(rows, r0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  bool any = false;
  for (int64_t base = r0 + wave * 64; base < r1; base += 256) {          // 64 ids per wave and step, one per lane
    const int64_t r = base + lane;
    const bool hit = r < r1 && ids[r] == bucket;
    uint64_t m = __ballot(hit);
    while (m) {                                                          // the wave walks its matching rows together,
      float4 g[4];                                                       // four row loads in flight
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m) {
          const int k = __ffsll((unsigned long long)m) - 1;
          m &= m - 1;
          g[u] = *reinterpret_cast<const float4*>(dy + (base + k) * 256 + lane * 4);
        }
      }
      acc.x += (g[0].x + g[1].x) + (g[2].x + g[3].x); acc.y += (g[0].y + g[1].y) + (g[2].y + g[3].y);
      acc.z += (g[0].z + g[1].z) + (g[2].z + g[3].z); acc.w += (g[0].w + g[1].w) + (g[2].w + g[3].w);
      any = true;
    }
  }
  red[wave][lane * 4 + 0] = acc.x; red[wave][lane * 4 + 1] = acc.y; red[wave][lane * 4 + 2] = acc.z; red[wave][lane * 4 + 3] = acc.w;
  const int hits = __syncthreads_or(any);
  if (pslots) {
    // STYLER_IO_PARAM_SLOTS: dpe / dee are [BEB_SLICES][nbuckets * 256] slot arrays; every (bucket, slice) block STORES its sum
    // (zeros included), the caller's multi-tensor reduce folds the slices in order: no atomics
    float* dst = (table ? dee : dpe) + ((int64_t)slice * nbuckets + bucket) * 256 + threadIdx.x;
    *dst = hits ? (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) : 0.f;
    return;
  }
  if (!hits) return;
  const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (t != 0.f) atomicAdd((table ? dee : dpe) + (int64_t)bucket * 256 + threadIdx.x, t);
}

extern "C" int styler_bucket_embed_bwd(const float* dy, const int32_t* p_ids, const int32_t* e_ids, float* dpitch_emb,
                                       float* denergy_emb, int B, int T, void* stream) {
  if (!dy || !p_ids || !e_ids || !dpitch_emb || !denergy_emb || B <= 0 || T <= 0) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * T;
  const int nbuckets = 256;                            // hparams.n_bins (modules.py:278-281)
  hipLaunchKernelGGL(bucket_embed_bwd_kernel, dim3(2 * nbuckets, BEB_SLICES), dim3(256), 0, (hipStream_t)stream, dy, p_ids,
                     e_ids, dpitch_emb, denergy_emb, rows, nbuckets, 0);
  return launch_status();
}

// The same with the two gradients as slot arrays [styler_bucket_embed_slices()][256 * 256] (every slot is stored).
extern "C" int styler_bucket_embed_slices(void) { return BEB_SLICES; }
extern "C" int styler_bucket_embed_bwd_slots(const float* dy, const int32_t* p_ids, const int32_t* e_ids, float* dpitch_slots,
                                             float* denergy_slots, int B, int T, void* stream) {
  if (!dy || !p_ids || !e_ids || !dpitch_slots || !denergy_slots || B <= 0 || T <= 0) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * T;
  const int nbuckets = 256;
  hipLaunchKernelGGL(bucket_embed_bwd_kernel, dim3(2 * nbuckets, BEB_SLICES), dim3(256), 0, (hipStream_t)stream, dy, p_ids,
                     e_ids, dpitch_slots, denergy_slots, rows, nbuckets, 1);
  return launch_status();
}

// ---- per-item row sum: out[b,:] (+)= sum_t x[b,t,:]  (gradient of the speaker row broadcast) ----------------
// Block = (64 float4 columns, item): its four waves take rows t = wave, wave + 4, ... eight loads in flight each, and
// meet in LDS (one wave walking all L rows alone was a chain of L dependent round trips on 96 waves chip-wide).
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ out,
                                                     int64_t ldo, int L, int C, int accumulate) {
  __shared__ float4 red[4][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const bool okc = c < C;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* xp = x + (int64_t)b * L * ldx + (okc ? c : 0);
  for (int t0 = wave; t0 < L; t0 += 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(xp + (int64_t)(t0 + 4 * u < L ? t0 + 4 * u : L - 1) * ldx);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (t0 + 4 * u < L) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave != 0 || !okc) return;
  acc.x = (red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x);
  acc.y = (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y);
  acc.z = (red[0][lane].z + red[1][lane].z) + (red[2][lane].z + red[3][lane].z);
  acc.w = (red[0][lane].w + red[1][lane].w) + (red[2][lane].w + red[3][lane].w);
  float4* o = reinterpret_cast<float4*>(out + (int64_t)b * ldo + c);
  if (accumulate) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
  *o = acc;
}

extern "C" int styler_rowsum(const float* x, int64_t ldx, float* out, int64_t ldo, int B, int L, int C, int accumulate,
                             void* stream) {
  if (!x || !out || B <= 0 || L <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldo & 3)) return STYLER_EINVAL;
  hipLaunchKernelGGL(rowsum_kernel, dim3((C / 4 + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, L, C,
                     accumulate);
  return launch_status();
}

// ---- masked error gradient: da = gscale * d/da err(a-b) / count on valid rows, 0 elsewhere -------------------
// Row geometry (C % 4 == 0, C <= 1024): thread = (row-lane, float4 column), four rows per thread in flight, the item length
// of a row from one 32-bit division per row -- the flat element loop it replaces paid two 64-bit divisions and a branch
// around its loads per ELEMENT.  Other shapes take the flat loop.
template <bool VEC>
__device__ __forceinline__ void masked_err_bwd_body(const float* __restrict__ a, int64_t lda,
                                                    const float* __restrict__ b, int64_t ldb,
                                                    const double* __restrict__ acc,
                                                    const float* __restrict__ gscale, float* __restrict__ da,
                                                    int kind, int64_t rows, int L, int C,
                                                    const int64_t* __restrict__ len, const unsigned bx, const unsigned nbx) {
  const float k = gscale[0] / (float)acc[1];
  if constexpr (VEC) {
    const int nq = C >> 2, lanes = 256 / nq;
    const int rl = threadIdx.x / nq, ql = threadIdx.x - rl * nq;
    if (rl >= lanes) return;
    const int64_t stride = (int64_t)nbx * lanes;
    for (int64_t row0 = (int64_t)bx * lanes + rl; row0 < rows; row0 += 4 * stride) {
      float4 x[4], y[4];
      int64_t rc[4], lv[4];
      uint32_t tt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = row0 + u * stride;
        rc[u] = r < rows ? r : rows - 1;
        tt[u] = 0u; lv[u] = 1;
      }
      if (len) {                                         // lengths first, as one batch (see masked_err_mean_kernel)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t bb = (uint32_t)rc[u] / (uint32_t)L;
          tt[u] = (uint32_t)rc[u] - bb * (uint32_t)L;
          lv[u] = len[bb];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = *reinterpret_cast<const float4*>(a + rc[u] * lda + ql * 4);
        y[u] = *reinterpret_cast<const float4*>(b + rc[u] * ldb + ql * 4);
      }
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ok[u] = (int64_t)tt[u] < lv[u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = row0 + u * stride;
        if (r >= rows) break;
        const float d[4] = {x[u].x - y[u].x, x[u].y - y[u].y, x[u].z - y[u].z, x[u].w - y[u].w};
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[e] = kind == 0 ? 2.f * d[e] * k : (d[e] > 0.f ? k : (d[e] < 0.f ? -k : 0.f));
          g[e] = ok[u] ? g[e] : 0.f;
        }
        *reinterpret_cast<float4*>(da + r * C + ql * 4) = make_float4(g[0], g[1], g[2], g[3]);
      }
    }
  } else {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < total; i += (int64_t)nbx * blockDim.x) {
      const int64_t row = i / C; const int c = (int)(i - row * C);
      const int64_t bb = row / L;
      float g = 0.f;
      if (!len || (row - bb * L) < len[bb]) {
        const float d = a[row * lda + c] - b[row * ldb + c];
        g = kind == 0 ? 2.f * d * k : (d > 0.f ? k : (d < 0.f ? -k : 0.f));
      }
      da[i] = g;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void masked_err_bwd_kernel(const float* __restrict__ a, int64_t lda,
                                                             const float* __restrict__ b, int64_t ldb,
                                                             const double* __restrict__ acc,
                                                             const float* __restrict__ gscale, float* __restrict__ da,
                                                             int kind, int64_t rows, int L, int C,
                                                             const int64_t* __restrict__ len) {
  masked_err_bwd_body<VEC>(a, lda, b, ldb, acc, gscale, da, kind, rows, L, C, len, blockIdx.x, gridDim.x);
}

// The backward of up to 8 masked-error terms in one launch (blockIdx.y = term), see styler_masked_err_mean_multi.
struct MaskedBwdTerms {
  const float* a[8]; const float* b[8]; const double* acc[8]; const float* g[8]; float* da[8];
  const int64_t* len[8];
  int64_t lda[8], ldb[8], rows[8];
  int32_t L[8], C[8], kind[8], vec[8], nblk[8];
  int32_t n;
};
__global__ __launch_bounds__(256) void masked_err_bwd_multi_kernel(const MaskedBwdTerms t) {
  const int k = blockIdx.y;
  if (k >= t.n || (int)blockIdx.x >= t.nblk[k]) return;
  if (t.vec[k])
    masked_err_bwd_body<true>(t.a[k], t.lda[k], t.b[k], t.ldb[k], t.acc[k], t.g[k], t.da[k], t.kind[k], t.rows[k], t.L[k], t.C[k],
                              t.len[k], blockIdx.x, (unsigned)t.nblk[k]);
  else
    masked_err_bwd_body<false>(t.a[k], t.lda[k], t.b[k], t.ldb[k], t.acc[k], t.g[k], t.da[k], t.kind[k], t.rows[k], t.L[k], t.C[k],
                               t.len[k], blockIdx.x, (unsigned)t.nblk[k]);
}

extern "C" int styler_masked_err_bwd_multi(const StylerMaskedTerm* terms, int count, void* stream) {
  if (!terms || count <= 0 || count > 8) return STYLER_EINVAL;
  MaskedBwdTerms t;
  int most = 0;
  for (int k = 0; k < count; ++k) {
    const StylerMaskedTerm& m = terms[k];
    if (!m.a || !m.b || !m.acc || !m.gscale || !m.da || m.B <= 0 || m.L <= 0 || m.C <= 0 || (m.kind != 0 && m.kind != 1)) return STYLER_EINVAL;
    const int64_t rows = (int64_t)m.B * m.L;
    const bool vec = !(m.C & 3) && m.C <= 1024 && !(m.lda & 3) && !(m.ldb & 3) && rows < ((int64_t)1 << 31) &&
                     !(((uintptr_t)m.a | (uintptr_t)m.b | (uintptr_t)m.da) & 15);
    t.a[k] = reinterpret_cast<const float*>(m.a); t.b[k] = reinterpret_cast<const float*>(m.b);
    t.acc[k] = reinterpret_cast<const double*>(m.acc); t.g[k] = reinterpret_cast<const float*>(m.gscale);
    t.da[k] = reinterpret_cast<float*>(m.da); t.len[k] = reinterpret_cast<const int64_t*>(m.len);
    t.lda[k] = m.lda; t.ldb[k] = m.ldb; t.rows[k] = rows; t.L[k] = m.L; t.C[k] = m.C; t.kind[k] = m.kind; t.vec[k] = vec ? 1 : 0;
    int64_t blocks;
    if (vec) {
      const int lanes = 256 / (m.C >> 2);
      blocks = (rows + (int64_t)lanes * 4 - 1) / ((int64_t)lanes * 4);
      if (blocks > 2048) blocks = 2048;
    } else {
      blocks = (rows * m.C + 255) / 256;
      if (blocks > 4096) blocks = 4096;
    }
    t.nblk[k] = (int)(blocks < 1 ? 1 : blocks);
    most = t.nblk[k] > most ? t.nblk[k] : most;
  }
  t.n = count;
  hipLaunchKernelGGL(masked_err_bwd_multi_kernel, dim3((unsigned)most, (unsigned)count), dim3(256), 0, (hipStream_t)stream, t);
  return launch_status();
}

extern "C" int styler_masked_err_bwd(const float* a, int64_t lda, const float* b, int64_t ldb, const double* acc,
                                     const float* gscale, float* da, int kind, int B, int L, int C, const int64_t* len,
                                     void* stream) {
  if (!a || !b || !acc || !gscale || !da || B <= 0 || L <= 0 || C <= 0) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  const bool vec = !(C & 3) && C <= 1024 && !(lda & 3) && !(ldb & 3) && rows < ((int64_t)1 << 31) &&
                   !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)da) & 15);
  if (vec) {
    const int lanes = 256 / (C >> 2);
    int64_t blocks = (rows + (int64_t)lanes * 4 - 1) / ((int64_t)lanes * 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(masked_err_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                       acc, gscale, da, kind, rows, L, C, len);
  } else {
    hipLaunchKernelGGL(masked_err_bwd_kernel<false>, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                       ldb, acc, gscale, da, kind, rows, L, C, len);
  }
  return launch_status();
}

// ---- NLL over [B,2] log-probabilities: loss = -mean_b logp[b, label[b]]; fwd (+ optional bwd into dlogp) -----
__global__ void nll_kernel(const float* __restrict__ logp, const int64_t* __restrict__ label, float* __restrict__ loss,
                           const float* __restrict__ gscale, float* __restrict__ dlogp, int B) {
  __shared__ float red[64];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) {
    const int l = (int)label[b];
    s -= logp[b * 2 + l];
    if (dlogp) {
      const float g = -gscale[0] / (float)B;
      dlogp[b * 2 + l] = g; dlogp[b * 2 + 1 - l] = 0.f;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0 && loss) {
    float t = 0.f;
    for (int i = 0; i < 64; ++i) t += red[i];
    loss[0] = t / (float)B;
  }
}

extern "C" int styler_nll(const float* logp, const int64_t* label, float* loss, const float* gscale, float* dlogp, int B,
                          void* stream) {
  if (!logp || !label || (!loss && !dlogp) || (dlogp && !gscale) || B <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(nll_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, logp, label, loss, gscale, dlogp, B);
  return launch_status();
}

// ---- dropout: y = x * keep / (1 - p), keep from a counter-based hash of (seed, element index) -------------
// The same (seed, index) stream regenerates the mask in backward: no mask tensor is stored.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                      int64_t ldy, int64_t rows, int C, float p, uint64_t seed_host,
                                                      const uint64_t* __restrict__ epoch) {
  const uint2 key = dropout_key(mix_drop_epoch(seed_host, epoch));
  const int nq = C / 4;
  const int64_t total = rows * nq;
  const uint32_t thr = dropout_thr16(p);
  const float sc = 1.f / (1.f - p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    float4 v = *reinterpret_cast<const float4*>(x + row * ldx + q * 4);
    v = dropout_apply4(v, key, (uint64_t)(row * C + q * 4), thr, sc);
    *reinterpret_cast<float4*>(y + row * ldy + q * 4) = v;
  }
}

extern "C" int styler_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, float p,
                              uint64_t seed, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3) || p < 0.f || p >= 1.f) return STYLER_EINVAL;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                     rows, C, p, seed, g_styler_drop_epoch);
  return launch_status();
}

const uint64_t* g_styler_drop_epoch = nullptr;

extern "C" int styler_set_dropout_counter(const uint64_t* counter_dev) {
  g_styler_drop_epoch = counter_dev;
  return 0;
}

// ---- optimizer: global grad norm (fp64 sum of squares) + fused clip + Adam on FLAT fp32 buffers ------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  // Round 6: 256 blocks with EIGHT independent 16-byte loads per thread and iteration (was 1024 blocks, one load per iteration).
  // The kernel's time was its atomics: every block ends in one fp64 atomic on the same address, and those are served one after
  // the other -- 30 us for the 118 MB gradient whatever the loads did; a quarter of the blocks is a quarter of the atomics.
  for (; i + 7 * stride + 3 < n; i += 8 * stride) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(g + i + u * stride);
    double t[2] = {0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 8; ++u)
      t[u & 1] += ((double)v[u].x * v[u].x + (double)v[u].y * v[u].y) + ((double)v[u].z * v[u].z + (double)v[u].w * v[u].w);
    s += t[0] + t[1];
  }
  for (; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      for (int64_t k = i; k < n; ++k) s += (double)g[k] * g[k];
    }
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int styler_sumsq(const float* g, int64_t n, double* out, void* stream) {
  if (!g || !out || n <= 0 || ((uintptr_t)g & 15)) return STYLER_EINVAL;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4 + 1, 256, 256)), dim3(256), 0, (hipStream_t)stream, g, n, out);
  return launch_status();
}

// clip_grad_norm_(max_norm) (train.py:181-182: coef = max_norm / (norm + 1e-6), applied only if < 1) fused with
// torch.optim.Adam (betas, eps, no weight decay, bias correction with step count `step`).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const double* __restrict__ sumsq, float max_norm, float lr, float b1,
                                                   float b2, float eps, float bc1, float bc2_sqrt, float grad_scale) {
  // g holds grad_scale^-1 times the gradient (the SUM over ranks of a data-parallel all-reduce, grad_scale = 1 / world):
  // the mean is never materialised, its norm is grad_scale * ||g||
  float coef = grad_scale;
  if (sumsq) {
    const float norm = (float)sqrt(sumsq[0]) * grad_scale;
    const float c = max_norm / (norm + 1e-6f);
    coef = (c < 1.f ? c : 1.f) * grad_scale;
  }
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

extern "C" int styler_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq,
                                float max_norm, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                                void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0 || !(grad_scale > 0.f)) return STYLER_EINVAL;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq,
                     max_norm, lr, beta1, beta2, eps, bc1, bc2s, grad_scale);
  return launch_status();
}
