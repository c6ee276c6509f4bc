// Loss head of the train step (loss.py:16-68, train.py:140-160) without torch glue kernels:
//   masked_err_mean : masked MSE / L1 term with the mean taken IN the kernel (last block), fp64 accumulation
//   nll3            : the three NLLLoss(mean) terms of a classifier triple, summed (loss.py:46-48, 60-68)
//   weighted_sum    : total = sum_i w_i * term_i over scalar device tensors (train.py:156-160), and its backward
//                     (g * w_i for every term in one launch)
// Each of these replaced 2-10 single-element aten kernels (div, cast, add, mul) that cost a launch slot apiece on the
// serial chain of the step.
#include <cstdlib>
#include "common.h"

// Blocks per masked-error term: every block ends in two fp64 atomics + a ticket on ONE address per term, which the L2
// serialises -- past ~100 blocks per term the atomics, not the loads, set the kernel's time (STYLER_LOSS_BLOCKS: the cap).
static inline unsigned loss_grid(int64_t work, int per_block, int cap) {
  static const int env_cap = [] { const char* e = getenv("STYLER_LOSS_BLOCKS"); return e ? atoi(e) : 0; }();
  if (env_cap > 0) cap = env_cap < 1024 ? env_cap : 1024;   // (an accumulator has 1024 block slots)
  int64_t b = (work + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// acc: STYLER_MASKED_ACC_DOUBLES (4 + 2 x 1024) doubles, zero on entry: [0] sum of errors, [1] count of valid elements (both written
// by the block that draws the last ticket), [2] arrival ticket (as uint64), [3] unused, [4 + 2 b], [5 + 2 b] the sums of block b.
// With mean_out == NULL (accumulate-only) just [0], [1] are touched, by atomics.
__device__ __forceinline__ void masked_err_mean_body(const float* __restrict__ a, int64_t lda,
                                                     const float* __restrict__ b, int64_t ldb,
                                                     double* __restrict__ acc, float* __restrict__ mean_out,
                                                     int kind, int64_t rows, int L, int C,
                                                     const int64_t* __restrict__ len, int vec, const unsigned bx,
                                                     const unsigned nbx) {
  __shared__ double red[4][2];
  double s = 0.0, n = 0.0;
  if (vec) {                                             // host: C % 4 == 0, C <= 1024, 16-byte aligned rows, rows < 2^31
    // thread = (row-lane, float4 column); four rows per thread in flight, fetched from clamped addresses and dropped by a
    // select; a row's item comes from one 32-bit division (the flat loop this replaces paid two 64-bit divisions per
    // element and a branch around its loads: thirteen dependent round trips per thread on the mel tensors)
    const int nq = C >> 2, lanes = 256 / nq;
    const int rl = threadIdx.x / nq, ql = threadIdx.x - rl * nq;
    const int64_t stride = (int64_t)nbx * lanes;
    if (rl < lanes) {
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
        if (len) {                                       // the lengths first, as one batch: a load next to its use inside
#pragma unroll                                           // this branch put a wait in front of the row loads
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
        for (int u = 0; u < 4; ++u) ok[u] = (row0 + u * stride < rows) & ((int64_t)tt[u] < lv[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d0 = x[u].x - y[u].x, d1 = x[u].y - y[u].y, d2 = x[u].z - y[u].z, d3 = x[u].w - y[u].w;
          double e;
          if (kind == 0) e = ((double)d0 * d0 + (double)d1 * d1) + ((double)d2 * d2 + (double)d3 * d3);
          else e = ((double)fabsf(d0) + (double)fabsf(d1)) + ((double)fabsf(d2) + (double)fabsf(d3));
          s += ok[u] ? e : 0.0;
          n += ok[u] ? 4.0 : 0.0;
        }
      }
    }
  } else {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < total; i += (int64_t)nbx * blockDim.x) {
      const int64_t row = i / C; const int c = (int)(i - row * C);
      const int64_t bb = row / L;
      if (len && (row - bb * L) >= len[bb]) continue;
      const float d = a[row * lda + c] - b[row * ldb + c];
      s += kind == 0 ? (double)d * d : (double)fabsf(d);
      n += 1.0;
    }
  }
  s = wave_sum_d(s); n = wave_sum_d(n);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s; red[wave][1] = n; }
  __syncthreads();
  __shared__ int last_flag;
  if (threadIdx.x == 0) {
    const double bs = red[0][0] + red[1][0] + red[2][0] + red[3][0], bn = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    last_flag = 0;
    if (!mean_out) {                                     // accumulate-only form (styler_masked_err_sum): device-scope atomics
      atomicAdd(&acc[0], bs);
      atomicAdd(&acc[1], bn);
    } else {
      // Round 6: the block's two sums go to ITS slot (acc[4 + 2 bx], system-scope stores: visible past the per-XCD L2s without an
      // agent-scope release fence, which writes back the whole L2), then one ticket; the block that draws the last ticket adds the
      // slots in a fixed order.  One atomic per block instead of three, and the loss scalars no longer depend on the order the
      // blocks arrive in (they were the last order-dependent values of a training step).
      __hip_atomic_store(&acc[4 + 2 * bx], bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&acc[5 + 2 * bx], bn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t = atomicAdd(reinterpret_cast<unsigned long long*>(&acc[2]), 1ull);
      last_flag = t == (unsigned long long)nbx - 1;
    }
  }
  __syncthreads();
  if (!last_flag) return;
  double ts = 0.0, tn = 0.0;
  for (unsigned i = threadIdx.x; i < nbx; i += 256) {    // (slot i to thread i % 256: a fixed assignment)
    ts += __hip_atomic_load(&acc[4 + 2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    tn += __hip_atomic_load(&acc[5 + 2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  ts = wave_sum_d(ts); tn = wave_sum_d(tn);              // fixed butterfly, then the four waves in order
  __syncthreads();
  if (lane == 0) { red[wave][0] = ts; red[wave][1] = tn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0], cnt = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    acc[0] = tot; acc[1] = cnt;                          // (the backward reads the count)
    mean_out[0] = (float)(tot / cnt);
  }
}

__global__ __launch_bounds__(256) void masked_err_mean_kernel(const float* __restrict__ a, int64_t lda,
                                                              const float* __restrict__ b, int64_t ldb,
                                                              double* __restrict__ acc, float* __restrict__ mean_out,
                                                              int kind, int64_t rows, int L, int C,
                                                              const int64_t* __restrict__ len, int vec) {
  masked_err_mean_body(a, lda, b, ldb, acc, mean_out, kind, rows, L, C, len, vec, blockIdx.x, gridDim.x);
}

// Up to 8 masked-error terms in ONE launch (blockIdx.y = term; the descriptors travel in the kernel arguments): the five
// terms of STYLERLoss.forward (loss.py:26-50) and the two of cal_mel_loss (16-24) were seven launches of 7-15 us each on the
// serial chain between the forward and the backward of a step.
struct MaskedTerms {
  const float* a[8]; const float* b[8];
  double* acc[8]; float* mean[8];
  const int64_t* len[8];
  int64_t lda[8], ldb[8], rows[8];
  int32_t L[8], C[8], kind[8], vec[8], nblk[8];
  int32_t n;
};
__global__ __launch_bounds__(256) void masked_err_mean_multi_kernel(const MaskedTerms t) {
  const int k = blockIdx.y;
  if (k >= t.n || (int)blockIdx.x >= t.nblk[k]) return;
  masked_err_mean_body(t.a[k], t.lda[k], t.b[k], t.ldb[k], t.acc[k], t.mean[k], t.kind[k], t.rows[k], t.L[k], t.C[k], t.len[k],
                       t.vec[k], blockIdx.x, (unsigned)t.nblk[k]);
}

extern "C" int styler_masked_err_mean_multi(const StylerMaskedTerm* terms, int count, void* stream) {
  if (!terms || count <= 0 || count > 8) return STYLER_EINVAL;
  MaskedTerms t;
  int most = 0;
  for (int k = 0; k < count; ++k) {
    const StylerMaskedTerm& m = terms[k];
    if (!m.a || !m.b || !m.acc || !m.mean || m.B <= 0 || m.L <= 0 || m.C <= 0 || (m.kind != 0 && m.kind != 1)) return STYLER_EINVAL;
    const int64_t rows = (int64_t)m.B * m.L;
    const int vec = !(m.C & 3) && m.C <= 1024 && !(m.lda & 3) && !(m.ldb & 3) && rows < ((int64_t)1 << 31) &&
                    !(((uintptr_t)m.a | (uintptr_t)m.b) & 15);
    t.a[k] = reinterpret_cast<const float*>(m.a); t.b[k] = reinterpret_cast<const float*>(m.b);
    t.acc[k] = reinterpret_cast<double*>(m.acc); t.mean[k] = reinterpret_cast<float*>(m.mean);
    t.len[k] = reinterpret_cast<const int64_t*>(m.len);
    t.lda[k] = m.lda; t.ldb[k] = m.ldb; t.rows[k] = rows; t.L[k] = m.L; t.C[k] = m.C; t.kind[k] = m.kind; t.vec[k] = vec;
    t.nblk[k] = (int)loss_grid(rows * m.C / (vec ? 4 : 1), 1024, 256);
    most = t.nblk[k] > most ? t.nblk[k] : most;
  }
  t.n = count;
  hipLaunchKernelGGL(masked_err_mean_multi_kernel, dim3((unsigned)most, (unsigned)count), dim3(256), 0, (hipStream_t)stream, t);
  return launch_status();
}

extern "C" int styler_masked_err_mean(const float* a, int64_t lda, const float* b, int64_t ldb, double* acc, float* mean_out,
                                      int kind, int B, int L, int C, const int64_t* len, void* stream) {
  if (!a || !b || !acc || B <= 0 || L <= 0 || C <= 0 || (kind != 0 && kind != 1)) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  const int vec = !(C & 3) && C <= 1024 && !(lda & 3) && !(ldb & 3) && rows < ((int64_t)1 << 31) &&
                  !(((uintptr_t)a | (uintptr_t)b) & 15);
  hipLaunchKernelGGL(masked_err_mean_kernel, dim3(loss_grid(rows * C / (vec ? 4 : 1), 1024, 256)), dim3(256), 0,
                     (hipStream_t)stream, a, lda, b, ldb, acc, mean_out, kind, rows, L, C, len, vec);
  return launch_status();
}

// ---- three NLL(mean) terms over [B,2] log-probabilities with one label vector, summed -----------------------------------
// forward: loss[0] = sum_k -mean_b logp_k[b, label[b]];  backward (dlogp3 != null): dlogp3[k][b][:] = -g/B at the label.
__global__ void nll3_kernel(const float* __restrict__ lp0, const float* __restrict__ lp1, const float* __restrict__ lp2,
                            const int64_t* __restrict__ label, int label_const, float* __restrict__ loss,
                            const float* __restrict__ gscale, float* __restrict__ dlogp3, int B) {
  const float* lp[3] = {lp0, lp1, lp2};
  float s = 0.f;
  const float g = (dlogp3 && gscale) ? -gscale[0] / (float)B : 0.f;
  for (int i = threadIdx.x; i < 3 * B; i += 64) {
    const int k = i / B, b = i - k * B;
    const int l = label ? (int)label[b] : label_const;
    s -= lp[k][b * 2 + l];
    if (dlogp3) { dlogp3[(k * B + b) * 2 + l] = g; dlogp3[(k * B + b) * 2 + 1 - l] = 0.f; }
  }
  s = wave_sum(s);
  if (threadIdx.x == 0 && loss) loss[0] = s / (float)B;
}

extern "C" int styler_nll3(const float* lp0, const float* lp1, const float* lp2, const int64_t* label, int label_const,
                           float* loss, const float* gscale, float* dlogp3, int B, void* stream) {
  if (!lp0 || !lp1 || !lp2 || (!loss && !dlogp3) || (dlogp3 && !gscale) || B <= 0) return STYLER_EINVAL;
  if (!label && label_const != 0 && label_const != 1) return STYLER_EINVAL;
  hipLaunchKernelGGL(nll3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lp0, lp1, lp2, label, label_const, loss,
                     gscale, dlogp3, B);
  return launch_status();
}

// ---- total = sum_i w_i * term_i over up to 16 scalar device tensors; g_out[i] = g[0] * w_i ------------------------------
struct ScalarTerms { const float* p[16]; float w[16]; int n; };

__global__ void weighted_sum_kernel(ScalarTerms t, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < t.n; ++i) s += t.w[i] * t.p[i][0];       // left to right, like the chained adds it replaces
    out[0] = s;
  }
}
__global__ void scale_weights_kernel(ScalarTerms t, const float* __restrict__ g, float* __restrict__ out) {
  if (threadIdx.x < t.n) out[threadIdx.x] = g[0] * t.w[threadIdx.x];
}

extern "C" int styler_weighted_sum(const float* const* terms, const float* weights, int n, float* out, void* stream) {
  if (!terms || !weights || !out || n <= 0 || n > 16) return STYLER_EINVAL;
  ScalarTerms t;
  t.n = n;
  for (int i = 0; i < 16; ++i) { t.p[i] = i < n ? terms[i] : nullptr; t.w[i] = i < n ? weights[i] : 0.f; }
  for (int i = 0; i < n; ++i) if (!t.p[i]) return STYLER_EINVAL;
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, out);
  return launch_status();
}

extern "C" int styler_scale_weights(const float* g, const float* weights, int n, float* out, void* stream) {
  if (!g || !weights || !out || n <= 0 || n > 16) return STYLER_EINVAL;
  ScalarTerms t;
  t.n = n;
  for (int i = 0; i < 16; ++i) { t.p[i] = nullptr; t.w[i] = i < n ? weights[i] : 0.f; }
  hipLaunchKernelGGL(scale_weights_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, g, out);
  return launch_status();
}

// ---- the tail of the train step's loss head in ONE launch each way (round 6) -----------------------------------------------
// forward : out[1] = NLL3 of the main pass' classifier triple, out[2] = NLL3 of the DAT pass' (loss.py:46-48, 60-68),
//           out[0] = sum_{i<n} w[i] * mean_i + w[n] * out[1] + w[n+1] * out[2]   (train.py:156-160, left to right)
// backward: gw[i] = g[0] * w[i] (i < n: the upstream gradients of the masked-error means) and the six d(logp) blocks
//           d6[k][b][:] = -(g[0] * w[n + k/3]) / B at the label, 0 elsewhere.
// Same per-thread summation order and the same float operations as nll3_kernel / weighted_sum_kernel / scale_weights_kernel:
// the results are bit-identical to the five-launch form (tests/test_14_train_step.py).
struct LossTail {
  const float* mean[8]; float w[10];
  const float* lp[6]; const int64_t* label[2];
  int32_t label_const[2], n, B;
};

__global__ void loss_tail_kernel(const LossTail t, float* __restrict__ out) {
  float cls[2];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
    float s = 0.f;
    for (int i = threadIdx.x; i < 3 * t.B; i += 64) {
      const int k = i / t.B, b = i - k * t.B;
      const int l = t.label[grp] ? (int)t.label[grp][b] : t.label_const[grp];
      s -= t.lp[3 * grp + k][b * 2 + l];
    }
    cls[grp] = wave_sum(s) / (float)t.B;
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < t.n; ++i) s += t.w[i] * t.mean[i][0];
    s += t.w[t.n] * cls[0];
    s += t.w[t.n + 1] * cls[1];
    out[0] = s; out[1] = cls[0]; out[2] = cls[1];
  }
}

__global__ void loss_tail_bwd_kernel(const LossTail t, const float* __restrict__ g, float* __restrict__ gw, float* __restrict__ d6) {
  if ((int)threadIdx.x < t.n) gw[threadIdx.x] = g[0] * t.w[threadIdx.x];
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
    const float gs = g[0] * t.w[t.n + grp];
    const float gg = -gs / (float)t.B;
    for (int i = threadIdx.x; i < 3 * t.B; i += 64) {
      const int k = i / t.B, b = i - k * t.B;
      const int l = t.label[grp] ? (int)t.label[grp][b] : t.label_const[grp];
      float* d = d6 + ((int64_t)(3 * grp + k) * t.B + b) * 2;
      d[l] = gg; d[1 - l] = 0.f;
    }
  }
}

static int loss_tail_fill(LossTail& t, const float* const* means, const float* weights, int n, const float* const* lp6,
                          const int64_t* label0, int const0, const int64_t* label1, int const1, int B) {
  if (!weights || !lp6 || n < 0 || n > 8 || B <= 0 || (n && !means)) return STYLER_EINVAL;
  if ((!label0 && const0 != 0 && const0 != 1) || (!label1 && const1 != 0 && const1 != 1)) return STYLER_EINVAL;
  for (int i = 0; i < 8; ++i) { t.mean[i] = i < n ? means[i] : nullptr; if (i < n && !t.mean[i]) return STYLER_EINVAL; }
  for (int i = 0; i < 10; ++i) t.w[i] = i < n + 2 ? weights[i] : 0.f;
  for (int i = 0; i < 6; ++i) { t.lp[i] = lp6[i]; if (!t.lp[i]) return STYLER_EINVAL; }
  t.label[0] = label0; t.label[1] = label1; t.label_const[0] = const0; t.label_const[1] = const1;
  t.n = n; t.B = B;
  return 0;
}

extern "C" int styler_loss_tail(const float* const* means, const float* weights, int n, const float* const* lp6,
                                const int64_t* label0, int label_const0, const int64_t* label1, int label_const1, int B,
                                float* out3, void* stream) {
  LossTail t;
  if (!out3) return STYLER_EINVAL;
  if (const int rc = loss_tail_fill(t, means, weights, n, lp6, label0, label_const0, label1, label_const1, B)) return rc;
  hipLaunchKernelGGL(loss_tail_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, out3);
  return launch_status();
}

extern "C" int styler_loss_tail_bwd(const float* g, const float* weights, int n, const float* const* lp6,
                                    const int64_t* label0, int label_const0, const int64_t* label1, int label_const1, int B,
                                    float* gw, float* d6, void* stream) {
  LossTail t;
  if (!g || !d6 || (n && !gw)) return STYLER_EINVAL;
  const float* dummy[8] = {g, g, g, g, g, g, g, g};                   // (the means are not read by the backward)
  if (const int rc = loss_tail_fill(t, dummy, weights, n, lp6, label0, label_const0, label1, label_const1, B)) return rc;
  hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, g, gw, d6);
  return launch_status();
}

// Self-test of common.h's wave_sum (half / row swaps + DPP) against the __shfl_xor butterfly it replaces: both sums of every
// 64-value group of `in`, for a bitwise comparison by the caller (tests/test_hip_parity.py).
__global__ void wave_sum_selftest_kernel(const float* __restrict__ in, float* __restrict__ out_swap, float* __restrict__ out_shfl) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const float v = in[i];
  float w = v;
  out_swap[i] = wave_sum(v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
  out_shfl[i] = w;
}

extern "C" int styler_wave_sum_selftest(const float* in, float* out_swap, float* out_shfl, int groups, void* stream) {
  if (!in || !out_swap || !out_shfl || groups <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(wave_sum_selftest_kernel, dim3(groups), dim3(64), 0, (hipStream_t)stream, in, out_swap, out_shfl);
  return launch_status();
}
