// bf16-operand attention (throughput mode): forward + backward on v_mfma_f32_32x32x16_bf16, fp32 softmax / accumulate.
// Same decomposition and MFMA orientations as attention.hip / attention_bwd.hip (a lane owns one query -- or, in the
// dK/dV kernel, one key -- column; P / dS stay in the accumulator registers that feed the next product).
//
// The bf16 MFMA consumes 8 consecutive k per lane.  For the products whose k axis is the head dimension (Q K^T,
// dO V^T) the operands are row-major [row][d] tiles (ds_read_b128) or registers.  For the products whose k axis is
// the key (or query) axis (P V, dS K, P^T dO, dS^T Q) the register operand is the packed P / dS accumulator, whose lane
// holds rows (r&3) + 8*(r>>2) + 4*h of the 32-row block; MFMA step s2 takes registers 8*s2 .. 8*s2+7, i.e. the rows
// {16*s2 + 4h + 0..3} and {16*s2 + 8 + 4h + 0..3}.  The k <-> row map only has to agree between the two operands, so the
// LDS operand is a TRANSPOSED tile [d][row] read as two ds_read_b64 at exactly those two row runs -- no shuffles.
// Transposed tiles are produced while staging by an in-register 8x4 transpose (8 float4 rows -> 4 x ds_write_b128).
#include "common.h"

#define AD 64
#define ALD 36            // dwords per LDS row (64 bf16 + 16 B pad)

__device__ __forceinline__ uint32_t cvtpk(float lo, float hi) {
  return cvt_pk_bf16_rne(lo, hi);
}

// 8 consecutive floats at p (scaled) -> bf16x8
__device__ __forceinline__ bf16x8 load8_bf16(const float* p, float scale) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  uint4 v = make_uint4(cvtpk(a.x * scale, a.y * scale), cvtpk(a.z * scale, a.w * scale), cvtpk(b.x * scale, b.y * scale),
                       cvtpk(b.z * scale, b.w * scale));
  return *reinterpret_cast<bf16x8*>(&v);
}

// accumulator registers 8*s2 .. 8*s2+7 -> bf16x8 (the B operand of MFMA step s2)
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s2) {
  uint4 v = make_uint4(cvtpk(a[s2 * 8 + 0], a[s2 * 8 + 1]), cvtpk(a[s2 * 8 + 2], a[s2 * 8 + 3]),
                       cvtpk(a[s2 * 8 + 4], a[s2 * 8 + 5]), cvtpk(a[s2 * 8 + 6], a[s2 * 8 + 7]));
  return *reinterpret_cast<bf16x8*>(&v);
}

// row-major tile: 64 rows x 64 floats (global row stride ld, column offset folded into src) -> bf16 [64][ALD]
__device__ __forceinline__ void stage_rows(uint32_t* dst, const float* src, int64_t ld, int row0, int nrows_valid, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = tid + p * 256;
    const int r = idx >> 4, c4 = (idx & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < nrows_valid) v = *reinterpret_cast<const float4*>(src + (int64_t)(row0 + r) * ld + c4);
    *reinterpret_cast<uint2*>(&dst[r * ALD + c4 / 2]) = make_uint2(cvtpk(v.x, v.y), cvtpk(v.z, v.w));
  }
}

// transposed tile [64 d][64 rows] from the same source, by 128 threads (t = 0..127): thread (g = t>>4, q4 = (t&15)*4)
// loads rows g*8 .. g*8+7, columns q4 .. q4+3 and writes 4 x 16 B
__device__ __forceinline__ void stage_transposed(uint32_t* dstT, const float* src, int64_t ld, int row0, int nrows_valid, int t) {
  const int g = t >> 4, q4 = (t & 15) * 4;
  float4 v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int r = row0 + g * 8 + e;
    v[e] = (r < nrows_valid) ? *reinterpret_cast<const float4*>(src + (int64_t)r * ld + q4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  *reinterpret_cast<uint4*>(&dstT[(q4 + 0) * ALD + g * 4]) =
      make_uint4(cvtpk(v[0].x, v[1].x), cvtpk(v[2].x, v[3].x), cvtpk(v[4].x, v[5].x), cvtpk(v[6].x, v[7].x));
  *reinterpret_cast<uint4*>(&dstT[(q4 + 1) * ALD + g * 4]) =
      make_uint4(cvtpk(v[0].y, v[1].y), cvtpk(v[2].y, v[3].y), cvtpk(v[4].y, v[5].y), cvtpk(v[6].y, v[7].y));
  *reinterpret_cast<uint4*>(&dstT[(q4 + 2) * ALD + g * 4]) =
      make_uint4(cvtpk(v[0].z, v[1].z), cvtpk(v[2].z, v[3].z), cvtpk(v[4].z, v[5].z), cvtpk(v[6].z, v[7].z));
  *reinterpret_cast<uint4*>(&dstT[(q4 + 3) * ALD + g * 4]) =
      make_uint4(cvtpk(v[0].w, v[1].w), cvtpk(v[2].w, v[3].w), cvtpk(v[4].w, v[5].w), cvtpk(v[6].w, v[7].w));
}

// The same two stagers split into a register load and an LDS store: the next tile's global loads are issued BEFORE the
// current tile is multiplied and land in LDS after it (single LDS buffer, the HBM/L2 latency hides behind the MFMAs).
__device__ __forceinline__ void load_rows(float4 (&v)[4], const float* src, int64_t ld, int row0, int nrows_valid, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = tid + p * 256;
    const int r = idx >> 4, c4 = (idx & 15) * 4;
    v[p] = (row0 + r < nrows_valid) ? *reinterpret_cast<const float4*>(src + (int64_t)(row0 + r) * ld + c4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void store_rows(uint32_t* dst, const float4 (&v)[4], int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int idx = tid + p * 256;
    const int r = idx >> 4, c4 = (idx & 15) * 4;
    *reinterpret_cast<uint2*>(&dst[r * ALD + c4 / 2]) = make_uint2(cvtpk(v[p].x, v[p].y), cvtpk(v[p].z, v[p].w));
  }
}
__device__ __forceinline__ void load_transposed(float4 (&v)[8], const float* src, int64_t ld, int row0, int nrows_valid, int t) {
  const int g = t >> 4, q4 = (t & 15) * 4;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int r = row0 + g * 8 + e;
    v[e] = (r < nrows_valid) ? *reinterpret_cast<const float4*>(src + (int64_t)r * ld + q4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void store_transposed(uint32_t* dstT, const float4 (&v)[8], int t) {
  const int g = t >> 4, q4 = (t & 15) * 4;
  *reinterpret_cast<uint4*>(&dstT[(q4 + 0) * ALD + g * 4]) =
      make_uint4(cvtpk(v[0].x, v[1].x), cvtpk(v[2].x, v[3].x), cvtpk(v[4].x, v[5].x), cvtpk(v[6].x, v[7].x));
  *reinterpret_cast<uint4*>(&dstT[(q4 + 1) * ALD + g * 4]) =
      make_uint4(cvtpk(v[0].y, v[1].y), cvtpk(v[2].y, v[3].y), cvtpk(v[4].y, v[5].y), cvtpk(v[6].y, v[7].y));
  *reinterpret_cast<uint4*>(&dstT[(q4 + 2) * ALD + g * 4]) =
      make_uint4(cvtpk(v[0].z, v[1].z), cvtpk(v[2].z, v[3].z), cvtpk(v[4].z, v[5].z), cvtpk(v[6].z, v[7].z));
  *reinterpret_cast<uint4*>(&dstT[(q4 + 3) * ALD + g * 4]) =
      make_uint4(cvtpk(v[0].w, v[1].w), cvtpk(v[2].w, v[3].w), cvtpk(v[4].w, v[5].w), cvtpk(v[6].w, v[7].w));
}

// A operand from a transposed tile: row d = dt*32 + li, the two 4-row runs of 32-row block `blk`, step s2, half lh
__device__ __forceinline__ bf16x8 fragT(const uint32_t* tT, int dt, int li, int blk, int s2, int lh) {
  const uint32_t* p = &tT[(dt * 32 + li) * ALD + blk * 16 + s2 * 8 + lh * 2];
  const uint2 a = *reinterpret_cast<const uint2*>(p);
  const uint2 b = *reinterpret_cast<const uint2*>(p + 4);
  uint4 v = make_uint4(a.x, a.y, b.x, b.y);
  return *reinterpret_cast<bf16x8*>(&v);
}

__device__ __forceinline__ void store_accT(float* op, const f32x16& a0, const f32x16& a1, int lh, float scale) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int d = 8 * g + 4 * lh;
    *reinterpret_cast<float4*>(op + d) =
        make_float4(a0[g * 4 + 0] * scale, a0[g * 4 + 1] * scale, a0[g * 4 + 2] * scale, a0[g * 4 + 3] * scale);
    *reinterpret_cast<float4*>(op + 32 + d) =
        make_float4(a1[g * 4 + 0] * scale, a1[g * 4 + 1] * scale, a1[g * 4 + 2] * scale, a1[g * 4 + 3] * scale);
  }
}

// ------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void attention_fwd_bf16_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 float* __restrict__ lse, int B, int L,
                                                                 const int64_t* __restrict__ len,
                                                                 const int* __restrict__ cu) {
  __shared__ __attribute__((aligned(16))) uint32_t sK[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sVT[64 * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;     // packed rows (pack.hip): items back to back
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;                                      // rows this item owns in memory
  if (Lr <= 0) return;
  const int q = q0 + li, qc = q < Lr ? q : Lr - 1;
  // Query rows at or past the item's length are don't-care (every caller zeroes them after the following
  // LayerNorm, Layers.py:29): blocks made only of such rows write zeros and leave.
  if (blockIdx.x * 128 >= klen) {
    if (q < Lr) {
      float* op = out + (rowbase + q) * 256 + head * AD + lh * 32;
#pragma unroll
      for (int d = 0; d < 32; d += 4) *reinterpret_cast<float4*>(op + d) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = 0.f;
    }
    return;
  }

  // scores are kept in the log2 domain: q is pre-scaled by log2(e) / sqrt(d_k), so that p = exp2(s - m) is one
  // v_exp_f32 (no range fix-ups: p underflowing to zero is exactly what softmax wants)
  bf16x8 qf[4];
#pragma unroll
  for (int st = 0; st < 4; ++st)
    qf[st] = load8_bf16(qkv + (rowbase + qc) * 768 + head * AD + st * 16 + lh * 8, 0.125f * 1.44269504088896f);

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const float* kbase = qkv + rowbase * 768 + 256 + head * AD;
  const float* vbase = qkv + rowbase * 768 + 512 + head * AD;
  const int ntiles = (klen + 63) / 64;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * 64;
    __syncthreads();
    // (a register prefetch of the next tile was measured here: 37.8 -> 75.8 us -- occupancy 3 -> 2 waves per SIMD and the
    // loads' waits land in front of the MFMAs; the dK/dV kernel, at one wave per SIMD anyway, keeps its prefetch)
    stage_rows(sK, kbase, 768, k0, Lr, tid);
    if (tid < 128) stage_transposed(sVT, vbase, 768, k0, Lr, tid);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (k0 + kb * 32 >= klen) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[(kb * 32 + li) * ALD + st * 8 + lh * 4]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);
      }
      if (k0 + kb * 32 + 32 > klen) {                  // only the block holding the length boundary masks keys
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= klen) s[r] = -INFINITY;
        }
      }
      float mb = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) mb = fmaxf(mb, s[r]);
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float m_new = fmaxf(m_run, mb);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); rs += s[r]; }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack_acc(s, s2);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sVT, 0, li, kb, s2, lh), pb, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sVT, 1, li, kb, s2, lh), pb, o1, 0, 0, 0);
      }
    }
  }
  if (q < Lr) {
    store_accT(out + (rowbase + q) * 256 + head * AD, o0, o1, lh, 1.f / l_run);
    if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = (m_run + log2f(l_run)) * 0.693147180559945f;   // natural log
  }
}

// ------------------------------------------------------------------------------------------------- dQ
__global__ __launch_bounds__(256) void attention_bwd_dq_bf16_kernel(const float* __restrict__ qkv,
                                                                    const float* __restrict__ o,
                                                                    const float* __restrict__ dout,
                                                                    const float* __restrict__ lse,
                                                                    float* __restrict__ dqkv, float* __restrict__ delta,
                                                                    int B, int L, const int64_t* __restrict__ len,
                                                                    const int* __restrict__ cu) {
  __shared__ __attribute__((aligned(16))) uint32_t sK[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sV[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sKT[64 * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;
  if (Lr <= 0) return;
  const int q = q0 + li, qc = q < Lr ? q : Lr - 1;

  // query rows at or past the item's length carry no gradient (they are zeroed after the LayerNorm that follows):
  // blocks made only of such rows write dQ = 0, delta = 0 and leave
  if (blockIdx.x * 128 >= klen) {
    if (q < Lr) {
      float* op = dqkv + (rowbase + q) * 768 + head * AD + lh * 32;
#pragma unroll
      for (int d = 0; d < 32; d += 4) *reinterpret_cast<float4*>(op + d) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lh == 0) delta[((int64_t)b * 4 + head) * L + q] = 0.f;
    }
    return;
  }
  constexpr float LOG2E = 1.44269504088896f;
  bf16x8 qf[4], dof[4];
  float dl = 0.f;
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int off = head * AD + st * 16 + lh * 8;
    qf[st] = load8_bf16(qkv + (rowbase + qc) * 768 + off, 0.125f * LOG2E);     // log2-domain scores, see forward
    const float* dp = dout + (rowbase + qc) * 256 + off;
    const float* op = o + (rowbase + qc) * 256 + off;
    dof[st] = load8_bf16(dp, 1.0f);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += dp[e] * op[e];
  }
  dl += __shfl_xor(dl, 32, 64);
  const float my_lse = lse[((int64_t)b * 4 + head) * L + qc] * LOG2E;
  if (q < Lr && lh == 0) delta[((int64_t)b * 4 + head) * L + q] = dl;

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const float* kbase = qkv + rowbase * 768 + 256 + head * AD;
  const float* vbase = qkv + rowbase * 768 + 512 + head * AD;
  const int ntiles = (klen + 63) / 64;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * 64;
    __syncthreads();
    // (no register prefetch here: +50 VGPRs would halve this kernel's occupancy, 2 -> 1 wave per SIMD)
    stage_rows(sK, kbase, 768, k0, Lr, tid);
    stage_rows(sV, vbase, 768, k0, Lr, tid);
    if (tid < 128) stage_transposed(sKT, kbase, 768, k0, Lr, tid);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (k0 + kb * 32 >= klen) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[(kb * 32 + li) * ALD + st * 8 + lh * 4]);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&sV[(kb * 32 + li) * ALD + st * 8 + lh * 4]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[st], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - my_lse) * (dp[r] - dl);
      if (k0 + kb * 32 + 32 > klen) {                  // only the block holding the length boundary masks keys
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= klen) s[r] = 0.f;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsb = pack_acc(s, s2);
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sKT, 0, li, kb, s2, lh), dsb, dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sKT, 1, li, kb, s2, lh), dsb, dq1, 0, 0, 0);
      }
    }
  }
  if (q < Lr) store_accT(dqkv + (rowbase + q) * 768 + head * AD, dq0, dq1, lh, 0.125f);
}

// ------------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256) void attention_bwd_dkv_bf16_kernel(const float* __restrict__ qkv,
                                                                     const float* __restrict__ dout,
                                                                     const float* __restrict__ lse,
                                                                     const float* __restrict__ delta,
                                                                     float* __restrict__ dqkv, int B, int L,
                                                                     const int64_t* __restrict__ len,
                                                                     const int* __restrict__ cu) {
  __shared__ __attribute__((aligned(16))) uint32_t sQ[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sDO[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sQT[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sDOT[64 * ALD];
  __shared__ float sLse[64], sDl[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int key0 = blockIdx.x * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;
  if (Lr <= 0) return;
  const int key = key0 + li, keyc = key < Lr ? key : Lr - 1;
  const bool key_ok = key < klen;

  constexpr float LOG2E = 1.44269504088896f;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    kf[st] = load8_bf16(qkv + (rowbase + keyc) * 768 + 256 + head * AD + st * 16 + lh * 8, 0.125f * LOG2E);
    vf[st] = load8_bf16(qkv + (rowbase + keyc) * 768 + 512 + head * AD + st * 16 + lh * 8, 1.0f);
  }
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }

  const float* qbase = qkv + rowbase * 768 + head * AD;
  const float* dobase = dout + rowbase * 256 + head * AD;
  // A lane owns one key column, so an invalid key only pollutes its own dK / dV, which are written as zeros below.
  // Query rows at or past klen have dO = 0 and delta = 0 (no gradient reaches them): they add nothing to any dK / dV,
  // so the query loop stops at klen; rows of the last tile past L are staged as zeros (lse = delta = 0 there).
  const bool block_live = blockIdx.x * 128 < klen;
  const int ntiles = block_live ? (klen + 63) / 64 : 0;
  float4 rq[4], rdo[4], rt[8];
  if (ntiles > 0) {
    load_rows(rq, qbase, 768, 0, Lr, tid);
    load_rows(rdo, dobase, 256, 0, Lr, tid);
    if (tid < 128) load_transposed(rt, qbase, 768, 0, Lr, tid);
    else load_transposed(rt, dobase, 256, 0, Lr, tid - 128);
  }
  for (int qt = 0; qt < ntiles; ++qt) {
    const int qb = qt * 64;
    __syncthreads();
    store_rows(sQ, rq, tid);
    store_rows(sDO, rdo, tid);
    if (tid < 128) store_transposed(sQT, rt, tid);
    else store_transposed(sDOT, rt, tid - 128);
    if (qt + 1 < ntiles) {                             // next tile: in flight while this one is multiplied
      load_rows(rq, qbase, 768, qb + 64, Lr, tid);
      load_rows(rdo, dobase, 256, qb + 64, Lr, tid);
      if (tid < 128) load_transposed(rt, qbase, 768, qb + 64, Lr, tid);
      else load_transposed(rt, dobase, 256, qb + 64, Lr, tid - 128);
    }
    if (tid < 64) {
      const int qq = qb + tid;
      // rows at or past klen: lse = +huge makes p exactly 0 (whatever dO / the forward's lse hold there)
      sLse[tid] = qq < klen ? lse[((int64_t)b * 4 + head) * L + qq] * LOG2E : 1e30f;
      sDl[tid] = qq < klen ? delta[((int64_t)b * 4 + head) * L + qq] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int qk = 0; qk < 2; ++qk) {
      if (qb + qk * 32 >= klen) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const bf16x8 qv = *reinterpret_cast<const bf16x8*>(&sQ[(qk * 32 + li) * ALD + st * 8 + lh * 4]);
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(&sDO[(qk * 32 + li) * ALD + st * 8 + lh * 4]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qv, kf[st], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dv, vf[st], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float p = __builtin_amdgcn_exp2f(s[r] - sLse[ql]);
        s[r] = p;
        dp[r] = p * (dp[r] - sDl[ql]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack_acc(s, s2), dsb = pack_acc(dp, s2);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sDOT, 0, li, qk, s2, lh), pb, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sDOT, 1, li, qk, s2, lh), pb, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sQT, 0, li, qk, s2, lh), dsb, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sQT, 1, li, qk, s2, lh), dsb, dk1, 0, 0, 0);
      }
    }
  }
  if (key < Lr) {
    // dK = dS^T (Q / sqrt(d_k)): the staged Q tiles are unscaled, so the 1/8 is applied here; invalid keys get zeros
    // (their accumulators may hold anything, inf included: written as literal zeros, never multiplied by 0)
    if (key_ok) {
      store_accT(dqkv + (rowbase + key) * 768 + 256 + head * AD, dk0, dk1, lh, 0.125f);
      store_accT(dqkv + (rowbase + key) * 768 + 512 + head * AD, dv0, dv1, lh, 1.0f);
    } else {
#pragma unroll
      for (int part = 1; part <= 2; ++part) {
        float* op = dqkv + (rowbase + key) * 768 + part * 256 + head * AD + lh * 32;
#pragma unroll
        for (int d = 0; d < 32; d += 4) *reinterpret_cast<float4*>(op + d) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
}

extern "C" int styler_attention_fwd_bf16(const float* qkv, float* out, float* lse, int B, int L, const int64_t* len,
                                         const int32_t* cu, void* stream) {
  if (!qkv || !out || B <= 0 || L <= 0 || (cu && !len)) return STYLER_EINVAL;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return STYLER_EALIGN;
  hipLaunchKernelGGL(attention_fwd_bf16_kernel, dim3((L + 127) / 128, 4, B), dim3(256), 0, (hipStream_t)stream, qkv, out,
                     lse, B, L, len, cu);
  return launch_status();
}

extern "C" int styler_attention_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse,
                                         float* dqkv, float* delta_ws, int B, int L, const int64_t* len,
                                         const int32_t* cu, void* stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || !delta_ws || B <= 0 || L <= 0) return STYLER_EINVAL;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) return STYLER_EALIGN;
  dim3 grid((L + 127) / 128, 4, B);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attention_bwd_dq_bf16_kernel, grid, dim3(256), 0, st, qkv, out, dout, lse, dqkv, delta_ws, B, L, len, cu);
  hipLaunchKernelGGL(attention_bwd_dkv_bf16_kernel, grid, dim3(256), 0, st, qkv, dout, lse, delta_ws, dqkv, B, L, len, cu);
  return launch_status();
}
