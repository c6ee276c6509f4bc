// Multi-head self-attention backward (4 heads x 64), exact-fp32 MFMA, recomputing P from the saved
// log-sum-exp (no [4B, L, L] tensor).  Two kernels, no atomics:
//   dq kernel : wave owns 32 queries, walks key tiles  -> dQ, and delta[b,h,q] = <dO, O> for the second kernel
//   dkv kernel: wave owns 32 keys,    walks query tiles -> dK, dV
// MFMA orientations follow attention.hip: the probability / dS tile always sits in the accumulator layout
// whose register r is exactly the B operand of MFMA step r of the next product, so P and dS never move.
//   S = Q K^T / 8, P = exp(S - lse), dP = dO V^T, dS = P * (dP - delta), dQ = dS K / 8, dK = dS^T Q / 8, dV = P^T dO.
#include "common.h"

#define ATT_D 64
#define ATT_LD 68

__device__ __forceinline__ void load_frag32(const float* p, float* f, float scale) {
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const float4 t = *reinterpret_cast<const float4*>(p + v * 4);
    f[v * 4 + 0] = t.x * scale; f[v * 4 + 1] = t.y * scale; f[v * 4 + 2] = t.z * scale; f[v * 4 + 3] = t.w * scale;
  }
}

__device__ __forceinline__ void store_acc_T(float* op, const f32x16& a0, const f32x16& a1, int lh, float scale) {
  // accumulators hold X^T[d][row]: this lane's row, d = 8*g + 4*lh + (0..3) (+32 for a1)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int d = 8 * g + 4 * lh;
    *reinterpret_cast<float4*>(op + d) =
        make_float4(a0[g * 4 + 0] * scale, a0[g * 4 + 1] * scale, a0[g * 4 + 2] * scale, a0[g * 4 + 3] * scale);
    *reinterpret_cast<float4*>(op + 32 + d) =
        make_float4(a1[g * 4 + 0] * scale, a1[g * 4 + 1] * scale, a1[g * 4 + 2] * scale, a1[g * 4 + 3] * scale);
  }
}

// ------------------------------------------------------------------------------------------------- dQ
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(const float* __restrict__ qkv,
                                                               const float* __restrict__ o,
                                                               const float* __restrict__ dout,
                                                               const float* __restrict__ lse,
                                                               float* __restrict__ dqkv, float* __restrict__ delta,
                                                               int B, int L, const int64_t* __restrict__ len, const int* __restrict__ cu) {
  __shared__ __attribute__((aligned(16))) float sK[64 * ATT_LD];
  __shared__ __attribute__((aligned(16))) float sV[64 * ATT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;     // packed rows (pack.hip): items back to back
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;                                      // rows this item owns in memory
  if (Lr <= 0) return;
  const int q = q0 + li, qc = q < Lr ? q : Lr - 1;

  float qf[32], dof[32];
  load_frag32(qkv + (rowbase + qc) * 768 + head * ATT_D + lh * 32, qf, 0.125f);
  load_frag32(dout + (rowbase + qc) * 256 + head * ATT_D + lh * 32, dof, 1.0f);
  float dl = 0.f;
  {
    const float* op = o + (rowbase + qc) * 256 + head * ATT_D + lh * 32;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const float4 t = *reinterpret_cast<const float4*>(op + v * 4);
      dl += t.x * dof[v * 4] + t.y * dof[v * 4 + 1] + t.z * dof[v * 4 + 2] + t.w * dof[v * 4 + 3];
    }
    dl += __shfl_xor(dl, 32, 64);
  }
  const float my_lse = lse[((int64_t)b * 4 + head) * L + qc];
  if (q < Lr && lh == 0) delta[((int64_t)b * 4 + head) * L + q] = dl;

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }

  const int ntiles = (klen + 63) / 64;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * 64;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = tid + p * 256;
      const int kr = idx >> 4, c4 = (idx & 15) * 4;
      const int key = k0 + kr;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < Lr) {
        const float* base = qkv + (rowbase + key) * 768 + head * ATT_D + c4;
        kv = *reinterpret_cast<const float4*>(base + 256);
        vv = *reinterpret_cast<const float4*>(base + 512);
      }
      *reinterpret_cast<float4*>(&sK[kr * ATT_LD + c4]) = kv;
      *reinterpret_cast<float4*>(&sV[kr * ATT_LD + c4]) = vv;
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (k0 + kb * 32 >= klen) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const float* kp = &sK[(kb * 32 + li) * ATT_LD + lh * 32];
      const float* vp = &sV[(kb * 32 + li) * ATT_LD + lh * 32];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(kp + v * 4);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(vp + v * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[v * 4 + e], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], dof[v * 4 + e], dp, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float p = key < klen ? expf(s[r] - my_lse) : 0.f;
        s[r] = p * (dp[r] - dl);                               // dS^T[key][q]
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        dq0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[kr * ATT_LD + li], s[r], dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[kr * ATT_LD + 32 + li], s[r], dq1, 0, 0, 0);
      }
    }
  }
  if (q < Lr) store_acc_T(dqkv + (rowbase + q) * 768 + head * ATT_D, dq0, dq1, lh, 0.125f);
}

// ------------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(const float* __restrict__ qkv,
                                                                const float* __restrict__ dout,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta,
                                                                float* __restrict__ dqkv, int B, int L,
                                                                const int64_t* __restrict__ len, const int* __restrict__ cu) {
  __shared__ __attribute__((aligned(16))) float sQ[64 * ATT_LD];
  __shared__ __attribute__((aligned(16))) float sDO[64 * ATT_LD];
  __shared__ float sLse[64], sDl[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int key0 = blockIdx.x * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;     // packed rows (pack.hip): items back to back
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;                                      // rows this item owns in memory
  if (Lr <= 0) return;
  const int key = key0 + li, keyc = key < Lr ? key : Lr - 1;
  const bool key_ok = key < klen;

  float kf[32], vf[32];
  load_frag32(qkv + (rowbase + keyc) * 768 + 256 + head * ATT_D + lh * 32, kf, 1.0f);
  load_frag32(qkv + (rowbase + keyc) * 768 + 512 + head * ATT_D + lh * 32, vf, 1.0f);

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }

  // whole block skips when all its keys are padding (their dK = dV = 0 is still written below)
  const bool block_live = blockIdx.x * 128 < klen;
  const int ntiles = block_live ? (Lr + 63) / 64 : 0;
  for (int qt = 0; qt < ntiles; ++qt) {
    const int qb = qt * 64;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = tid + p * 256;
      const int qr = idx >> 4, c4 = (idx & 15) * 4;
      const int qq = qb + qr;
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), dv = qv;
      if (qq < Lr) {
        qv = *reinterpret_cast<const float4*>(qkv + (rowbase + qq) * 768 + head * ATT_D + c4);
        dv = *reinterpret_cast<const float4*>(dout + (rowbase + qq) * 256 + head * ATT_D + c4);
      }
      *reinterpret_cast<float4*>(&sQ[qr * ATT_LD + c4]) = qv;
      *reinterpret_cast<float4*>(&sDO[qr * ATT_LD + c4]) = dv;
    }
    if (tid < 64) {
      const int qq = qb + tid;
      sLse[tid] = qq < Lr ? lse[((int64_t)b * 4 + head) * L + qq] : 0.f;
      sDl[tid] = qq < Lr ? delta[((int64_t)b * 4 + head) * L + qq] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int qk = 0; qk < 2; ++qk) {
      if (qb + qk * 32 >= Lr) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const float* qp = &sQ[(qk * 32 + li) * ATT_LD + lh * 32];
      const float* dop = &sDO[(qk * 32 + li) * ATT_LD + lh * 32];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + v * 4);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dop + v * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[e], kf[v * 4 + e], s, 0, 0, 0);       // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[e], vf[v * 4 + e], dp, 0, 0, 0);     // dP[q][key]
        }
      }
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = key_ok && (qb + ql < Lr);
        const float p = ok ? expf(s[r] * 0.125f - sLse[ql]) : 0.f;
        s[r] = p;
        ds[r] = p * (dp[r] - sDl[ql]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sDO[ql * ATT_LD + li], s[r], dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sDO[ql * ATT_LD + 32 + li], s[r], dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[ql * ATT_LD + li], ds[r], dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[ql * ATT_LD + 32 + li], ds[r], dk1, 0, 0, 0);
      }
    }
  }
  if (key < Lr) {
    store_acc_T(dqkv + (rowbase + key) * 768 + 256 + head * ATT_D, dk0, dk1, lh, 0.125f);
    store_acc_T(dqkv + (rowbase + key) * 768 + 512 + head * ATT_D, dv0, dv1, lh, 1.0f);
  }
}

extern "C" int styler_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                                    float* dqkv, float* delta_ws, int B, int L, const int64_t* len,
                                    const int32_t* cu, void* stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || !delta_ws || B <= 0 || L <= 0) return STYLER_EINVAL;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) return STYLER_EALIGN;
  dim3 grid((L + 127) / 128, 4, B);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attention_bwd_dq_kernel, grid, dim3(256), 0, st, qkv, out, dout, lse, dqkv, delta_ws, B, L, len, cu);
  hipLaunchKernelGGL(attention_bwd_dkv_kernel, grid, dim3(256), 0, st, qkv, dout, lse, delta_ws, dqkv, B, L, len, cu);
  return launch_status();
}
