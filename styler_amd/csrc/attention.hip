// Multi-head self-attention forward (4 heads, d_k = 64) on fp32 MFMA, online softmax.
//
// Reference: transformer/Modules.py:14-25 (bmm -> /sqrt(d_k) -> masked_fill(-inf) -> softmax ->
// bmm) as used by SubLayers.py:44-56.  The [4B, L, L] score tensor is never materialised.
//
// Work split: grid (ceil(L/128), head, b); block = 4 waves; each wave owns 32 query rows and
// walks the key axis in tiles of 64 keys staged in LDS (K and V, 16 KiB each).
//
// MFMA orientation (v_mfma_f32_32x32x2_f32, C: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)):
//   S^T[key][q]  = sum_d K[key][d] * Q[q][d]      A = K (LDS, ds_read_b128), B = Q (registers)
//   O^T[d][q]   += sum_key V[key][d] * P^T[key][q] A = V (LDS), B = P^T = the S^T accumulator itself
// so a lane holds ONE query column (q = lane&31) and 16 keys per tile: the row max / row sum
// are in-register plus one lane^32 exchange, and P feeds the second MFMA with no data movement
// (MFMA step r consumes key (r&3)+8*(r>>2)+4*h from half h: exactly accumulator register r).
#include "common.h"

#define ATT_D 64
#define ATT_KT 64          // keys per LDS tile
#define ATT_KLD 68         // K row stride (dwords): 16 consecutive rows -> 16 distinct 16-B slots

__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            float* __restrict__ lse, int B, int L,
                                                            const int64_t* __restrict__ len, const int* __restrict__ cu) {
  __shared__ __attribute__((aligned(16))) float sK[ATT_KT * ATT_KLD];
  __shared__ __attribute__((aligned(16))) float sV[ATT_KT * ATT_D];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;     // packed rows (pack.hip): items back to back
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;                                      // rows this item owns in memory
  if (Lr <= 0) return;

  // Q fragment: lane (q = li, h) holds Q[q][h*32 .. h*32+31], pre-scaled by 1/sqrt(64) (exact)
  float qf[32];
  {
    const int q = q0 + li;
    const float* qp = qkv + (rowbase + (q < Lr ? q : Lr - 1)) * 768 + head * ATT_D + lh * 32;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      float4 t = *reinterpret_cast<const float4*>(qp + v * 4);
      qf[v * 4 + 0] = t.x * 0.125f; qf[v * 4 + 1] = t.y * 0.125f;
      qf[v * 4 + 2] = t.z * 0.125f; qf[v * 4 + 3] = t.w * 0.125f;
    }
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const int ntiles = (klen + ATT_KT - 1) / ATT_KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * ATT_KT;
    __syncthreads();
    // stage K and V tiles: 64 keys x 64 floats each = 1024 float4 -> 4 per thread
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
      *reinterpret_cast<float4*>(&sK[kr * ATT_KLD + c4]) = kv;
      *reinterpret_cast<float4*>(&sV[kr * ATT_D + c4]) = vv;
    }
    __syncthreads();

#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {                  // two 32-key blocks per tile
      if (k0 + kb * 32 >= klen) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const float* kp = &sK[(kb * 32 + li) * ATT_KLD + lh * 32];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        f32x4 kf = *reinterpret_cast<const f32x4*>(kp + v * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[v * 4 + e], s, 0, 0, 0);
      }
      // mask keys >= klen, online softmax per query column
      float mb = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= klen) s[r] = -INFINITY;
        mb = fmaxf(mb, s[r]);
      }
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float m_new = fmaxf(m_run, mb);
      const float alpha = expf(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - m_new); rs += s[r]; }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      // O^T += V^T P^T
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kr = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v0 = sV[kr * ATT_D + li];
        const float v1 = sV[kr * ATT_D + 32 + li];
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o1, 0, 0, 0);
      }
    }
  }

  const int q = q0 + li;
  if (q < Lr) {
    const float inv = 1.f / l_run;
    float* op = out + (rowbase + q) * 256 + head * ATT_D;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 8 * g + 4 * lh;
      *reinterpret_cast<float4*>(op + d) =
          make_float4(o0[g * 4 + 0] * inv, o0[g * 4 + 1] * inv, o0[g * 4 + 2] * inv, o0[g * 4 + 3] * inv);
      *reinterpret_cast<float4*>(op + 32 + d) =
          make_float4(o1[g * 4 + 0] * inv, o1[g * 4 + 1] * inv, o1[g * 4 + 2] * inv, o1[g * 4 + 3] * inv);
    }
    if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = m_run + logf(l_run);
  }
}

extern "C" int styler_attention_fwd(const float* qkv, float* out, float* lse, int B, int L, const int64_t* len,
                                    const int32_t* cu, void* stream) {
  if (!qkv || !out || B <= 0 || L <= 0) return STYLER_EINVAL;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return STYLER_EALIGN;
  dim3 grid((L + 127) / 128, 4, B);
  hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, lse, B, L, len, cu);
  return launch_status();
}
