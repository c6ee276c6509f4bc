// GEMM / Conv1d-as-implicit-GEMM on MFMA (gfx950).
//
//   y[m, n] = act(scale[n] * sum_{j<kw, c<cin} x[row(m) + j - kw/2, c] * w[n, j*cin + c] + shift[n]) (+ res)
//
// m = b*L + t indexes the padded rectangle [B, L]; rows shifted outside [0, L) of their own
// item read as zero ('same' zero padding), so no im2col buffer is ever built.  The K loop
// walks (tap j, channel chunk); A tiles are gathered straight from the channels-last
// activation with a per-row validity predicate.
//
// Tile engine: 256 threads = 4 waves as 2x2, each wave owns TM x TN MFMA tiles of 32x32,
// so the block tile is (64*TM) x (64*TN).  Two arithmetic modes share the skeleton:
//   * F32 : v_mfma_f32_32x32x2_f32, BK = 32 floats, exact fp32 (== an fmaf chain).
//   * BF16: v_mfma_f32_32x32x16_bf16, BK = 64; activations are converted fp32->bf16 (RNE)
//           while being staged into LDS, weights come from a bf16 shadow copy.
// K is permuted inside a chunk so that each lane's fragment is CONTIGUOUS in LDS (the k <-> lane
// map only has to agree between A and B): lane (i = l&31, h = l>>5) reads 16 floats (F32) or
// 8 bf16 per MFMA step with ds_read_b128.  LDS rows are padded to 36 dwords: for
// ds_read_b128 that makes 16 consecutive rows hit 16 distinct 16-byte slots (conflict-free).
#include "common.h"

struct GemmArgs {
  const float* x; int64_t ldx;
  const void* w;
  const float* scale; const float* shift;
  const float* res; int64_t ldres;
  float* y; int64_t ldy;
  int B, L, cin, n, kw, act;
  const int64_t* len;
};

template <int TM, int TN, bool BF16>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmArgs a) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int BK = BF16 ? 64 : 32;
  constexpr int LDS_LD = 36;                       // dwords per LDS row (32 data + 4 pad)
  constexpr int A_V = BK / 4;                      // float4 per A row (global)
  constexpr int A_RPP = 256 / A_V;                 // rows per pass
  constexpr int A_P = BM / A_RPP;                  // passes
  constexpr int B_V = BF16 ? 8 : 8;                // 16-byte vectors per B row
  constexpr int B_RPP = 256 / B_V;
  constexpr int B_P = BN / B_RPP;

  __shared__ __attribute__((aligned(16))) uint32_t sA[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) uint32_t sB[BN * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int64_t M = (int64_t)a.B * a.L;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int pad = a.kw / 2;
  const int ktot = a.kw * a.cin;
  const int cpt = (a.cin + BK - 1) / BK;           // chunks per tap
  const int nq = a.kw * cpt;

  // ---- per-thread load coordinates ----
  const int a_col = (tid % A_V) * 4;
  int a_t[A_P]; int64_t a_m[A_P];
#pragma unroll
  for (int p = 0; p < A_P; ++p) {
    int64_t m = m0 + tid / A_V + p * A_RPP;
    a_m[p] = m;
    a_t[p] = (m < M) ? (int)(m % a.L) : -1000000;  // invalid rows never pass the range test
  }
  const int b_col = (tid % B_V) * (BF16 ? 8 : 4);  // element offset inside the chunk
  int b_n[B_P];
#pragma unroll
  for (int p = 0; p < B_P; ++p) b_n[p] = n0 + tid / B_V + p * B_RPP;

  float4 ra[A_P];
  uint4 rb[B_P];

  auto load_global = [&](int q) {
    const int j = q / cpt, c0 = (q - j * cpt) * BK;
    const int sh = j - pad;
#pragma unroll
    for (int p = 0; p < A_P; ++p) {
      const int tt = a_t[p] + sh;
      const bool ok = (tt >= 0) && (tt < a.L) && (c0 + a_col < a.cin);
      ra[p] = ok ? *reinterpret_cast<const float4*>(a.x + (a_m[p] + sh) * a.ldx + c0 + a_col)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < B_P; ++p) {
      const bool ok = (b_n[p] < a.n) && (c0 + b_col < a.cin);
      if (BF16) {
        const uint16_t* wp = reinterpret_cast<const uint16_t*>(a.w) + (int64_t)b_n[p] * ktot + j * a.cin + c0 + b_col;
        rb[p] = ok ? *reinterpret_cast<const uint4*>(wp) : make_uint4(0, 0, 0, 0);
      } else {
        const float* wp = reinterpret_cast<const float*>(a.w) + (int64_t)b_n[p] * ktot + j * a.cin + c0 + b_col;
        rb[p] = ok ? *reinterpret_cast<const uint4*>(wp) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int p = 0; p < A_P; ++p) {
      const int r = tid / A_V + p * A_RPP;
      if (BF16) {
        uint2 v = make_uint2(pack_bf16x2(ra[p].x, ra[p].y), pack_bf16x2(ra[p].z, ra[p].w));
        *reinterpret_cast<uint2*>(&sA[r * LDS_LD + a_col / 2]) = v;
      } else {
        *reinterpret_cast<float4*>(&sA[r * LDS_LD + a_col]) = ra[p];
      }
    }
#pragma unroll
    for (int p = 0; p < B_P; ++p) {
      const int r = tid / B_V + p * B_RPP;
      *reinterpret_cast<uint4*>(&sB[r * LDS_LD + (tid % B_V) * 4]) = rb[p];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_global(0);
  store_lds();
  __syncthreads();

  for (int q = 0; q < nq; ++q) {
    if (q + 1 < nq) load_global(q + 1);

    if (BF16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = *reinterpret_cast<const bf16x8*>(&sA[((wm * TM + i) * 32 + li) * LDS_LD + s * 8 + lh * 4]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const bf16x8*>(&sB[((wn * TN + j) * 32 + li) * LDS_LD + s * 8 + lh * 4]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
      f32x4 fa[TM][4], fb[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          fa[i][v] = *reinterpret_cast<const f32x4*>(&sA[((wm * TM + i) * 32 + li) * LDS_LD + lh * 16 + v * 4]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          fb[j][v] = *reinterpret_cast<const f32x4*>(&sB[((wn * TN + j) * 32 + li) * LDS_LD + lh * 16 + v * 4]);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk >> 2][kk & 3], fb[j][kk >> 2][kk & 3],
                                                             acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (q + 1 < nq) {
      store_lds();
      __syncthreads();
    }
  }

  // ---- epilogue: C layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + li;
    if (col >= a.n) continue;
    const float sc = a.scale ? a.scale[col] : 1.f;
    const float sf = a.shift ? a.shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= M) continue;
        float v = apply_act(acc[i][j][r] * sc + sf, a.act);
        if (a.res) v += a.res[row * a.ldres + col];
        if (a.len) {
          const int64_t b = row / a.L;
          if ((row - b * a.L) >= a.len[b]) v = 0.f;
        }
        a.y[row * a.ldy + col] = v;
      }
    }
  }
}

template <int TM, int TN, bool BF16>
static int launch_gemm(const GemmArgs& a, hipStream_t st) {
  const int64_t M = (int64_t)a.B * a.L;
  dim3 grid((unsigned)((M + 64 * TM - 1) / (64 * TM)), (unsigned)((a.n + 64 * TN - 1) / (64 * TN)));
  hipLaunchKernelGGL((conv_gemm_kernel<TM, TN, BF16>), grid, dim3(256), 0, st, a);
  return launch_status();
}

// Tile choice: the 128x128 tile needs >= ~1 block per CU to pay; otherwise 64x64 (4x the blocks).
// Returns bit0 = 128x128 tile (else 64x64), bit1 = bf16 MFMA (else fp32 MFMA).
extern "C" int styler_conv_gemm_variant(int B, int L, int cin, int n, int kw, int prec) {
  (void)cin; (void)kw;
  const int64_t M = (int64_t)B * L;
  const int64_t big_blocks = ((M + 127) / 128) * ((n + 127) / 128);
  const int big = (big_blocks >= 192 && n >= 96) ? 1 : 0;
  return big | (prec == STYLER_PREC_BF16 ? 2 : 0);
}

extern "C" int styler_conv_gemm(const float* x, int64_t ldx, const void* w, const float* scale,
                                const float* shift, const float* res, int64_t ldres, float* y,
                                int64_t ldy, int B, int L, int cin, int n, int kw, int act, int prec,
                                const int64_t* len, void* stream) {
  if (!x || !w || !y || B <= 0 || L <= 0 || cin <= 0 || n <= 0 || kw <= 0 || !(kw & 1)) return STYLER_EINVAL;
  if ((cin & 3) || (ldx & 3) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return STYLER_EALIGN;
  if (prec == STYLER_PREC_BF16 && (cin & 7)) return STYLER_EALIGN;
  GemmArgs a{x, ldx, w, scale, shift, res, ldres, y, ldy, B, L, cin, n, kw, act, len};
  hipStream_t st = (hipStream_t)stream;
  const int64_t M = (int64_t)B * L;
  const bool big = styler_conv_gemm_variant(B, L, cin, n, kw, prec) & 1;
  if (prec == STYLER_PREC_BF16) return big ? launch_gemm<2, 2, true>(a, st) : launch_gemm<1, 1, true>(a, st);
  return big ? launch_gemm<2, 2, false>(a, st) : launch_gemm<1, 1, false>(a, st);
}

// ---------------------------------------------------------------------------------------
__global__ void cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t count) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < count; i += stride) {
    if (i + 3 < count) {
      float4 v = *reinterpret_cast<const float4*>(src + i);
      *reinterpret_cast<uint2*>(dst + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    } else {
      for (int64_t k = i; k < count; ++k) dst[k] = (uint16_t)f32_to_bf16_bits(src[k]);
    }
  }
}

extern "C" int styler_cast_bf16(const float* src, uint16_t* dst, int64_t count, void* stream) {
  if (!src || !dst || count < 0) return STYLER_EINVAL;
  if (count == 0) return 0;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return STYLER_EALIGN;
  int64_t blocks = (count / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, count);
  return launch_status();
}

__global__ void repack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int cin,
                                   int kw, int to_kernel) {
  const int64_t total = (int64_t)n * cin * kw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // i enumerates the DESTINATION
    if (to_kernel) {  // dst [n, kw, cin] <- src [n, cin, kw]
      const int c = i % cin; const int j = (i / cin) % kw; const int64_t o = i / ((int64_t)cin * kw);
      dst[i] = src[(o * cin + c) * kw + j];
    } else {          // dst [n, cin, kw] <- src [n, kw, cin]
      const int j = i % kw; const int c = (i / kw) % cin; const int64_t o = i / ((int64_t)cin * kw);
      dst[i] = src[(o * kw + j) * cin + c];
    }
  }
}

extern "C" int styler_repack_conv_weight(const float* src, float* dst, int n, int cin, int kw,
                                         int to_kernel_layout, void* stream) {
  if (!src || !dst || n <= 0 || cin <= 0 || kw <= 0) return STYLER_EINVAL;
  const int64_t total = (int64_t)n * cin * kw;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(repack_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n,
                     cin, kw, to_kernel_layout);
  return launch_status();
}

extern "C" int styler_abi_version(void) { return 1; }
