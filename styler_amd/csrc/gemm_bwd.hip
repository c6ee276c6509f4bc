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
#include <cstdlib>
#include <type_traits>
#include "common.h"

__device__ __forceinline__ uint32_t cvt_pk_bf16_b(float lo, float hi) {
  return cvt_pk_bf16_rne(lo, hi);
}

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

// Up to 8 act_bwd problems in one launch (blockIdx.y = problem; descriptors in the kernel arguments): the ReLU backward of
// the members of a grouped Linear node (autograd.ConvGemmMultiFn).
struct ActSegs {
  const float* dy[8]; const float* y[8]; float* dz[8];
  int64_t lddy[8], ldy[8], rows[8];
  int32_t C[8], act[8];
  int32_t n;
};
__global__ __launch_bounds__(256) void act_bwd_multi_kernel(const ActSegs g) {
  const int k = blockIdx.y;
  if (k >= g.n) return;
  const float* __restrict__ dy = g.dy[k];
  const float* __restrict__ y = g.y[k];
  float* __restrict__ dz = g.dz[k];
  const int C = g.C[k], act = g.act[k], nq = C / 4;
  const int64_t total = g.rows[k] * nq, lddy = g.lddy[k], ldy = g.ldy[k];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nq; const int q = (int)(i - row * nq);
    float4 gq = *reinterpret_cast<const float4*>(dy + row * lddy + q * 4);
    const float4 v = *reinterpret_cast<const float4*>(y + row * ldy + q * 4);
    if (act == STYLER_ACT_RELU) {
      gq.x = v.x > 0.f ? gq.x : 0.f; gq.y = v.y > 0.f ? gq.y : 0.f; gq.z = v.z > 0.f ? gq.z : 0.f; gq.w = v.w > 0.f ? gq.w : 0.f;
    } else if (act == STYLER_ACT_TANH) {
      gq.x *= 1.f - v.x * v.x; gq.y *= 1.f - v.y * v.y; gq.z *= 1.f - v.z * v.z; gq.w *= 1.f - v.w * v.w;
    }
    *reinterpret_cast<float4*>(dz + row * (int64_t)C + q * 4) = gq;
  }
}

extern "C" int styler_act_bwd_multi(const StylerActSeg* segs, int count, void* stream) {
  if (!segs || count <= 0 || count > 8) return STYLER_EINVAL;
  ActSegs g;
  int64_t most = 0;
  for (int k = 0; k < count; ++k) {
    const StylerActSeg& s = segs[k];
    if (!s.dy || !s.y || !s.dz || s.rows <= 0 || s.C <= 0 || (s.C & 3) || (s.act != STYLER_ACT_RELU && s.act != STYLER_ACT_TANH)) return STYLER_EINVAL;
    if ((s.lddy & 3) || (s.ldy & 3) || (((uintptr_t)s.dy | (uintptr_t)s.y | (uintptr_t)s.dz) & 15)) return STYLER_EALIGN;
    g.dy[k] = reinterpret_cast<const float*>(s.dy); g.y[k] = reinterpret_cast<const float*>(s.y); g.dz[k] = reinterpret_cast<float*>(s.dz);
    g.lddy[k] = s.lddy; g.ldy[k] = s.ldy; g.rows[k] = s.rows; g.C[k] = s.C; g.act[k] = s.act;
    const int64_t t = s.rows * (s.C / 4);
    most = t > most ? t : most;
  }
  g.n = count;
  int64_t bx = (most + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(act_bwd_multi_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream, g);
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
                                                    float* __restrict__ db, float* __restrict__ db2, int64_t sn,
                                                    int64_t sc, int64_t sj, int B,
                                                    int L, int n, int cin, int pad_bits, int ct, int chunks_per_split,
                                                    float* __restrict__ ws, const int2* __restrict__ rowinfo,
                                                    const int64_t* __restrict__ counts) {
  // pad_bits = (pad_left & 0xff) | (0x800: db is a [splits][n] slot array -- STYLER_IO_DB_SLOTS -- that this launch STORES its
  // per-split column sums into; the caller's multi-tensor reduce folds them in split order: no atomics, fixed order)
  const int pad_left = (int)(int8_t)(pad_bits & 0xff);
  const bool db_slots = pad_bits & 0x800;
  constexpr int XR = WG_BK + KW - 1;                 // x rows per chunk (with halo)
  __shared__ __attribute__((aligned(16))) float sA[2][WG_BK * WG_LD];
  __shared__ __attribute__((aligned(16))) float sB[2][XR * WG_LD];
  __shared__ uint32_t sMask[2][WG_BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x;
  const int n0 = (tile / ct) * 64, c0 = (tile % ct) * 64;
  // packed rows (pack.hip): the row count lives on the device (counts[0]); chunks are spread evenly over the launch's
  // splits (an empty split still writes its zero partial tile); tap validity comes from rowinfo = (t, len - 1 - t)
  const int64_t M = counts ? counts[0] : (int64_t)B * L;
  const int64_t nchunks = (M + WG_BK - 1) / WG_BK;
  if (counts) chunks_per_split = (int)((nchunks + gridDim.y - 1) / gridDim.y);
  const int64_t ch0 = (int64_t)blockIdx.y * chunks_per_split;
  int64_t ch1 = ch0 + chunks_per_split; if (ch1 > nchunks) ch1 = nchunks;
  if (ch0 >= ch1 && !counts) return;

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
        int t, rem;
        if (rowinfo) { const int2 ri = rowinfo[m]; t = ri.x; rem = ri.y; }
        else { t = (int)(m % L); rem = L - 1 - t; }
#pragma unroll
        for (int j = 0; j < KW; ++j) {
          const int o = j - pad_left;
          if (o >= -t && o <= rem) bits |= 1u << j;
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

  if (ch0 < ch1) {
    load(ch0);
    store(0);
  }
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
    // partial tile -> workspace [split][n][KW][cin] (coalesced 128-B rows); styler_wgrad's reduce kernel sums the
    // splits into the parameter-layout gradient (no atomics on the weight gradient)
    float* wp = ws + (int64_t)blockIdx.y * n * KW * cin;
#pragma unroll
    for (int j = 0; j < KW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (nn < n) wp[((int64_t)nn * KW + j) * cin + c] = acc[j][r];
      }
  }
  if (do_bias && n0 + tid < n) {
    if (db_slots) db[(int64_t)blockIdx.y * n + n0 + tid] = bsum;
    else {
      atomicAdd(db + n0 + tid, bsum);
      if (db2) atomicAdd(db2 + n0 + tid, bsum);
    }
  }
}

#define WB_BK 64

// ---------------------------------------------------------------------------------------------
// bf16 wgrad (throughput mode): the same contraction on v_mfma_f32_32x32x16_bf16, transpose-read engine.
// MFMA wants 8 CONSECUTIVE k (= time rows) per lane, but both operands live time-major in HBM.  K chunks (64 rows)
// never straddle utterances (chunks are enumerated per item; halo rows outside the item read as zeros), so no per-tap
// masks exist; a lane slides ONE window of 8 + KW - 1 rows of its x column over the KW taps in registers (even taps:
// register-aligned; odd taps: v_alignbit).
//   * the block tile is (64*TA) n-features x (64*TB) c-features x KW taps, 4 waves as 2x2, each wave TA x TB MFMA
//     tiles per tap (a 128x128 tile halves the operand bytes per MFMA of the Linear shapes);
//   * fragments come from gfx950's LDS transpose read (ds_read_b64_tr_b16): the time-major image is kept as
//     [k/4][f/16] sub-tiles of [4 rows][16 features] bf16 (128 B each); one read hands every lane 4 consecutive time
//     rows of ITS feature, i.e. half an MFMA fragment -- 2 reads per A fragment, 2..4 per x window, instead of
//     8..16 two-byte reads plus the packing VALU.
//   Row r of sub-tile fs sits in 32-byte slot (r + fs) & 3: a staging pass (16 lanes = one time row, four
//   sub-tiles) and a transpose read (32 lanes = 2 sub-tiles x 4 rows = one full 256-B bank row) are both
//   conflict-free.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint2 lds_tr_read(const uint16_t* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(const_cast<uint16_t*>(p)));
  return *reinterpret_cast<const uint2*>(&v);
}

// the low parts of two fp32 values, packed: bf16(v - float(bf16(v))), same rounding as the high part's
__device__ __forceinline__ uint32_t lo_pk(float a, float b) {
  const uint32_t h = cvt_pk_bf16_b(a, b);
  return cvt_pk_bf16_b(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u));
}

// DZ16 / X16: the operand lives in HBM as bf16 already (the FFN hidden activation / its gradient in throughput mode):
// 8-byte loads, no conversion on the staging path.
// `pad_parts` = (pad_left & 0xff) | parts << 8.  parts (the bf16x3 arithmetic on fp32-typed operands, ops.wgrad): bit 0 = stage the LOW
// part of x, v - float(bf16(v)), instead of its rounding to bf16 (the high part); bit 1 = the same for dz.
// Block -> (split, tile) of a bf16 weight-gradient launch with FEWER than 8 splits (round 5).  Workgroup b runs on XCD b % 8
// (observed dispatch rule, speed only), and each XCD has its own L2.  The former map (tile = b % tiles, split = b / tiles) put
// the c-tiles of one n-tile on four different XCDs: the decoder FFN's k = 9 gradient (16 x 4 tiles, 4 splits) fetched every dz
// column block into FOUR L2s and every x column block into two -- 250 MB per launch for 69 MB of operands, L2 hit rate 46 %
// (profiles/r04_pmc_traffic.jsonl).  Now the 8 XCDs form an (s8 x n8 x c8) grid over (splits x n-tiles x c-tiles) and an XCD owns
// the whole sub-box: its blocks walk the same time rows of the same dz / x column blocks together.  Among the exact
// factorisations (s8 | splits, n8 | nt, c8 | ct) the one with the smallest per-XCD operand footprint
// ((nt / n8) FA + (ct / c8) FB) / s8 wins; 83 MB for the shape above.  Without an exact factorisation: the former map.
// pad bit 0x1000 (STYLER_WGRAD_XCDMAP=0) keeps the former map for A/B runs.
__device__ __forceinline__ void wgrad_xcd_box_map(const int bid, const int tiles, const int ct, const int splits, const int FA,
                                                  const int FB, const bool legacy, int& tile, int& split) {
  tile = bid % tiles;
  split = bid / tiles;
  if (legacy) return;
  const int nt = tiles / ct;
  int best = 0x7fffffff, bs = 0, bn = 0, bc = 0;
#pragma unroll
  for (int ls = 0; ls < 4; ++ls)
#pragma unroll
    for (int ln = 0; ln + ls < 4; ++ln) {
      const int s8 = 1 << ls, n8 = 1 << ln, c8 = 8 >> (ls + ln);
      if (splits % s8 || nt % n8 || ct % c8) continue;
      const int cost = ((nt / n8) * FA + (ct / c8) * FB) * (8 >> ls);
      if (cost < best) { best = cost; bs = s8; bn = n8; bc = c8; }
    }
  if (!bs) return;
  const int xcd = bid & 7, l = bid >> 3;
  const int xs = xcd % bs, xn = (xcd / bs) % bn, xc = xcd / (bs * bn);
  const int Nl = nt / bn, Cl = ct / bc, Sl = splits / bs;
  const int cl = l % Cl, nl = (l / Cl) % Nl, sl = l / (Cl * Nl);
  if (sl >= Sl) return;                              // (cannot happen for an exact factorisation; keeps the former map safe)
  tile = (xn * Nl + nl) * ct + xc * Cl + cl;
  split = xs * Sl + sl;
}

template <int KW, int TA, int TB, bool DZ16 = false, bool X16 = false>
__device__ __forceinline__ void wgrad_tr_body(const int bid, const void* __restrict__ dz, int64_t lddz,
                                              const void* __restrict__ x, int64_t ldx, float* __restrict__ db,
                                              float* __restrict__ db2, int B, int L, int n, int cin, int pad_parts, int ct,
                                              int cpi, int chunks_per_split, int tiles, int splits,
                                              float* __restrict__ ws, const int4* __restrict__ chunktab,
                                              const int64_t* __restrict__ counts) {
  const int pad_left = (int)(int8_t)(pad_parts & 0xff);   // (-1: the shifted Linear of the LSTM's reverse direction)
  const bool x_lo = !X16 && (pad_parts & 0x100), dz_lo = !DZ16 && (pad_parts & 0x200);
  const bool db_slots = pad_parts & 0x800;           // STYLER_IO_DB_SLOTS: db = [splits][n] slots, stored (see wgrad_kernel)
  constexpr int FA = 64 * TA, FB = 64 * TB;          // features per block tile
  constexpr int XR = KW == 1 ? 64 : 72;              // x rows per chunk incl. halo (KW - 1 <= 8)
  constexpr int NR = (8 + KW - 1 + 3) / 4;           // transpose reads per x window
  constexpr int SA = FA / 16, SB = FB / 16;          // sub-tiles per 4-row block
  constexpr int VA = FA / 4, VB = FB / 4;            // float4 per staged row
  constexpr int RPA = 256 / VA, RPB = 256 / VB;      // rows per staging pass
  constexpr int PA = 64 / RPA, PB = (XR + RPB - 1) / RPB;
  constexpr int SMEM_MAIN = 2 * (64 * FA + XR * FB) * 2;
  constexpr int SMEM_BIAS = RPA * FA * 4;
  __shared__ __attribute__((aligned(256))) unsigned char smem_raw[SMEM_MAIN > SMEM_BIAS ? SMEM_MAIN : SMEM_BIAS];
  uint16_t* const sA = reinterpret_cast<uint16_t*>(smem_raw);            // 2 x [16][SA][4][16]
  uint16_t* const sB = sA + 2 * 64 * FA;                                 // 2 x [XR/4][SB][4][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  // Block -> (split, tile): workgroup b runs on XCD b % 8 (observed dispatch rule, speed only).  With >= 8 splits an
  // XCD owns a contiguous run of the (split, tile) pairs in split-major order, i.e. whole splits (plus at most two
  // partial ones when the split count is not a multiple of 8): all tiles of a split read the SAME time rows of dz and
  // x, so each operand slice is fetched into that XCD's L2 once and re-read from there by the other tiles (tile-major
  // order scattered them over all eight L2s: 40 % hit rate, 2.4x the algorithmic HBM bytes).  Round 4: the run form
  // replaced "XCD = split % 8", which needed split counts in multiples of 8 to balance the XCDs -- 25 tiles could only
  // be launched as 200 or 400 blocks, never as one full round of the chip.
  int tile, split;
  if (splits >= 8) {
    const int total = tiles * splits, per = (total + 7) >> 3;
    const int i = (bid & 7) * per + (bid >> 3);
    if (i >= total) return;
    split = i / tiles;
    tile = i - split * tiles;
  } else {
    wgrad_xcd_box_map(bid, tiles, ct, splits, FA, FB, pad_parts & 0x1000, tile, split);
  }
  const int n0 = (tile / ct) * FA, c0 = (tile % ct) * FB;
  // Packed rows (pack.hip): the number of rows / chunks lives on the device (`counts`); the chunks are spread evenly
  // over the launch's splits, and a split that gets none still writes its (zero) partial tile for the reduction.
  // kw > 1 enumerates the item-aligned chunks of `chunktab`; kw == 1 (no shifts) is one flat item of counts[0] rows.
  int64_t nchunks = (int64_t)B * cpi;
  if (counts) {
    nchunks = chunktab ? counts[1] : (counts[0] + WB_BK - 1) / WB_BK;
    chunks_per_split = (int)((nchunks + splits - 1) / splits);
    if (!chunktab) L = (int)counts[0];
  }
  const int64_t ch0 = (int64_t)split * chunks_per_split;
  int64_t ch1 = ch0 + chunks_per_split; if (ch1 > nchunks) ch1 = nchunks;
  if (ch0 >= ch1 && !counts) return;

  // ---- staging coordinates ----
  const int aq = (tid % VA) * 4, ar = tid / VA;      // feature / first row of this thread's float4s (dz)
  const int bq = (tid % VB) * 4, br = tid / VB;      // (x)
  const uint32_t sa_off = ((ar >> 2) * SA + (aq >> 4)) * 64 + (((ar & 3) + (aq >> 4)) & 3) * 16 + (aq & 15);
  const uint32_t sb_off = ((br >> 2) * SB + (bq >> 4)) * 64 + (((br & 3) + (bq >> 4)) & 3) * 16 + (bq & 15);
  // Operand fetch = raw buffer loads against per-chunk descriptors (base = the chunk's first row inside the item,
  // num_records = the bytes from there to the item's end): time rows after the item (ragged last chunk, trailing
  // halo) are out of range and read as zeros in hardware; halo rows BEFORE the item get a negative offset, which
  // wraps far above num_records; feature columns past the edge carry the OOB marker.  No masks, no branches, one
  // v_add per load.
  constexpr uint32_t OOB = 0x80000000u;
  constexpr int64_t REC_MAX = (int64_t)1 << 30;      // chunk-relative offsets are < 3 MB; markers and wraps are > 2^30
  const int cinp = (cin + 3) & ~3;
  constexpr int AES = DZ16 ? 2 : 4, BES = X16 ? 2 : 4;       // bytes per operand element in HBM
  const uint32_t va0 = n0 + aq < n ? (uint32_t)((ar * lddz + n0 + aq) * AES) : OOB;
  const uint32_t vb0 = c0 + bq < cinp ? (uint32_t)((br * ldx + c0 + bq) * BES) : OOB;
  const uint32_t a_pstep = (uint32_t)(RPA * lddz * AES), b_pstep = (uint32_t)(RPB * ldx * BES);
  float4 ra0[PA], rb0[PB];                           // (bf16 operands use the first 8 bytes of each)
  // The two buffer descriptors of a chunk: chunk-only quantities (64-bit multiplies, clamps, a division in the padded form).
  // Round 5: the block builds them once for ITS chunks into an LDS table, one chunk per thread, instead of every wave
  // rebuilding them in front of every load (wgrad_dma_body has the measurements; pad bit 0x2000 = the former path).
  const int ich0_t = (int)ch0, nall_t = (int)(ch1 - ch0);
  constexpr int WT_TCAP = 256;
  __shared__ __attribute__((aligned(16))) uint4 s_desc_t[2 * WT_TCAP];
  auto make_desc = [&](int ch, uint4& r0, uint4& r1) {
    int t0, Li;
    int64_t rowb;                                    // first row of the chunk's item, the item's length
    if (chunktab) {
      const int4 e = chunktab[ch];
      t0 = e.y; Li = e.z; rowb = e.x - e.y;
    } else {
      const int b = ch / cpi;
      t0 = (ch - b * cpi) * WB_BK; Li = L; rowb = (int64_t)b * L;
    }
    const int tx = t0 - pad_left;                    // first x row of the chunk (halo included); may be < 0
    const int txb = tx > 0 ? tx : 0;
    int64_t a_rec = ((int64_t)(Li - t0 - 1) * lddz + n) * AES;
    int64_t b_rec = ((int64_t)(Li - txb - 1) * ldx + cinp) * BES;
    a_rec = a_rec > REC_MAX ? REC_MAX : a_rec;
    b_rec = b_rec > REC_MAX ? REC_MAX : (b_rec < 0 ? 0 : b_rec);
    const uint64_t a_base = (uint64_t)(reinterpret_cast<const char*>(dz) + (rowb + t0) * lddz * AES);
    const uint64_t b_base = (uint64_t)(reinterpret_cast<const char*>(x) + (rowb + txb) * ldx * BES);
    r0 = make_uint4((uint32_t)a_base, (uint32_t)(a_base >> 32), (uint32_t)b_base, (uint32_t)(b_base >> 32));
    r1 = make_uint4((uint32_t)a_rec, (uint32_t)b_rec, (uint32_t)((tx - txb) * (int)ldx * BES), 0u);   // .z <= 0: rows before the item wrap out of range
  };
  const bool use_tab_t = !(pad_parts & 0x2000) && nall_t > 0 && nall_t <= WT_TCAP;
  if (use_tab_t) {
    for (int k = tid; k < nall_t; k += 256) make_desc(ich0_t + k, s_desc_t[2 * k], s_desc_t[2 * k + 1]);
    __syncthreads();
  }
  auto load = [&](float4 (&ra)[PA], float4 (&rb)[PB], int ch) {
    uint4 r0, r1;
    if (use_tab_t) { r0 = s_desc_t[2 * (ch - ich0_t)]; r1 = s_desc_t[2 * (ch - ich0_t) + 1]; }
    else make_desc(ch, r0, r1);
    const uint32_t a_lo = __builtin_amdgcn_readfirstlane(r0.x), a_hi = __builtin_amdgcn_readfirstlane(r0.y);
    const uint32_t b_lo = __builtin_amdgcn_readfirstlane(r0.z), b_hi = __builtin_amdgcn_readfirstlane(r0.w);
    const char* a_base = reinterpret_cast<const char*>(((uint64_t)a_hi << 32) | a_lo);
    const char* b_base = reinterpret_cast<const char*>(((uint64_t)b_hi << 32) | b_lo);
    const __amdgpu_buffer_rsrc_t ra_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a_base), 0, (int)__builtin_amdgcn_readfirstlane(r1.x), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b_base), 0, (int)__builtin_amdgcn_readfirstlane(r1.y), 0x00020000);
    const uint32_t b_off = __builtin_amdgcn_readfirstlane(r1.z);
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      if (DZ16) {
        const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(ra_rsrc, va0 + p * a_pstep, 0, 0);
        ra[p].x = __int_as_float(v.x); ra[p].y = __int_as_float(v.y);
      } else {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra_rsrc, va0 + p * a_pstep, 0, 0);
        ra[p] = *reinterpret_cast<const float4*>(&v);
      }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      if (p * RPB >= XR) continue;
      if (X16) {
        const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rb_rsrc, vb0 + (b_off + p * b_pstep), 0, 0);
        rb[p].x = __int_as_float(v.x); rb[p].y = __int_as_float(v.y);
      } else {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb_rsrc, vb0 + (b_off + p * b_pstep), 0, 0);
        rb[p] = *reinterpret_cast<const float4*>(&v);
      }
    }
  };
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  auto store = [&](const float4 (&ra)[PA], const float4 (&rb)[PB], int buf) {
    uint16_t* da = sA + buf * 64 * FA + sa_off;
    uint16_t* dbp = sB + buf * XR * FB + sb_off;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      if (DZ16) {                                    // 4 bf16 in .x/.y: stored as they are; widened only for the bias sum
        const uint32_t lo = __float_as_uint(ra[p].x), hi = __float_as_uint(ra[p].y);
        bs.x += __uint_as_float(lo << 16); bs.y += __uint_as_float(lo & 0xffff0000u);
        bs.z += __uint_as_float(hi << 16); bs.w += __uint_as_float(hi & 0xffff0000u);
        *reinterpret_cast<uint2*>(da + p * (RPA / 4) * SA * 64) = make_uint2(lo, hi);
      } else {
        bs.x += ra[p].x; bs.y += ra[p].y; bs.z += ra[p].z; bs.w += ra[p].w;
        *reinterpret_cast<uint2*>(da + p * (RPA / 4) * SA * 64) =
            dz_lo ? make_uint2(lo_pk(ra[p].x, ra[p].y), lo_pk(ra[p].z, ra[p].w))
                  : make_uint2(cvt_pk_bf16_b(ra[p].x, ra[p].y), cvt_pk_bf16_b(ra[p].z, ra[p].w));
      }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p)
      if (br + p * RPB < XR)
        *reinterpret_cast<uint2*>(dbp + p * (RPB / 4) * SB * 64) =
            X16 ? make_uint2(__float_as_uint(rb[p].x), __float_as_uint(rb[p].y))
                : (x_lo ? make_uint2(lo_pk(rb[p].x, rb[p].y), lo_pk(rb[p].z, rb[p].w))
                        : make_uint2(cvt_pk_bf16_b(rb[p].x, rb[p].y), cvt_pk_bf16_b(rb[p].z, rb[p].w)));
  };

  // ---- fragment read coordinates: 16-lane group (ch = feature half, lh = k half), q = lane in the group ----
  const int q = lane & 15, chf = (lane >> 4) & 1;
  uint32_t fa_off[TA], fb_off[TB];
#pragma unroll
  for (int i = 0; i < TA; ++i) {
    const int fs = 2 * (wm * TA + i) + chf;
    fa_off[i] = (lh * 2 * SA + fs) * 64 + (((q >> 2) + fs) & 3) * 16 + (q & 3) * 4;
  }
#pragma unroll
  for (int i = 0; i < TB; ++i) {
    const int fs = 2 * (wn * TB + i) + chf;
    fb_off[i] = (lh * 2 * SB + fs) * 64 + (((q >> 2) + fs) & 3) * 16 + (q & 3) * 4;
  }

  f32x16 acc[TA][TB][KW];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int jt = 0; jt < TB; ++jt)
#pragma unroll
      for (int j = 0; j < KW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jt][j][r] = 0.f;

  auto compute = [&](int buf) {
    const uint16_t* pa = sA + buf * 64 * FA;
    const uint16_t* pb = sB + buf * XR * FB;
#pragma unroll
    for (int s = 0; s < WB_BK / 16; ++s) {
      bf16x8 fa[TA];
#pragma unroll
      for (int i = 0; i < TA; ++i) {
        const uint2 lo = lds_tr_read(pa + fa_off[i] + (s * 4 + 0) * SA * 64);
        const uint2 hi = lds_tr_read(pa + fa_off[i] + (s * 4 + 1) * SA * 64);
        const uint4 a4 = make_uint4(lo.x, lo.y, hi.x, hi.y);
        fa[i] = *reinterpret_cast<const bf16x8*>(&a4);
      }
#pragma unroll
      for (int jt = 0; jt < TB; ++jt) {
        uint32_t win[2 * NR + 1];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const uint2 v = lds_tr_read(pb + fb_off[jt] + (s * 4 + r) * SB * 64);
          win[2 * r] = v.x; win[2 * r + 1] = v.y;
        }
        win[2 * NR] = 0u;
#pragma unroll
        for (int j = 0; j < KW; ++j) {
          uint4 b4;
          const int o = j / 2;
          if ((j & 1) == 0) {
            b4 = make_uint4(win[o], win[o + 1], win[o + 2], win[o + 3]);
          } else {
            b4 = make_uint4(__builtin_amdgcn_alignbit(win[o + 1], win[o], 16), __builtin_amdgcn_alignbit(win[o + 2], win[o + 1], 16),
                            __builtin_amdgcn_alignbit(win[o + 3], win[o + 2], 16), __builtin_amdgcn_alignbit(win[o + 4], win[o + 3], 16));
          }
          const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&b4);
#pragma unroll
          for (int i = 0; i < TA; ++i) acc[i][jt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc[i][jt][j], 0, 0, 0);
        }
      }
    }
  };

  const int ich0 = (int)ch0, ich1 = (int)ch1;
  if (ich0 < ich1) {
    load(ra0, rb0, ich0);
    store(ra0, rb0, 0);
  }
  __syncthreads();
  int buf = 0;
  for (int ch = ich0; ch < ich1; ++ch) {
    const bool more = ch + 1 < ich1;
    if (more) load(ra0, rb0, ch + 1);
    compute(buf);
    if (more) store(ra0, rb0, buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial tile -> workspace [split][n][KW][cin]; C layout: col (= c) = lane&31, row (= n) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* wp = ws + (int64_t)split * n * KW * cin;
#pragma unroll
  for (int jt = 0; jt < TB; ++jt) {
    const int c = c0 + (wn * TB + jt) * 32 + li;
    if (c >= cin) continue;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < KW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nn = n0 + (wm * TA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (nn < n) wp[((int64_t)nn * KW + j) * cin + c] = acc[i][jt][j][r];
        }
  }
  if (db && (tile % ct) == 0) {                      // bias gradient from the fp32 staging registers
    float* sBias = reinterpret_cast<float*>(smem_raw);                   // [RPA][FA]; the loop's last barrier is behind us
    *reinterpret_cast<float4*>(&sBias[ar * FA + aq]) = bs;
    __syncthreads();
    if (tid < FA && n0 + tid < n) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < RPA; ++r) t += sBias[r * FA + tid];
      if (db_slots) db[(int64_t)split * n + n0 + tid] = t;
      else {
        atomicAdd(db + n0 + tid, t);
        if (db2) atomicAdd(db2 + n0 + tid, t);
      }
    }
  }
}

template <int KW, int TA, int TB, bool DZ16 = false, bool X16 = false>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const void* __restrict__ dz, int64_t lddz,
                                                       const void* __restrict__ x, int64_t ldx,
                                                       float* __restrict__ db, float* __restrict__ db2, int B, int L,
                                                       int n, int cin, int pad_left, int ct, int cpi,
                                                       int chunks_per_split, int tiles, int splits,
                                                       float* __restrict__ ws, const int4* __restrict__ chunktab,
                                                       const int64_t* __restrict__ counts) {
  wgrad_tr_body<KW, TA, TB, DZ16, X16>(blockIdx.x, dz, lddz, x, ldx, db, db2, B, L, n, cin, pad_left, ct, cpi,
                                       chunks_per_split, tiles, splits, ws, chunktab, counts);
}

// ---------------------------------------------------------------------------------------------
// Round 4: the same engine with its operands fetched by LDS-DMA (buffer_load_dwordx4 ... lds) into a ring of NST stages
// -- for launches whose BOTH operands already live in HBM as bf16 (the decoder's FFN / attention / projection
// gradients on the bf16 residual stream, the k = 5 gradients of the AudioEncoder / PostNet stacks: 70 % of the step's
// weight-gradient time).  What changes against wgrad_tr_body, and only that:
//   * no staging registers, no ds_write pass, no conversion: a 16-byte piece (8 features of one time row) goes from L2
//     straight to its slot of the [k/4][f/16][4 rows][16 features] image.  The DMA writes lane l of a piece at
//     base + 16 l, so the IMAGE order is produced on the SOURCE side: lane l of piece p fetches sub-tile 8p + (l >> 3),
//     32-byte slot (l & 7) >> 1 = (row + sub-tile) & 3, feature half l & 1 -- the slot rotation that keeps the transpose
//     reads conflict-free is a per-lane source address, computed once;
//   * the chunk that is computed was requested NST - 1 iterations earlier: `s_waitcnt vmcnt(pieces still allowed in
//     flight)` + ONE raw s_barrier per chunk, the refill of the stage read in the previous iteration is issued right
//     behind that barrier (every wave has left its reads of it), vmcnt never drains inside the loop;
//   * rows outside the item (ragged last chunk, halo rows before / after the item) and feature columns past the edge
//     are out of range of the per-chunk buffer descriptor: the DMA writes zeros (as the register loads read zeros);
//   * the bias gradient has no staging registers to come from: one extra MFMA per A fragment against a vector of ones
//     (c-tile 0, wave column 0 only) accumulates colsum(dz) in fp32 -- products with 1.0 are exact.
// compute() is wgrad_tr_body's, on the same LDS image: the partial tiles are BIT-IDENTICAL to the register-staged kernel's
// for the same split plan (tests/test_91_bf16_acts.py keeps both forms and compares with torch.equal).
typedef __attribute__((address_space(3))) void wg_lds_void;
template <int N> __device__ __forceinline__ void wg_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// STYLER_WGRAD_TRACE builds only (tools/wgrad_trace.sh: a second library next to the product's): per-wave cycle sums of the
// phases of the ring loop -- [DMA wait | barrier | issue | MFMA half 1 | barrier | MFMA half 2] -- for the first 16 blocks.
#ifdef STYLER_WGRAD_TRACE
__device__ uint64_t* g_wgrad_trace = nullptr;
extern "C" int styler_wgrad_trace_ptr(void* p) {
  uint64_t* v = reinterpret_cast<uint64_t*>(p);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_trace), &v, sizeof(v));
}
#define WGT_DECL uint64_t wgt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t wgt_t = __builtin_amdgcn_s_memtime(); const uint64_t wgt_t0 = wgt_t;
#define WGT(k) { __builtin_amdgcn_s_waitcnt(0xc07f); const uint64_t t_ = __builtin_amdgcn_s_memtime(); wgt[k] += t_ - wgt_t; wgt_t = t_; }
#else
#define WGT_DECL
#define WGT(k)
#endif

// (TAG only makes the specialisations of the two kernels that share a shape distinct: hipcc's host pass fails the SECOND
// kernel's call of one and the same body specialisation with "substitution failure")
template <int KW, int TA, int TB, int NST, int KG, int TAG = 0>
__device__ __forceinline__ void wgrad_dma_body(const int bid, const void* __restrict__ dz, int64_t lddz,
                                               const void* __restrict__ x, int64_t ldx, float* __restrict__ db,
                                               float* __restrict__ db2, int B, int L, int n, int cin, int pad_cat, int ct,
                                               int cpi, int chunks_per_split, int tiles, int splits,
                                               float* __restrict__ ws, const int4* __restrict__ chunktab,
                                               const int64_t* __restrict__ counts) {
  // pad_cat = (pad_left & 0xff) | (x3cat ? 0x400 : 0).  x3cat (STYLER_IO_X3CAT, the bf16x3 arithmetic): dz and x are
  // [hi | lo (| hi)] split tensors (n resp. cin columns per part) and the launch computes dz_hi^T x_hi + dz_hi^T x_lo + dz_lo^T x_hi
  // as ONE contraction over three times the chunks: chunk ch of part p = ch / (chunks per part) reads column block
  // (0, 0, 1)[p] of dz and column block (0, 1, 0)[p] of x.  One set of partial tiles instead of three; the bias row sums skip part 1 (dz_hi twice).
  const int pad_left = (int)(int8_t)(pad_cat & 0xff);
  const bool x3cat = pad_cat & 0x400;
  const bool db_slots = pad_cat & 0x800;             // STYLER_IO_DB_SLOTS: db = [splits][n] slots, stored (see wgrad_kernel)
  constexpr int FA = 64 * TA, FB = 64 * TB;
  constexpr int XR = KW == 1 ? 64 : 72;
  constexpr int NR = (8 + KW - 1 + 3) / 4;
  constexpr int SA = FA / 16, SB = FB / 16;
  constexpr int A_BYTES = 64 * FA * 2, B_BYTES = XR * FB * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = A_BYTES / 4096;                 // 1 KB pieces per wave: dz image
  constexpr int PB = (64 * FB * 2) / 4096;           //                       x image, rows 0..63
  constexpr bool HALO = KW > 1;                      // rows 64..71 of the x image: one more piece (FB == 64), wave 3's
  static_assert(!HALO || FB == 64, "the halo piece assumes a 64-feature x tile");
  constexpr int D = NST - 1;                         // prefetch distance in chunks
  // KG = 2: the block is TWO groups of four waves that split the block's K range (group g takes the chunks ch0 + g, + 2,
  // ...) with a ring each, and add their accumulators through LDS at the end: the same two-blocks-per-CU occupancy as two
  // independent blocks, but ONE partial tile instead of two leaves the CU (the split-K partials of a training step and the
  // pass that folds them: 1.39 GB -> 0.7 GB).
  constexpr int RING = NST * STAGE;
  constexpr int NACC = TA * TB * KW * 16 + TA * 16;    // accumulator registers per lane (tiles + bias)
  constexpr int XCH = KG == 2 ? ((NACC + 1) / 2) * 1024 : 0;            // LDS of one half of the exchange
  constexpr int SMEM = KG * RING > XCH ? KG * RING : XCH;
  __shared__ __attribute__((aligned(1024))) unsigned char smem_all[SMEM];        // the ONLY LDS object of the kernel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KG == 2 ? wave8 >> 2 : 0, wave = wave8 & 3;
  unsigned char* const smem = smem_all + grp * RING;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  int tile, split;
  if (splits >= 8) {                                 // an XCD owns a contiguous run of (split, tile) pairs (see wgrad_tr_body)
    const int total = tiles * splits, per = (total + 7) >> 3;
    const int i = (bid & 7) * per + (bid >> 3);
    if (i >= total) return;
    split = i / tiles;
    tile = i - split * tiles;
  } else {
    wgrad_xcd_box_map(bid, tiles, ct, splits, FA, FB, pad_cat & 0x1000, tile, split);
  }
  const int n0 = (tile / ct) * FA, c0 = (tile % ct) * FB;
  int64_t nchunks = (int64_t)B * cpi;
  if (counts) nchunks = chunktab ? counts[1] : (counts[0] + WB_BK - 1) / WB_BK;
  const int nch1 = (int)nchunks;                     // chunks per part
  if (x3cat) nchunks *= 3;
  if (counts) {
    chunks_per_split = (int)((nchunks + splits - 1) / splits);
    if (!chunktab) L = (int)counts[0];
  }
  const int64_t ch0 = (int64_t)split * chunks_per_split;
  int64_t ch1 = ch0 + chunks_per_split; if (ch1 > nchunks) ch1 = nchunks;
  if (ch0 >= ch1 && !counts) return;
  auto part_of = [&](int ch) -> int { return x3cat ? (ch >= 2 * nch1 ? 2 : (ch >= nch1 ? 1 : 0)) : 0; };

  // ---- DMA source offsets (bytes, relative to the chunk's descriptor base), one per piece of this wave ----
  constexpr uint32_t OOB = 0x80000000u;
  constexpr int64_t REC_MAX = (int64_t)1 << 30;
  auto piece_off = [&](int p, int S, int64_t ld, int f0, int flim) -> uint32_t {
    const int st = 8 * p + (lane >> 3);              // sub-tile of the image
    const int k4 = st / S, fs = st - k4 * S;
    const int w = lane & 7, slot = w >> 1, half = w & 1;
    const int r = (slot - fs) & 3;                   // row r of sub-tile fs sits in 32-byte slot (r + fs) & 3
    const int row = 4 * k4 + r, feat = f0 + 16 * fs + 8 * half;
    return feat + 8 <= flim ? (uint32_t)((row * ld + feat) * 2) : OOB;
  };
  uint32_t va[PA], vb[PB], vh = OOB;
#pragma unroll
  for (int q = 0; q < PA; ++q) va[q] = piece_off(wave + 4 * q, SA, lddz, n0, n);
#pragma unroll
  for (int q = 0; q < PB; ++q) vb[q] = piece_off(wave + 4 * q, SB, ldx, c0, cin);
  if (HALO) vh = piece_off(4 * PB, SB, ldx, c0, cin);
  const uint32_t lds_w = (uint32_t)wave * 1024u;

  // chunk table entry of chunk `ch` (packed / item-aligned chunks)
  auto entry = [&](int ch) -> int4 {
    return (chunktab && ch < (int)ch1) ? chunktab[ch - part_of(ch) * nch1] : make_int4(0, 0, 0, 0);
  };
  // The two buffer descriptors of a chunk (bases, record counts, halo offset) depend on the chunk only -- not on the tile,
  // the wave or the stage.  Until round 5 every wave rebuilt them in front of every DMA issue: 64-bit multiplies, clamps, an
  // integer division in the padded-rectangle form.  tools/wgrad_trace.py (phase cycle sums, trace build) put that arithmetic at
  // 12-18 % of a wave's life in the k = 9 / k = 5 kernels and 17-30 % in the Linear tile -- twice what the DMA instructions
  // themselves cost.  Now the block computes the descriptors of ITS chunks once, one chunk per thread, into an LDS table
  // (32 bytes per chunk); an issue reads one entry (fetched an iteration ahead) and moves it to scalar registers.
  // pad bit 0x2000 (STYLER_WGRAD_DESCTAB=0) or more than WG_TCAP chunks in the block: the former on-the-fly path.
  struct WgDesc { uint32_t a_lo, a_hi, b_lo, b_hi, a_rec, b_rec, b_off, pad; };
  auto make_desc = [&](int chk, const int4 e) -> WgDesc {
    const int part = part_of(chk), ch = chk - part * nch1;
    int t0, Li;
    int64_t rowb;
    if (chunktab) {
      t0 = e.y; Li = e.z; rowb = e.x - e.y;
    } else {
      const int b = ch / cpi;
      t0 = (ch - b * cpi) * WB_BK; Li = L; rowb = (int64_t)b * L;
    }
    const int tx = t0 - pad_left;
    const int txb = tx > 0 ? tx : 0;
    int64_t a_rec = ((int64_t)(Li - t0 - 1) * lddz + n) * 2;
    int64_t b_rec = ((int64_t)(Li - txb - 1) * ldx + cin) * 2;
    a_rec = a_rec > REC_MAX ? REC_MAX : a_rec;
    b_rec = b_rec > REC_MAX ? REC_MAX : (b_rec < 0 ? 0 : b_rec);
    // (x3cat: splits are [hi | lo (| hi)] -- column block (0, 0, 1)[part] of dz, column block (0, 1, 0)[part] of x)
    const uint64_t a_base = (uint64_t)(reinterpret_cast<const char*>(dz) + ((rowb + t0) * lddz + (int64_t)(part >> 1) * n) * 2);
    const uint64_t b_base = (uint64_t)(reinterpret_cast<const char*>(x) + ((rowb + txb) * ldx + (int64_t)(part & 1) * cin) * 2);
    WgDesc d;
    d.a_lo = (uint32_t)a_base; d.a_hi = (uint32_t)(a_base >> 32); d.b_lo = (uint32_t)b_base; d.b_hi = (uint32_t)(b_base >> 32);
    d.a_rec = (uint32_t)a_rec; d.b_rec = (uint32_t)b_rec;
    d.b_off = (uint32_t)((tx - txb) * (int)ldx * 2);   // <= 0: rows before the item wrap out of range
    d.pad = 0u;
    return d;
  };
  constexpr int WG_TCAP = 256;
  __shared__ __attribute__((aligned(16))) uint4 s_desc[2 * WG_TCAP];
  const int nall_blk = (int)(ch1 - ch0);
  const bool use_tab = !(pad_cat & 0x2000) && nall_blk <= WG_TCAP;
  if (use_tab) {
    for (int k = tid; k < nall_blk; k += 256 * KG) {
      const int chk = (int)ch0 + k;
      const WgDesc d = make_desc(chk, entry(chk));
      s_desc[2 * k] = make_uint4(d.a_lo, d.a_hi, d.b_lo, d.b_hi);
      s_desc[2 * k + 1] = make_uint4(d.a_rec, d.b_rec, d.b_off, 0u);
    }
    __syncthreads();
  }
  // (raw 8 dwords of a chunk's table entry: read an iteration ahead, the LDS latency stays off the issue path)
  auto tab_read = [&](int chk, uint4& r0, uint4& r1) {
    const int k = chk - (int)ch0;
    if (use_tab) {
      if (k >= 0 && k < nall_blk) { r0 = s_desc[2 * k]; r1 = s_desc[2 * k + 1]; }
    } else {                                         // (no table: the chunk-table entry itself travels an iteration ahead, as in round 4)
      const int4 e = entry(chk);
      r0 = make_uint4((uint32_t)e.x, (uint32_t)e.y, (uint32_t)e.z, (uint32_t)e.w);
    }
  };
  WGT_DECL
  // The refill of a stage is PA + PB (+ 1 halo) DMA pieces per wave.  `ilv` (pad bit 0x4000, STYLER_WGRAD_ILV): they are not issued
  // in one burst behind the barrier but between the K steps of the chunk being computed (`piece`, called from compute):
  // an LDS-DMA instruction holds the wave's issue for 60-180 cycles, which a burst puts in front of the first MFMA.
  constexpr int NPIECE = PA + PB + (HALO ? 1 : 0), NSTEP = WB_BK / 16, PPS = (NPIECE + NSTEP - 1) / NSTEP;
  const bool ilv = (pad_cat & 0x4000) || (KW == 5 && TA == 2);      // measured: +3.5 % on the tall k = 5 tile, neutral elsewhere
  __amdgpu_buffer_rsrc_t pend_ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dz), 0, 0, 0x00020000), pend_rb = pend_ra;
  uint32_t pend_boff = 0u;
  int pend_stage = 0;
  bool pend_on = false;
  auto piece = [&](int q) {                          // DMA piece q of the pending refill
    if (q < PA) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(pend_ra, (wg_lds_void*)(smem + pend_stage * STAGE + lds_w + q * 4096), 16, va[q], 0, 0, 0);
    } else if (q < PA + PB) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(pend_rb, (wg_lds_void*)(smem + pend_stage * STAGE + A_BYTES + lds_w + (q - PA) * 4096), 16,
                                               vb[q - PA] + pend_boff, 0, 0, 0);
    } else if (HALO && wave == 3) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(pend_rb, (wg_lds_void*)(smem + pend_stage * STAGE + A_BYTES + 4 * PB * 1024), 16,
                                               vh + pend_boff, 0, 0, 0);
    }
  };
  auto after_step = [&](int s) {                     // called by compute behind K step s of the current chunk
    if (!pend_on) return;
#pragma unroll
    for (int q = 0; q < PPS; ++q)
      if (s * PPS + q < NPIECE) piece(s * PPS + q);
    if (s == NSTEP - 1) pend_on = false;
  };
  auto launch = [&](int stage, const WgDesc d, const bool defer) {
    const char* a_base = reinterpret_cast<const char*>(((uint64_t)d.a_hi << 32) | d.a_lo);
    const char* b_base = reinterpret_cast<const char*>(((uint64_t)d.b_hi << 32) | d.b_lo);
    pend_ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a_base), 0, (int)d.a_rec, 0x00020000);
    pend_rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b_base), 0, (int)d.b_rec, 0x00020000);
    pend_boff = d.b_off;
    pend_stage = stage;
    WGT(6)                                           // (trace builds: descriptor arithmetic | the DMA instructions)
    if (defer) { pend_on = true; return; }
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) piece(q);
  };
  // issue the DMA of chunk `chk` into `stage`: from the table entry (r0, r1) read earlier, or (no table) built on the spot
  auto issue = [&](int chk, int stage, const uint4 r0, const uint4 r1, const bool defer = false) {
    if (use_tab) {
      WgDesc d;
      d.a_lo = __builtin_amdgcn_readfirstlane(r0.x); d.a_hi = __builtin_amdgcn_readfirstlane(r0.y);
      d.b_lo = __builtin_amdgcn_readfirstlane(r0.z); d.b_hi = __builtin_amdgcn_readfirstlane(r0.w);
      d.a_rec = __builtin_amdgcn_readfirstlane(r1.x); d.b_rec = __builtin_amdgcn_readfirstlane(r1.y);
      d.b_off = __builtin_amdgcn_readfirstlane(r1.z); d.pad = 0u;
      launch(stage, d, defer);
    } else {
      launch(stage, make_desc(chk, make_int4((int)__builtin_amdgcn_readfirstlane(r0.x), (int)__builtin_amdgcn_readfirstlane(r0.y),
                                             (int)__builtin_amdgcn_readfirstlane(r0.z), (int)__builtin_amdgcn_readfirstlane(r0.w))), defer);
    }
  };

  // ---- fragment read coordinates (wgrad_tr_body's) ----
  const int q16 = lane & 15, chf = (lane >> 4) & 1;
  uint32_t fa_off[TA], fb_off[TB];
#pragma unroll
  for (int i = 0; i < TA; ++i) {
    const int fs = 2 * (wm * TA + i) + chf;
    fa_off[i] = (lh * 2 * SA + fs) * 64 + (((q16 >> 2) + fs) & 3) * 16 + (q16 & 3) * 4;
  }
#pragma unroll
  for (int i = 0; i < TB; ++i) {
    const int fs = 2 * (wn * TB + i) + chf;
    fb_off[i] = (lh * 2 * SB + fs) * 64 + (((q16 >> 2) + fs) & 3) * 16 + (q16 & 3) * 4;
  }

  f32x16 acc[TA][TB][KW];
  f32x16 accb[TA];
#pragma unroll
  for (int i = 0; i < TA; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
    for (int jt = 0; jt < TB; ++jt)
#pragma unroll
      for (int j = 0; j < KW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jt][j][r] = 0.f;
  }
  const bool do_bias = db && (tile % ct) == 0 && wn == 0;
  const uint4 ones4 = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
  const bf16x8 ones = *reinterpret_cast<const bf16x8*>(&ones4);

  auto compute = [&](int stage, const bool bias_now, auto s_lo_tag, auto s_hi_tag) {
    constexpr int S_LO = decltype(s_lo_tag)::value, S_HI = decltype(s_hi_tag)::value;
    const uint16_t* pa = reinterpret_cast<const uint16_t*>(smem + stage * STAGE);
    const uint16_t* pb = reinterpret_cast<const uint16_t*>(smem + stage * STAGE + A_BYTES);
#pragma unroll
    for (int s = S_LO; s < S_HI; ++s) {
      bf16x8 fa[TA];
#pragma unroll
      for (int i = 0; i < TA; ++i) {
        const uint2 lo = lds_tr_read(pa + fa_off[i] + (s * 4 + 0) * SA * 64);
        const uint2 hi = lds_tr_read(pa + fa_off[i] + (s * 4 + 1) * SA * 64);
        const uint4 a4 = make_uint4(lo.x, lo.y, hi.x, hi.y);
        fa[i] = *reinterpret_cast<const bf16x8*>(&a4);
      }
#pragma unroll
      for (int jt = 0; jt < TB; ++jt) {
        uint32_t win[2 * NR + 1];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const uint2 v = lds_tr_read(pb + fb_off[jt] + (s * 4 + r) * SB * 64);
          win[2 * r] = v.x; win[2 * r + 1] = v.y;
        }
        win[2 * NR] = 0u;
#pragma unroll
        for (int j = 0; j < KW; ++j) {
          uint4 b4;
          const int o = j / 2;
          if ((j & 1) == 0) {
            b4 = make_uint4(win[o], win[o + 1], win[o + 2], win[o + 3]);
          } else {
            b4 = make_uint4(__builtin_amdgcn_alignbit(win[o + 1], win[o], 16), __builtin_amdgcn_alignbit(win[o + 2], win[o + 1], 16),
                            __builtin_amdgcn_alignbit(win[o + 3], win[o + 2], 16), __builtin_amdgcn_alignbit(win[o + 4], win[o + 3], 16));
          }
          const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&b4);
#pragma unroll
          for (int i = 0; i < TA; ++i) acc[i][jt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb, acc[i][jt][j], 0, 0, 0);
        }
      }
      if (bias_now) {
#pragma unroll
        for (int i = 0; i < TA; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], ones, accb[i], 0, 0, 0);
      }
      after_step(s);                                 // (ilv: this step's share of the pending refill's DMA pieces)
    }
  };

  // ---- the ring ----  (group g owns the block's chunks ch0 + g + KG * k; the trip count is group 0's, a group that has
  // run out of chunks only keeps meeting the barriers)
  const int ich0 = (int)ch0 + grp, nall = (int)(ch1 - ch0);
  const int nch = nall > grp ? (nall - grp + KG - 1) / KG : 0, trips = (nall + KG - 1) / KG;
  constexpr int PW = PA + PB;                        // pieces per chunk of waves 0..2 (wave 3: + the halo piece)
  uint4 nx0 = make_uint4(0u, 0u, 0u, 0u), nx1 = nx0;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nch) {
      tab_read(ich0 + KG * d, nx0, nx1);
      issue(ich0 + KG * d, d, nx0, nx1);
    }
  tab_read(ich0 + KG * D, nx0, nx1);
  int st_c = 0, st_i = D % NST;                      // stage computed / stage refilled in the current iteration
  // KG = 2: the two groups run ONE BARRIER APART (group 1 meets one extra barrier before the loop, group 0 one behind it),
  // and an iteration has two phases -- [wait, barrier, refill the ring, first half of the chunk's MFMAs] and [barrier, second
  // half]: while one group waits for its DMA and issues the next, the other group of the same SIMDs is inside its pure-MFMA
  // phase (in lockstep both groups left the matrix pipe idle during that part of every iteration).
  using std::integral_constant;
  constexpr int NS = WB_BK / 16;
  // (only where a half chunk is still a long MFMA run -- k = 9: 18 MFMAs per wave and phase; measured: k = 9 123 -> 117 us,
  //  k = 5 (10 per phase) 122 -> 127 us, the Linear tile unchanged)
#ifndef STYLER_WGRAD_STAG_TALL
#define STYLER_WGRAD_STAG_TALL 1
#endif
  constexpr bool STAG = KG == 2 && (KW == 9 || (STYLER_WGRAD_STAG_TALL && KW == 5 && TA == 2));    // (tall k = 5: 20 MFMAs per wave and phase)
  if (STAG && grp == 1) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int i = 0; i < trips; ++i) {
    const bool live = i < nch;
    WGT(7)
    if (live) {
      const int rem = nch - 1 - i < D - 1 ? nch - 1 - i : D - 1;        // chunks requested after chunk i
      if (HALO && wave == 3) {
        if (rem >= 2) wg_vm_wait<2 * (PW + 1)>(); else if (rem == 1) wg_vm_wait<PW + 1>(); else wg_vm_wait<0>();
      } else {
        if (rem >= 2) wg_vm_wait<2 * PW>(); else if (rem == 1) wg_vm_wait<PW>(); else wg_vm_wait<0>();
      }
    }
    WGT(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    WGT(1)
    const bool bias_now = do_bias && part_of(ich0 + KG * i) != 1;
    if (live) {
      if (i + D < nch) {                             // the stage read in iteration i - 1: every wave is past those reads
        issue(ich0 + KG * (i + D), st_i, nx0, nx1, ilv);
        tab_read(ich0 + KG * (i + D + 1), nx0, nx1);
      }
      WGT(2)
      if (STAG) compute(st_c, bias_now, integral_constant<int, 0>{}, integral_constant<int, NS / 2>{});
      else compute(st_c, bias_now, integral_constant<int, 0>{}, integral_constant<int, NS>{});
    }
    WGT(3)
    if (STAG) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      WGT(4)
      if (live) compute(st_c, bias_now, integral_constant<int, NS / 2>{}, integral_constant<int, NS>{});
      WGT(5)
    }
    st_c = st_c + 1 == NST ? 0 : st_c + 1;
    st_i = st_i + 1 == NST ? 0 : st_i + 1;
  }
#ifdef STYLER_WGRAD_TRACE
  if (g_wgrad_trace && bid < 16 && lane == 0) {
    uint64_t* t = g_wgrad_trace + ((int64_t)bid * 8 + wave8) * 10;
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = wgt[k];
    t[8] = (uint64_t)trips;
    t[9] = __builtin_amdgcn_s_memtime() - wgt_t0;
  }
#endif
  if (STAG && grp == 0) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (KG == 2) {
    // group 1 hands its accumulators to group 0 through LDS, half of the registers at a time ([register][lane of the
    // group]: consecutive lanes, conflict-free); group 0 adds in a fixed order -- own + other -- and carries on alone
    float* const xch = reinterpret_cast<float*>(smem_all);
    const int t256 = tid & 255;
    constexpr int HALF = (NACC + 1) / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();                               // the rings (h = 0) / the first half (h = 1) are no longer read
      if (grp == 1) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < TA; ++i) {
#pragma unroll
          for (int jt = 0; jt < TB; ++jt)
#pragma unroll
            for (int j = 0; j < KW; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r, ++k)
                if (k >= h * HALF && k < (h + 1) * HALF) xch[(k - h * HALF) * 256 + t256] = acc[i][jt][j][r];
#pragma unroll
          for (int r = 0; r < 16; ++r, ++k)
            if (k >= h * HALF && k < (h + 1) * HALF) xch[(k - h * HALF) * 256 + t256] = accb[i][r];
        }
      }
      __syncthreads();
      if (grp == 0) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < TA; ++i) {
#pragma unroll
          for (int jt = 0; jt < TB; ++jt)
#pragma unroll
            for (int j = 0; j < KW; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r, ++k)
                if (k >= h * HALF && k < (h + 1) * HALF) acc[i][jt][j][r] += xch[(k - h * HALF) * 256 + t256];
#pragma unroll
          for (int r = 0; r < 16; ++r, ++k)
            if (k >= h * HALF && k < (h + 1) * HALF) accb[i][r] += xch[(k - h * HALF) * 256 + t256];
        }
      }
    }
    if (grp == 1) return;
  }
  // partial tile -> workspace [split][n][KW][cin] (as wgrad_tr_body)
  float* wp = ws + (int64_t)split * n * KW * cin;
#pragma unroll
  for (int jt = 0; jt < TB; ++jt) {
    const int c = c0 + (wn * TB + jt) * 32 + li;
    if (c >= cin) continue;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < KW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int nn = n0 + (wm * TA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (nn < n) wp[((int64_t)nn * KW + j) * cin + c] = acc[i][jt][j][r];
        }
  }
  if (do_bias && li == 0) {                          // every column of accb holds the row sums: lanes 0 / 32 own 16 rows each
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = n0 + (wm * TA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (nn < n) {
          if (db_slots) db[(int64_t)split * n + nn] = accb[i][r];
          else {
            atomicAdd(db + nn, accb[i][r]);
            if (db2) atomicAdd(db2 + nn, accb[i][r]);
          }
        }
      }
  }
}

template <int KW, int TA, int TB, int NST, int KG>
__global__ __launch_bounds__(256 * KG) void wgrad_dma_kernel(const void* __restrict__ dz, int64_t lddz,
                                                        const void* __restrict__ x, int64_t ldx,
                                                        float* __restrict__ db, float* __restrict__ db2, int B, int L,
                                                        int n, int cin, int pad_left, int ct, int cpi,
                                                        int chunks_per_split, int tiles, int splits,
                                                        float* __restrict__ ws, const int4* __restrict__ chunktab,
                                                        const int64_t* __restrict__ counts) {
  wgrad_dma_body<KW, TA, TB, NST, KG>(blockIdx.x, dz, lddz, x, ldx, db, db2, B, L, n, cin, pad_left, ct, cpi,
                                      chunks_per_split, tiles, splits, ws, chunktab, counts);
}

// Many weight gradients in ONE launch.  Launched one by one, a weight gradient is alone on the chip and needs >= 2 blocks
// per CU of its own: 8..37 split-K partial tiles per output tile (2 GB of partials per training step).  As members of
// one launch per kernel variant the gradients of a whole backward pass fill the chip together, so a member needs only
// 1..3 splits (runtime: WgradArena picks the count from the group's size in the previous step).  Descriptor i owns
// blocks [block_start[i], block_start[i] + nblocks[i]); gaps (alignment to the 8 XCDs) belong to no member.
template <int KW, int TA, int TB, bool DZ16, bool X16>
__global__ __launch_bounds__(256) void wgrad_tr_group_kernel(const StylerWgradGroupDesc* __restrict__ desc, int count) {
  int lo = 0, hi = count - 1;                        // last descriptor with block_start <= blockIdx.x
  const int bid = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].block_start <= bid) lo = mid; else hi = mid - 1; }
  const StylerWgradGroupDesc d = desc[lo];
  if (bid - d.block_start >= d.nblocks) return;
  wgrad_tr_body<KW, TA, TB, DZ16, X16>(bid - d.block_start, reinterpret_cast<const void*>(d.dz), d.lddz,
                                       reinterpret_cast<const void*>(d.x), d.ldx, reinterpret_cast<float*>(d.db),
                                       reinterpret_cast<float*>(d.db2), d.B, d.L, d.n, d.cin, d.pad_left, d.ct, d.cpi, d.cps,
                                       d.tiles, d.splits, reinterpret_cast<float*>(d.ws),
                                       reinterpret_cast<const int4*>(d.chunktab), reinterpret_cast<const int64_t*>(d.counts));
}

// The grouped launch on the LDS-DMA ring (both operands of every member bf16-resident: the decoder's attention projections
// of all four layers, sixteen 256 x 256 gradients over 27 k rows, deferred to the flush of the step).
__global__ __launch_bounds__(256) void wgrad_dma_group_lin128_kernel(const StylerWgradGroupDesc* __restrict__ desc, int count) {
  int lo = 0, hi = count - 1;
  const int bid = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].block_start <= bid) lo = mid; else hi = mid - 1; }
  const StylerWgradGroupDesc d = desc[lo];
  if (bid - d.block_start >= d.nblocks) return;
  const void* pdz = reinterpret_cast<const void*>(d.dz);
  const void* px = reinterpret_cast<const void*>(d.x);
  float* pdb = reinterpret_cast<float*>(d.db);
  float* pdb2 = reinterpret_cast<float*>(d.db2);
  float* pws = reinterpret_cast<float*>(d.ws);
  const int4* pct = reinterpret_cast<const int4*>(d.chunktab);
  const int64_t* pcn = reinterpret_cast<const int64_t*>(d.counts);
  const int lb = bid - d.block_start;
  const int64_t lddz = d.lddz, ldx = d.ldx;
  const int B = d.B, L = d.L, n = d.n, cin = d.cin, pad_left = d.pad_left & 0xcff, ct = d.ct, cpi = d.cpi, cps = d.cps, tiles = d.tiles,
            splits = d.splits;
  wgrad_dma_body<1, 2, 2, 2, 1, 1>(lb, pdz, lddz, px, ldx, pdb, pdb2, B, L, n, cin, pad_left, ct, cpi, cps, tiles, splits, pws,
                                     pct, pcn);
}

// dw[nn*sn + c*sc + j*sj] += sum over splits of ws[split][nn][j][c]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int64_t sn,
                                                           int64_t sc, int64_t sj, int n, int cin, int kw, int splits) {
  const int64_t per = (int64_t)n * kw * cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * per + i];
    const int c = (int)(i % cin); const int j = (int)((i / cin) % kw); const int64_t nn = i / ((int64_t)cin * kw);
    dw[nn * sn + c * sc + j * sj] += s;
  }
}

// Block tile of the bf16 engine, in 64-feature units (TA over n, TB over cin).  STYLER_WGRAD_TILE=11 / 22 force the
// smallest / largest tile (experiments).
static void wgrad_tile(int n, int cin, int kw, int prec, int* TA, int* TB) {
  *TA = 1; *TB = 1;
  if (prec != STYLER_PREC_BF16) return;
  static const int tile_env = [] { const char* e = getenv("STYLER_WGRAD_TILE"); return e ? atoi(e) : 0; }();
  if (tile_env == 11) return;
  if (tile_env == 22) {                              // experiments: the largest tile the tap count allows
    if (kw <= 5 && n > 64) *TA = 2;
    if (kw == 1 && cin > 64) *TB = 2;
    return;
  }
  // Linear gradients: the 128x128 tile from 16 64x64 tiles (256 x 256) up.  A 64x64 tile loads 32 KB of operands per 64-row
  // chunk for 4 MFMAs per wave -- the grouped launch of the small gradients ran at the CU's vector-memory rate, not at the
  // MFMA rate; the 128x128 tile halves the bytes per FLOP.  Gradients below 48 tiles still run as members of the grouped
  // launch, at 8 split-K partials (ops.wgrad): train step 11.515 -> 11.446 ms same-box.  With taps the operand reuse
  // already is KW-fold and the larger tile only costs occupancy.  STYLER_WGRAD_LIN128_TILES overrides the bound.
  static const int lin128 = [] { const char* e = getenv("STYLER_WGRAD_LIN128_TILES"); return e ? atoi(e) : 16; }();
  if (kw == 1 && ((n + 63) / 64) * ((cin + 63) / 64) >= lin128 && n > 64 && cin > 64) { *TA = 2; *TB = 2; }
}

// The LDS-DMA ring for bf16-resident operands (wgrad_dma_kernel).  Mode 2 (default): 512-thread blocks of two K groups --
// half the split-K partial tiles; mode 1: 256-thread blocks, the register-staged kernel's split plan and its partial
// tiles bit for bit (the A/B switch of the parity tests); mode 0 / STYLER_WGRAD_DMA=0: the register-staged kernel.
static int g_wgrad_dma = [] { const char* e = getenv("STYLER_WGRAD_DMA"); return e ? atoi(e) : 2; }();
static int g_wgrad_dma_nst128 = [] { const char* e = getenv("STYLER_WGRAD_DMA_NST128"); return e ? atoi(e) : 2; }();
extern "C" int styler_wgrad_dma_config(int mode, int stages128) {
  const int prev = g_wgrad_dma | (g_wgrad_dma_nst128 << 2);
  if (mode >= 0 && mode <= 2) g_wgrad_dma = mode;
  if (stages128 == 2 || stages128 == 3) g_wgrad_dma_nst128 = stages128;
  return prev;
}
// Round 5: XCD box map of launches with fewer than 8 splits (wgrad_xcd_box_map); 0 = the former tile-major map (A/B runs).
// Measured (profiles/r05_wgrad_map_bench.txt, same-box A/B): the box map does NOT pay -- k = 9 123.5 -> 126.7 us, PostNet k = 5
// 128.6 -> 134.1 us, the step unchanged within noise.  The operands of a launch (55 + 14 MB) live in the 256 MB Infinity Cache, so
// the re-fetches of the tile-major map never reached HBM; what the kernel waits for is not the L2 miss rate.  Default: off.
static int g_wgrad_xcd_map = [] { const char* e = getenv("STYLER_WGRAD_XCDMAP"); return e ? atoi(e) : 0; }();
// Round 5: the k = 5 gradients on the LDS-DMA ring with a 128 (n) x 64 (c) x 5 taps block tile (TA = 2): 16 + 9 KB of operands
// per 64-row chunk for twice the MFMAs of the 64 x 64 tile's 8 + 9 KB (154 -> 210 FLOP per operand byte; the k = 5 kernels ran
// at the CU's L2 -> LDS rate, not at the MFMA rate).  n % 128 == 0, both operands bf16-resident, mode 2 only.
// Measured (same file): PostNet 512 -> 512: 128.6 -> 115.2 us (+11.6 %); 256 -> 256 (16 tiles of 64 x 64: the tall tile doubles
// its split count to 32): 43.6 -> 44.4 us.  Default: on, for gradients of at least 64 tiles of 64 x 64.
// knob 2 (experiment): a ring of FOUR stages (prefetch distance 3 chunks) for the 64 x 64 k = 5 / k = 9 kernels of mode 2
// experiment: the Linear (k = 1) gradients on the register-staged kernel although both operands are bf16-resident (its 128 x 128
// tile issues 8 DMA pieces per 16 MFMAs on the ring: stand-alone the register-staged kernel is 8-27 % faster there)
static int g_wgrad_lin_dma = [] { const char* e = getenv("STYLER_WGRAD_LIN_DMA"); return e ? atoi(e) : 1; }();
static int g_wgrad_ring4 = [] { const char* e = getenv("STYLER_WGRAD_RING4"); return e ? atoi(e) : 0; }();
static int g_wgrad_k5_tall = [] { const char* e = getenv("STYLER_WGRAD_K5_TALL"); return e ? atoi(e) : 1; }();
extern "C" int styler_wgrad_tune(int knob, int value) {
  int* const k = knob == 0 ? &g_wgrad_xcd_map : knob == 1 ? &g_wgrad_k5_tall : knob == 2 ? &g_wgrad_ring4 : nullptr;
  if (!k) return STYLER_EINVAL;
  const int prev = *k;
  if (value == 0 || value == 1 || (value == 2 && knob == 1)) *k = value;
  return prev;
}
static bool wgrad_k5_tall(int n, int cin, int kw, int prec, int io_flags);
static void wgrad_tile(int n, int cin, int kw, int prec, int* TA, int* TB);
// K groups per block of a launch with these operand formats (1: every other kernel)
static int wgrad_kgroups(int n, int cin, int kw, int prec, int io_flags) {
  if (g_wgrad_dma != 2 || prec != STYLER_PREC_BF16) return 1;
  if (!(io_flags & STYLER_IO_Y_BF16) || !(io_flags & STYLER_IO_X_BF16) || (n & 7) || (cin & 7)) return 1;
  int TA, TB;
  wgrad_tile(n, cin, kw, prec, &TA, &TB);
  if (kw == 1) return (TA == 2 && TB == 2 && g_wgrad_lin_dma) ? 2 : 1;
  return ((kw == 5 || kw == 9) && TA == 1 && TB == 1) ? 2 : 1;
}
static bool wgrad_k5_tall(int n, int cin, int kw, int prec, int io_flags) {
  return g_wgrad_k5_tall && g_wgrad_dma == 2 && prec == STYLER_PREC_BF16 && kw == 5 && (io_flags & STYLER_IO_Y_BF16) &&
         (io_flags & STYLER_IO_X_BF16) && !(n & 127) && !(cin & 7) && ((n / 64) * ((cin + 63) / 64) >= 64 || g_wgrad_k5_tall >= 2);
}

// STYLER_IO_X3CAT launches exist on the LDS-DMA ring only: both parts bf16-resident, whole 16-byte pieces, a ring kernel
// for the (taps, tile) combination.
static bool wgrad_x3cat_ok(int n, int cin, int kw, int pad_left) {
  if (!g_wgrad_dma || (n & 7) || (cin & 7)) return false;
  int TA, TB;
  wgrad_tile(n, cin, kw, STYLER_PREC_BF16, &TA, &TB);
  if (kw == 1) return pad_left == 0 && TA == 2 && TB == 2;
  return (kw == 5 || kw == 9) && pad_left == kw / 2 && TA == 1 && TB == 1;
}
extern "C" int styler_wgrad_x3cat_ok(int n, int cin, int kw, int pad_left) { return wgrad_x3cat_ok(n, cin, kw, pad_left) ? 1 : 0; }

static void wgrad_plan(int B, int L, int n, int cin, int kw, int pad_left, int prec, int* Be, int* Le, int* cpi, int* cps,
                       int* splits, int want_splits = 0, int kg = 1, int kcat = 1, bool tall = false) {
  int TA, TB;
  wgrad_tile(n, cin, kw, prec, &TA, &TB);
  if (tall) TA = 2;
  const int fa = 64 * TA, fb = 64 * TB;
  const int nt = (n + fa - 1) / fa, ct = (cin + fb - 1) / fb;
  *Be = B; *Le = L;
  int64_t nchunks;
  if (prec == STYLER_PREC_BF16) {
    if (kw == 1 && pad_left == 0) { *Le = (int)((int64_t)B * L); *Be = 1; }   // no shifts: one flat item
    *cpi = (*Le + WB_BK - 1) / WB_BK;
    nchunks = (int64_t)(*Be) * (*cpi) * kcat;        // (kcat = 3: STYLER_IO_X3CAT, three parts along the contraction axis)
  } else {
    *cpi = 0;
    nchunks = ((int64_t)B * L + WG_BK - 1) / WG_BK;
  }
  // ~2 blocks per CU for a problem that is launched alone; every extra split costs a partial tile round trip through HBM
  // (the partials of one backward pass add up to GBs).  The small Linear gradients that run as members of ONE grouped
  // launch (wgrad_tr_group_kernel: bf16, kw = 1, 64x64 tile) share the chip, so they get a quarter of the splits.
  static const int grp_target = [] { const char* e = getenv("STYLER_WGRAD_GROUP_BLOCKS"); return e ? atoi(e) : 128; }();
  static const int big_target = [] { const char* e = getenv("STYLER_WGRAD_BLOCKS"); return e ? atoi(e) : 512; }();
  const bool group_member = prec == STYLER_PREC_BF16 && kw == 1 && ((n + 63) / 64) * ((cin + 63) / 64) < 48;
  const int target = (group_member ? grp_target : big_target) / kg;     // two K groups per block: half the blocks
  // as many splits as fit the target WITHOUT exceeding it (a few blocks over a full round of the chip run a second round)
  int64_t sp = prec == STYLER_PREC_BF16 ? target / (nt * ct) : (target + nt * ct - 1) / (nt * ct);
  if (want_splits > 0 && want_splits < sp) sp = want_splits;               // grouped launch: the group fills the chip
  if (sp > nchunks / 4) sp = nchunks / 4;
  if (sp < 1) sp = 1;
  *cps = (int)((nchunks + sp - 1) / sp);
  *splits = (int)((nchunks + *cps - 1) / *cps);
}

extern "C" int64_t styler_wgrad_workspace_bytes_io(int B, int L, int n, int cin, int kw, int pad_left, int prec, int io_flags) {
  if (B <= 0 || L <= 0 || n <= 0 || cin <= 0 || kw <= 0) return 0;
  int Be, Le, cpi, cps, splits;
  wgrad_plan(B, L, n, cin, kw, pad_left, prec, &Be, &Le, &cpi, &cps, &splits, 0, wgrad_kgroups(n, cin, kw, prec, io_flags),
             (io_flags & STYLER_IO_X3CAT) ? 3 : 1, wgrad_k5_tall(n, cin, kw, prec, io_flags));
  return (int64_t)splits * n * kw * cin * 4;
}
extern "C" int64_t styler_wgrad_workspace_bytes(int B, int L, int n, int cin, int kw, int pad_left, int prec) {
  return styler_wgrad_workspace_bytes_io(B, L, n, cin, kw, pad_left, prec, 0);
}

static int wgrad_impl(const float* dz, int64_t lddz, const float* x, int64_t ldx, float* dw, float* db, float* db2,
                      int64_t stride_n, int64_t stride_c, int64_t stride_j, int B, int L, int n, int cin, int kw,
                      int pad_left, int prec, void* workspace, int defer_reduce, const int32_t* rowinfo,
                      const int32_t* chunktab, const int64_t* counts, int io_flags, void* stream) {
  if (!dz || !x || !dw || !workspace || B <= 0 || L <= 0 || n <= 0 || cin <= 0 || (db2 && !db)) return STYLER_EINVAL;
  const bool dz16 = io_flags & STYLER_IO_Y_BF16, x16 = io_flags & STYLER_IO_X_BF16;
  if (dz16 || x16) {                                 // bf16-resident operands: the two shapes of the FFN sublayer, and the
    // k = 5 convolutions of the AudioEncoder / PostNet stacks (any combination of the two operands)
    const bool ok = prec == STYLER_PREC_BF16 && (!dz16 || !(lddz & 7)) && (!x16 || !(ldx & 7)) &&
                    ((kw == 1 && pad_left == 0) || (dz16 && kw == 9) || kw == 5);
    if (!ok) return STYLER_EINVAL;
  }
  if (kw != 1 && kw != 3 && kw != 5 && kw != 9) return STYLER_EINVAL;
  if ((lddz & 3) || (ldx & 3) || (n & 3) || ldx < ((cin + 3) & ~3) || ((uintptr_t)dz & 15) || ((uintptr_t)x & 15)) return STYLER_EALIGN;
  int TA, TB;
  wgrad_tile(n, cin, kw, prec, &TA, &TB);
  if (wgrad_k5_tall(n, cin, kw, prec, io_flags)) TA = 2;
  const int fa = 64 * TA, fb = 64 * TB;
  const int nt = (n + fa - 1) / fa, ct = (cin + fb - 1) / fb;
  hipStream_t st = (hipStream_t)stream;
  int Be, Le, cpi, cps, splits;
  const int kg = wgrad_kgroups(n, cin, kw, prec, io_flags);
  const bool x3cat = io_flags & STYLER_IO_X3CAT;
  if (x3cat && !(prec == STYLER_PREC_BF16 && dz16 && x16 && wgrad_x3cat_ok(n, cin, kw, pad_left) && !(lddz & 7) && !(ldx & 7) &&
                 lddz >= 2 * (int64_t)n && ldx >= 2 * (int64_t)cin))
    return STYLER_EINVAL;
  const bool tall = wgrad_k5_tall(n, cin, kw, prec, io_flags);
  wgrad_plan(B, L, n, cin, kw, pad_left, prec, &Be, &Le, &cpi, &cps, &splits, 0, kg, x3cat ? 3 : 1, tall);
  static const int desctab_env = [] { const char* e = getenv("STYLER_WGRAD_DESCTAB"); return e ? atoi(e) : 1; }();
  static const int ilv_env = [] { const char* e = getenv("STYLER_WGRAD_ILV"); return e ? atoi(e) : 0; }();
  const int legacy_map = (g_wgrad_xcd_map ? 0 : 0x1000) | (desctab_env ? 0 : 0x2000) | (ilv_env ? 0x4000 : 0);
  const int pad_cat = (pad_left & 0xff) | (x3cat ? 0x400 : 0) | ((io_flags & STYLER_IO_DB_SLOTS) ? 0x800 : 0) | legacy_map;
  float* ws = reinterpret_cast<float*>(workspace);
  const dim3 grid(nt * ct, (unsigned)splits);
  // low-part flags of the register-staged kernels (fp32-typed operands only; see wgrad_tr_body)
  const bool db_slots = io_flags & STYLER_IO_DB_SLOTS;
  if (db_slots && (!db || db2 || !defer_reduce)) return STYLER_EINVAL;
  const int pad_k = (pad_left & 0xff) | ((io_flags & STYLER_IO_X_LO) && !x16 ? 0x100 : 0) | ((io_flags & STYLER_IO_DZ_LO) && !dz16 ? 0x200 : 0) |
                    (db_slots ? 0x800 : 0) | legacy_map;
  if (prec == STYLER_PREC_BF16) {
    const int tiles = nt * ct;
    const dim3 grid1((unsigned)(splits >= 8 ? (tiles * splits + 7) / 8 * 8 : tiles * splits));
#define WT_LAUNCH(K, A_, B_) hipLaunchKernelGGL((wgrad_tr_kernel<K, A_, B_>), grid1, dim3(256), 0, st, dz, lddz, x, ldx, db, \
                                                db2, Be, Le, n, cin, pad_k, ct, cpi, cps, tiles, splits, ws, \
                                                kw > 1 ? reinterpret_cast<const int4*>(chunktab) : nullptr, counts)
    // both operands bf16-resident, rows and feature counts in whole 16-byte pieces: the LDS-DMA ring (wgrad_dma_kernel)
    const bool dma = g_wgrad_dma && dz16 && x16 && !(n & 7) && !(cin & 7) && !((uintptr_t)dz & 15) && !((uintptr_t)x & 15);
    if (x3cat && !dma) return STYLER_EALIGN;
#define WD_LAUNCH(K, A_, B_, S_, G_) hipLaunchKernelGGL((wgrad_dma_kernel<K, A_, B_, S_, G_>), grid1, dim3(256 * G_), 0, st, dz, lddz, \
                                                        x, ldx, db, db2, Be, Le, n, cin, pad_cat, ct, cpi, cps, tiles, splits, ws,   \
                                                        kw > 1 ? reinterpret_cast<const int4*>(chunktab) : nullptr, counts)
    if (dma && kw == 1 && TA == 2 && TB == 2 && g_wgrad_lin_dma) {
      if (kg == 2) WD_LAUNCH(1, 2, 2, 2, 2);
      else if (g_wgrad_dma_nst128 == 3) WD_LAUNCH(1, 2, 2, 3, 1); else WD_LAUNCH(1, 2, 2, 2, 1);
    } else if (dma && kw == 5 && TA == 2 && TB == 1 && tall) {
      WD_LAUNCH(5, 2, 1, 3, 2);
    } else if (dma && kw == 5 && TA == 1 && TB == 1) {
      if (kg == 2 && g_wgrad_ring4) WD_LAUNCH(5, 1, 1, 4, 2);
      else if (kg == 2) WD_LAUNCH(5, 1, 1, 3, 2); else WD_LAUNCH(5, 1, 1, 3, 1);
    } else if (dma && kw == 9 && TA == 1 && TB == 1) {
      if (kg == 2 && g_wgrad_ring4) WD_LAUNCH(9, 1, 1, 4, 2);
      else if (kg == 2) WD_LAUNCH(9, 1, 1, 3, 2); else WD_LAUNCH(9, 1, 1, 3, 1);
    } else
#undef WD_LAUNCH
    if (kw == 1) {
      if (x16 || dz16) {                             // (x16: the FFN hidden activation; dz16: the attention's dqkv)
        if (TA != 2 || TB != 2) return STYLER_EINVAL;
        if (x16 && dz16)                             // (both: the decoder's bf16 residual stream, round 3)
          hipLaunchKernelGGL((wgrad_tr_kernel<1, 2, 2, true, true>), grid1, dim3(256), 0, st, dz, lddz, x, ldx, db, db2, Be, Le,
                             n, cin, pad_k, ct, cpi, cps, tiles, splits, ws, nullptr, counts);
        else if (x16)
          hipLaunchKernelGGL((wgrad_tr_kernel<1, 2, 2, false, true>), grid1, dim3(256), 0, st, dz, lddz, x, ldx, db, db2, Be, Le,
                             n, cin, pad_k, ct, cpi, cps, tiles, splits, ws, nullptr, counts);
        else
          hipLaunchKernelGGL((wgrad_tr_kernel<1, 2, 2, true, false>), grid1, dim3(256), 0, st, dz, lddz, x, ldx, db, db2, Be, Le,
                             n, cin, pad_k, ct, cpi, cps, tiles, splits, ws, nullptr, counts);
      } else if (TA == 2 && TB == 2) WT_LAUNCH(1, 2, 2); else if (TA == 2) WT_LAUNCH(1, 2, 1);
      else if (TB == 2) WT_LAUNCH(1, 1, 2); else WT_LAUNCH(1, 1, 1);
    } else if (kw == 3) {
      if (TA == 2) WT_LAUNCH(3, 2, 1); else WT_LAUNCH(3, 1, 1);
    } else if (kw == 5) {
#define WT5(D_, X_) hipLaunchKernelGGL((wgrad_tr_kernel<5, 1, 1, D_, X_>), grid1, dim3(256), 0, st, dz, lddz, x, ldx, db, db2, Be, \
                                       Le, n, cin, pad_k, ct, cpi, cps, tiles, splits, ws,                                     \
                                       reinterpret_cast<const int4*>(chunktab), counts)
      if (dz16 || x16) {
        if (TA != 1 || TB != 1) return STYLER_EINVAL;
        if (dz16 && x16) WT5(true, true); else if (dz16) WT5(true, false); else WT5(false, true);
      } else if (TA == 2) WT_LAUNCH(5, 2, 1); else WT_LAUNCH(5, 1, 1);
#undef WT5
    } else {
      if (dz16 && x16)
        hipLaunchKernelGGL((wgrad_tr_kernel<9, 1, 1, true, true>), grid1, dim3(256), 0, st, dz, lddz, x, ldx, db, db2, Be, Le,
                           n, cin, pad_k, ct, cpi, cps, tiles, splits, ws, reinterpret_cast<const int4*>(chunktab), counts);
      else if (dz16)
        hipLaunchKernelGGL((wgrad_tr_kernel<9, 1, 1, true, false>), grid1, dim3(256), 0, st, dz, lddz, x, ldx, db, db2, Be, Le,
                           n, cin, pad_k, ct, cpi, cps, tiles, splits, ws, reinterpret_cast<const int4*>(chunktab), counts);
      else
        WT_LAUNCH(9, 1, 1);
    }
#undef WT_LAUNCH
  } else {
#define WG_LAUNCH(K) hipLaunchKernelGGL(wgrad_kernel<K>, grid, dim3(256), 0, st, dz, lddz, x, ldx, dw, db, db2, stride_n, \
                                        stride_c, stride_j, B, L, n, cin, pad_k, ct, cps, ws, \
                                        reinterpret_cast<const int2*>(rowinfo), counts)
    if (kw == 1) WG_LAUNCH(1); else if (kw == 3) WG_LAUNCH(3); else if (kw == 5) WG_LAUNCH(5); else WG_LAUNCH(9);
#undef WG_LAUNCH
  }
  if (defer_reduce) return launch_status();          // the caller batches all reductions: styler_wgrad_reduce_multi
  const int64_t per = (int64_t)n * kw * cin;
  int64_t rb = (per + 255) / 256; if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, ws, dw, stride_n, stride_c, stride_j, n,
                     cin, kw, splits);
  return launch_status();
}

extern "C" int styler_wgrad(const float* dz, int64_t lddz, const float* x, int64_t ldx, float* dw, float* db,
                            float* db2, int64_t stride_n, int64_t stride_c, int64_t stride_j, int B, int L, int n, int cin,
                            int kw, int pad_left, int prec, void* workspace, int defer_reduce, int io_flags, void* stream) {
  return wgrad_impl(dz, lddz, x, ldx, dw, db, db2, stride_n, stride_c, stride_j, B, L, n, cin, kw, pad_left, prec, workspace,
                    defer_reduce, nullptr, nullptr, nullptr, io_flags, stream);
}

// Packed-rows variant (pack.hip): dz / x hold `rows` rows of capacity, the first counts[0] of them valid, items back to
// back.  Workspace / split count are those of styler_wgrad_workspace_bytes / styler_wgrad_splits for (B = 1, L = rows).
extern "C" int styler_wgrad_packed(const float* dz, int64_t lddz, const float* x, int64_t ldx, float* dw, float* db,
                                   int64_t stride_n, int64_t stride_c, int64_t stride_j, int rows, int n, int cin, int kw,
                                   int prec, void* workspace, int defer_reduce, const int32_t* rowinfo,
                                   const int32_t* chunktab, const int64_t* counts, int io_flags, void* stream) {
  if (!rowinfo || !chunktab || !counts) return STYLER_EINVAL;
  return wgrad_impl(dz, lddz, x, ldx, dw, db, nullptr, stride_n, stride_c, stride_j, 1, rows, n, cin, kw, kw / 2, prec,
                    workspace, defer_reduce, rowinfo, chunktab, counts, io_flags, stream);
}

// Kernel variants a grouped launch can run (index = StylerWgradGroupDesc.variant).
//   0: Linear, 64x64 tile      1: Linear, 128x128 tile      2: Linear, 128x128 tile, x resident as bf16
//   3: k = 3                   4: k = 5                     5: k = 9            6: k = 9, dz resident as bf16
//   7: Linear, 128x128 tile, dz resident as bf16
static int wgrad_variant(int kw, int TA, int TB, bool dz16, bool x16) {
  if (kw == 1) {
    if (dz16) return (TA == 2 && TB == 2) ? (x16 ? 8 : 7) : -1;       // 8: both operands bf16 (the decoder's bf16 stream)
    if (x16) return (TA == 2 && TB == 2) ? 2 : -1;
    if (TA == 1 && TB == 1) return 0;
    if (TA == 2 && TB == 2) return 1;
    return -1;
  }
  if (TA != 1 || TB != 1 || x16) return -1;
  if (kw == 3) return dz16 ? -1 : 3;
  if (kw == 5) return dz16 ? -1 : 4;
  if (kw == 9) return dz16 ? 6 : 5;
  return -1;
}

// Host-side descriptor of one member of a grouped launch (styler_wgrad_group): any bf16-mode weight gradient the
// stand-alone entry points accept (padded or packed rows, conv taps, bf16-resident operands).  `want_splits` > 0 caps the
// split count (0: the stand-alone policy).  The caller provides `workspace` (splits * n * kw * cin floats; it may be NULL in
// a first call that only asks for the plan, then be stored into out->ws) and assigns block_start when it lays out the
// launch.  Returns the member's block count (also out->nblocks), 0 when the problem does not qualify (not bf16 mode, or
// a tile / storage combination without a grouped variant: launch it with styler_wgrad), < 0 on bad arguments.
extern "C" int styler_wgrad_group_desc(StylerWgradGroupDesc* out, const float* dz, int64_t lddz, const float* x, int64_t ldx,
                                       float* db, float* db2, int B, int L, int n, int cin, int kw, int pad_left, int prec,
                                       void* workspace, const int64_t* packed_counts, const int32_t* packed_chunktab,
                                       int io_flags, int want_splits) {
  if (!out || !dz || !x || B <= 0 || L <= 0 || n <= 0 || cin <= 0 || (db2 && !db)) return STYLER_EINVAL;
  if (kw != 1 && kw != 3 && kw != 5 && kw != 9) return STYLER_EINVAL;
  if ((lddz & 3) || (ldx & 3) || (n & 3) || ldx < ((cin + 3) & ~3) || ((uintptr_t)dz & 15) || ((uintptr_t)x & 15)) return STYLER_EALIGN;
  if (prec != STYLER_PREC_BF16) return 0;
  const bool dz16 = io_flags & STYLER_IO_Y_BF16, x16 = io_flags & STYLER_IO_X_BF16;
  if ((dz16 || x16) && ((lddz & 7) || (ldx & 7))) return STYLER_EALIGN;
  if (packed_counts && (B != 1 || pad_left != kw / 2 || (kw > 1 && !packed_chunktab))) return STYLER_EINVAL;
  int TA, TB;
  wgrad_tile(n, cin, kw, prec, &TA, &TB);
  const int variant = wgrad_variant(kw, TA, TB, dz16, x16);
  if (variant < 0) return 0;
  int Be, Le, cpi, cps, splits;
  const bool x3cat = io_flags & STYLER_IO_X3CAT;
  if (x3cat && !(dz16 && x16 && wgrad_x3cat_ok(n, cin, kw, pad_left) && variant == 8 && lddz >= 2 * (int64_t)n && ldx >= 2 * (int64_t)cin))
    return STYLER_EINVAL;
  wgrad_plan(B, L, n, cin, kw, pad_left, prec, &Be, &Le, &cpi, &cps, &splits, want_splits, 1, x3cat ? 3 : 1);
  const int nt = (n + 64 * TA - 1) / (64 * TA), ct = (cin + 64 * TB - 1) / (64 * TB), tiles = nt * ct;
  out->dz = (uint64_t)(uintptr_t)dz; out->x = (uint64_t)(uintptr_t)x; out->db = (uint64_t)(uintptr_t)db;
  out->db2 = (uint64_t)(uintptr_t)db2; out->ws = (uint64_t)(uintptr_t)workspace;
  out->counts = (uint64_t)(uintptr_t)packed_counts;
  out->chunktab = (uint64_t)(uintptr_t)(kw > 1 ? packed_chunktab : nullptr);
  out->lddz = lddz; out->ldx = ldx;
  out->B = Be; out->L = Le; out->n = n; out->cin = cin; out->ct = ct; out->cpi = cpi;
  out->pad_left = (pad_left & 0xff) | ((io_flags & STYLER_IO_X_LO) && !x16 ? 0x100 : 0) | ((io_flags & STYLER_IO_DZ_LO) && !dz16 ? 0x200 : 0) |
                  (x3cat ? 0x400 : 0) | ((io_flags & STYLER_IO_DB_SLOTS) ? 0x800 : 0);
  if ((io_flags & STYLER_IO_DB_SLOTS) && (!db || db2)) return STYLER_EINVAL;
  out->cps = cps; out->tiles = tiles; out->splits = splits; out->block_start = 0;
  out->nblocks = splits >= 8 ? (tiles * splits + 7) / 8 * 8 : tiles * splits;
  out->variant = variant; out->kw = kw;
  return out->nblocks;
}

extern "C" int styler_wgrad_group(const StylerWgradGroupDesc* desc_dev, int count, int total_blocks, int variant,
                                  void* stream) {
  if (!desc_dev || count <= 0 || total_blocks <= 0) return STYLER_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)total_blocks), block(256);
#define WGG(K, A_, B_, D_, X_) hipLaunchKernelGGL((wgrad_tr_group_kernel<K, A_, B_, D_, X_>), grid, block, 0, st, desc_dev, count)
  switch (variant) {
    case 0: WGG(1, 1, 1, false, false); break;
    case 1: WGG(1, 2, 2, false, false); break;
    case 2: WGG(1, 2, 2, false, true); break;
    case 3: WGG(3, 1, 1, false, false); break;
    case 4: WGG(5, 1, 1, false, false); break;
    case 5: WGG(9, 1, 1, false, false); break;
    case 6: WGG(9, 1, 1, true, false); break;
    case 7: WGG(1, 2, 2, true, false); break;
    case 8:
      if (g_wgrad_dma && g_wgrad_lin_dma) hipLaunchKernelGGL(wgrad_dma_group_lin128_kernel, grid, block, 0, st, desc_dev, count);
      else WGG(1, 2, 2, true, true);
      break;
    default: return STYLER_EINVAL;
  }
#undef WGG
  return launch_status();
}

extern "C" int styler_wgrad_splits_io(int B, int L, int n, int cin, int kw, int pad_left, int prec, int io_flags) {
  if (B <= 0 || L <= 0 || n <= 0 || cin <= 0 || kw <= 0) return 0;
  int Be, Le, cpi, cps, splits;
  wgrad_plan(B, L, n, cin, kw, pad_left, prec, &Be, &Le, &cpi, &cps, &splits, 0, wgrad_kgroups(n, cin, kw, prec, io_flags),
             (io_flags & STYLER_IO_X3CAT) ? 3 : 1, wgrad_k5_tall(n, cin, kw, prec, io_flags));
  return splits;
}
extern "C" int styler_wgrad_splits(int B, int L, int n, int cin, int kw, int pad_left, int prec) {
  return styler_wgrad_splits_io(B, L, n, cin, kw, pad_left, prec, 0);
}

// One launch reducing the split-K partials of MANY weight gradients (a whole backward pass): descriptor i covers
// blocks [block_start[i], block_start[i+1]); every block handles 1024 consecutive outputs of its descriptor.
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const StylerWgradDesc* __restrict__ desc, int count,
                                                                 const int32_t* __restrict__ blockmap) {
  int lo = 0, hi = count - 1;                        // last descriptor with block_start <= blockIdx.x
  const int64_t bid = blockIdx.x;
  if (blockmap) lo = blockmap[bid];                  // round 6: the block's descriptor given by the host (styler_wgrad_reduce_multi_map)
  else while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].block_start <= bid) lo = mid; else hi = mid - 1; }
  const StylerWgradDesc d = desc[lo];
  const int64_t per = (int64_t)d.n * d.kw * d.cin;   // multiple of 4 (n % 4 == 0); every slice is 16-byte aligned
  const float* ws = reinterpret_cast<const float*>(d.ws);
  float* dw = reinterpret_cast<float*>(d.dw);
  if (d.kw > 1 && d.stride_j == 1 && d.stride_c == d.kw && !(d.cin & 3)) {
    // Conv taps into the PARAMETER layout [n, cin, kw]: the partials are [split][n][kw][cin].  One block owns (one n, 128
    // channels, all taps): it sums the splits as kw coalesced runs of 128 floats, transposes the [kw][128] tile in LDS and
    // read-modify-writes ONE contiguous run of 128 * kw floats of dw.  (Walked in workspace order, consecutive lanes hit dw
    // with a stride of kw floats: a 5-9x amplified read-modify-write of the whole 118 MB gradient.)
    // (round 4: a block takes up to RW = 512 channels of its n -- for cin <= 512 the whole [kw][cin] slab of a partial, ONE
    // contiguous run per split, instead of kw runs of 512 bytes at a stride of cin floats)
    constexpr int RW = 512;
    __shared__ float tile[9][RW + 1];
    const int ct = (d.cin + RW - 1) / RW;
    const int64_t t = bid - d.block_start;
    const int nn = (int)(t / ct), c0 = (int)(t % ct) * RW;
    const int cn = d.cin - c0 < RW ? d.cin - c0 : RW;               // multiple of 4
    const int q = threadIdx.x & 127, jj = threadIdx.x >> 7;         // 128 float4 columns x 2 taps per pass
    for (int j = jj; j < d.kw; j += 2) {
      if (q * 4 < cn) {
        const float* p = ws + ((int64_t)nn * d.kw + j) * d.cin + c0 + q * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int sp = 0;
        for (; sp + 4 <= d.splits; sp += 4) {          // four partials in flight per thread (a one-at-a-time loop is
          const float4 a = *reinterpret_cast<const float4*>(p + (int64_t)sp * per);      // one memory round trip per split)
          const float4 b = *reinterpret_cast<const float4*>(p + (int64_t)(sp + 1) * per);
          const float4 c = *reinterpret_cast<const float4*>(p + (int64_t)(sp + 2) * per);
          const float4 e = *reinterpret_cast<const float4*>(p + (int64_t)(sp + 3) * per);
          s.x += (a.x + b.x) + (c.x + e.x); s.y += (a.y + b.y) + (c.y + e.y);
          s.z += (a.z + b.z) + (c.z + e.z); s.w += (a.w + b.w) + (c.w + e.w);
        }
        for (; sp < d.splits; ++sp) {
          const float4 a = *reinterpret_cast<const float4*>(p + (int64_t)sp * per);
          s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        tile[j][q * 4 + 0] = s.x; tile[j][q * 4 + 1] = s.y; tile[j][q * 4 + 2] = s.z; tile[j][q * 4 + 3] = s.w;
      }
    }
    __syncthreads();
    float* out = dw + (int64_t)nn * d.stride_n + (int64_t)c0 * d.kw;
    for (int i = threadIdx.x; i < cn * d.kw; i += 256) {
      const int c = i / d.kw, j = i - c * d.kw;
      out[i] += tile[j][c];
    }
    return;
  }
  if (per & 3) {
    // a vector whose length is not a multiple of 4 (slot folds of tiny parameter gradients: the classifier's two output
    // biases, the predictor tail's scalar bias -- round 4): one thread per element, splits in order
    const int64_t i = (bid - d.block_start) * 1024 + threadIdx.x;
    for (int64_t e = i; e < per && e < i + 1024; e += 256) {
      float t = 0.f;
      for (int sp = 0; sp < d.splits; ++sp) t += ws[(int64_t)sp * per + e];
      const int c = (int)(e % d.cin); const int j = (int)((e / d.cin) % d.kw); const int64_t nn = e / ((int64_t)d.cin * d.kw);
      dw[nn * d.stride_n + c * d.stride_c + j * d.stride_j] += t;
    }
    return;
  }
  if (per <= 512 && d.splits >= 16) {
    // a short vector with many partials (LayerNorm's parameter gradients: one slot per block of the backward kernel, 256
    // slots of 256 floats): one block, but all of its threads -- 256 / (per / 4) groups share the slots, LDS joins them
    __shared__ float4 part[256];
    const int nq = (int)(per / 4), groups = 256 / nq;
    const int g = threadIdx.x / nq, qi = threadIdx.x - g * nq;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < groups) {
      const float* p = ws + qi * 4 + (int64_t)g * per;
      int sp = g;
      for (; sp + 3 * groups < d.splits; sp += 4 * groups) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + (int64_t)groups * per);
        const float4 c = *reinterpret_cast<const float4*>(p + 2 * (int64_t)groups * per);
        const float4 e = *reinterpret_cast<const float4*>(p + 3 * (int64_t)groups * per);
        s.x += (a.x + b.x) + (c.x + e.x); s.y += (a.y + b.y) + (c.y + e.y);
        s.z += (a.z + b.z) + (c.z + e.z); s.w += (a.w + b.w) + (c.w + e.w);
        p += 4 * (int64_t)groups * per;
      }
      for (; sp < d.splits; sp += groups) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        p += (int64_t)groups * per;
      }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (g == 0) {
      for (int gg = 1; gg < groups; ++gg) {
        const float4 a = part[gg * nq + qi];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      }
      const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t ii = (int64_t)qi * 4 + k;
        const int c = (int)(ii % d.cin); const int j = (int)((ii / d.cin) % d.kw); const int64_t nn = ii / ((int64_t)d.cin * d.kw);
        dw[nn * d.stride_n + c * d.stride_c + j * d.stride_j] += v[k];
      }
    }
    return;
  }
  const int64_t i = (bid - d.block_start) * 1024 + threadIdx.x * 4;      // four consecutive outputs per thread: the
  if (i >= per) return;                                                   // partials stream as 16-byte loads
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* p = ws + i;
  int sp = 0;
  for (; sp + 4 <= d.splits; sp += 4) {              // four independent loads in flight per thread
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + per);
    const float4 c = *reinterpret_cast<const float4*>(p + 2 * per), e = *reinterpret_cast<const float4*>(p + 3 * per);
    s.x += (a.x + b.x) + (c.x + e.x); s.y += (a.y + b.y) + (c.y + e.y);
    s.z += (a.z + b.z) + (c.z + e.z); s.w += (a.w + b.w) + (c.w + e.w);
    p += 4 * per;
  }
  for (; sp < d.splits; ++sp) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    p += per;
  }
  const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t ii = i + k;
    const int c = (int)(ii % d.cin); const int j = (int)((ii / d.cin) % d.kw); const int64_t nn = ii / ((int64_t)d.cin * d.kw);
    dw[nn * d.stride_n + c * d.stride_c + j * d.stride_j] += v[k];
  }
}

// blocks a descriptor of styler_wgrad_reduce_multi owns (the host lays block_start out with this)
extern "C" int64_t styler_wgrad_reduce_blocks(int n, int cin, int kw, int64_t stride_c, int64_t stride_j) {
  if (n <= 0 || cin <= 0 || kw <= 0) return 0;
  if (kw > 1 && stride_j == 1 && stride_c == kw && !(cin & 3)) return (int64_t)n * ((cin + 511) / 512);
  return ((int64_t)n * cin * kw + 1023) / 1024;
}

extern "C" int styler_wgrad_reduce_multi(const StylerWgradDesc* desc_dev, int count, int64_t total_blocks, void* stream) {
  if (!desc_dev || count <= 0 || total_blocks <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, desc_dev,
                     count, (const int32_t*)nullptr);
  return launch_status();
}
extern "C" int styler_wgrad_reduce_multi_map(const StylerWgradDesc* desc_dev, int count, int64_t total_blocks, const int32_t* blockmap,
                                             void* stream) {
  if (!desc_dev || count <= 0 || total_blocks <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, desc_dev,
                     count, blockmap);
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
  // 32 rows per block: enough blocks to cover the chip for the S-domain shapes without flooding the atomics
  int rpb = 32;
  if (rows / rpb > 2048) rpb = (int)((rows + 2047) / 2048);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(C / 4 >= 256 ? 256 : 64), 0,
                     (hipStream_t)stream, dz, lddz, out, out2, rows, C, rpb);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// dst[c, j', nn] = src[nn, c, kw-1-j']     (src = parameter layout [n, cin, kw]; kw = 1: plain transpose)
template <typename OutT>
__global__ void repack_bwd_kernel(const float* __restrict__ src, OutT* __restrict__ dst, int n, int cin, int kw) {
  const int64_t total = (int64_t)n * cin * kw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int nn = (int)(i % n); const int j = (int)((i / n) % kw); const int64_t c = i / ((int64_t)n * kw);
    const float v = src[((int64_t)nn * cin + c) * kw + (kw - 1 - j)];
    if constexpr (sizeof(OutT) == 2) dst[i] = (OutT)f32_to_bf16_bits(v); else dst[i] = v;
  }
}

extern "C" int styler_repack_weight_bwd(const float* src, void* dst, int n, int cin, int kw, int out_bf16, void* stream) {
  if (!src || !dst || n <= 0 || cin <= 0 || kw <= 0) return STYLER_EINVAL;
  const int64_t total = (int64_t)n * cin * kw;
  int64_t blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  if (out_bf16)
    hipLaunchKernelGGL(repack_bwd_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<uint16_t*>(dst), n, cin, kw);
  else
    hipLaunchKernelGGL(repack_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<float*>(dst), n, cin, kw);
  return launch_status();
}
