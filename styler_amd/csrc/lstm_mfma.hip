// Recurrent half of the BiLSTM layers on the matrix cores (round 6; bf16 and bf16x3 modes -- fp32 mode keeps lstm.hip).
// Replaces: nn.LSTM(.., bidirectional=True) forward and autograd, modules.py:100-101,117,132,147,162,179-182.
//
// Why.  lstm.hip gives every (item, direction) its own block and every gate row a thread that keeps a row of W_hh in
// registers: per step H/2 packed FMAs per thread, 768 blocks = 3 per CU, and with 15 waves per CU the step time IS the VALU
// instruction count (1.4 us per step forward, 2.2 us backward: 0.43 ms of the 9.2 ms training step for ~1 GFLOP).  The
// recurrence is a chain of tiny GEMMs -- h_{t-1} [rows, H] x W_hh^T [H, 4H] forward, dgates_{t+1} [rows, 4H] x W_hh [4H, H]
// backward -- so here a block takes a few items of one (layer, direction) and runs each step as v_mfma_f32_16x16x32_bf16:
//   * wave w owns hidden units 16 w .. 16 w + 15 and keeps ITS slice of W_hh as MFMA B fragments in registers for the whole
//     sequence (forward: the unit's four gate rows, 4 x KS fragments; backward: the unit's column, 4H / 32 fragments);
//   * the C layout of the 16 x 16 tile gives a lane the four gates of (4 rows, 1 unit): the gate non-linearities, the cell
//     update and the gate gradients are lane-local, no cross-thread exchange;
//   * the only exchange is the step's A operand -- h_t (forward) or dgates_t (backward) as bf16 in LDS, double-buffered, one
//     barrier per step;
//   * gx / gates / cell / dout rows of later steps are fetched two steps ahead; outputs leave with plain stores;
//   * RPL = rows per lane.  The 16 x 16 tile has room for 16 items, but what paces a step is the gate arithmetic (5
//     transcendental pairs per (item, unit)), not the 12 MFMAs: with 16 items a lane evaluates 4 (item, unit) pairs and a
//     5-wave block needs ~2400 SIMD cycles per step (first version: 102 us forward, slower than lstm.hip).  A block therefore
//     fills only rows 4 rg + r, r < RPL, of the tile (RPL = 1: 4 items per block, 192 blocks for the step's 4 x 2 x 96
//     sequences) -- every lane stays busy with ONE pair, the unused tile rows are zeros the MFMA multiplies for free.
// PARTS = 1: bf16 operands, fp32 accumulation (the arithmetic of every other GEMM of the bf16 mode).  PARTS = 3: the bf16x3
// form of DESIGN 3.11 -- h and W_hh split hi + lo, a.w = a_hi w_hi + a_lo w_hi + a_hi w_lo -- for the bf16x3 mode.
// Same saved tensors as lstm.hip (post-activation gates [B,S,2*4H], cell [B,S,2H], out [B,S,2H]; dgp [B,S,2*4H]), so the
// weight / input gradient GEMMs behind it are unchanged.
#include "common.h"

namespace lstm_mfma_detail {

constexpr int MB = 16;                                   // rows of the MFMA tile (items per block = 4 RPL of them)

template <int H> struct LstmGeom {
  static constexpr int KP = (H + 31) / 32 * 32;          // forward contraction, padded to whole MFMA steps (80 -> 96)
  static constexpr int KS = KP / 32;
  static constexpr int LDH = KP + 8;                     // bf16 per LDS row of h (+ 16 B: consecutive rows 4 banks apart)
  static constexpr int KG = 4 * H;                       // backward contraction (320 / 256: whole steps)
  static constexpr int KSG = KG / 32;
  static constexpr int LDG = KG + 8;
  static constexpr int NW = H / 16;                      // waves that own hidden units (5 / 4)
};

__device__ __forceinline__ void lds_barrier() {          // LDS traffic of this wave retired, then the workgroup barrier;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // global loads / stores stay in flight across it
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
  uint32_t u[4] = {cvt_pk_bf16_rne(v[0], v[1]), cvt_pk_bf16_rne(v[2], v[3]), cvt_pk_bf16_rne(v[4], v[5]), cvt_pk_bf16_rne(v[6], v[7])};
  return *reinterpret_cast<const bf16x8*>(u);
}
__device__ __forceinline__ float bf16_hi(float v) { return __uint_as_float(f32_to_bf16_bits(v) << 16); }

// ---------------------------------------------------------------------------------------------------------------------
template <int H, int PARTS, int RPL, bool SAVE>
__device__ __forceinline__ void lstm_mfma_fwd_body(const StylerLstmDesc& d, int B, int S, uint16_t* hb) {
  using G = LstmGeom<H>;
  constexpr int NP = PARTS == 3 ? 2 : 1;                 // LDS images of h: hi (and lo)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * (4 * RPL), dir = blockIdx.y;
  const int col = lane & 15, rg = lane >> 4;
  for (int i = tid; i < NP * 2 * MB * G::LDH / 2; i += blockDim.x) reinterpret_cast<uint32_t*>(hb)[i] = 0u;   // h_0 = 0; K padding
  if (wave >= G::NW) {                                   // surplus wave of a shared launch (H = 64): barriers only
    lds_barrier();
    for (int step = 0; step < S; ++step) lds_barrier();
    return;
  }
  const int u = 16 * wave + col;                         // this lane's hidden unit
  // W_hh rows q H + u (gate q of the unit) as B fragments: lane holds k = 32 kk + 8 rg .. + 7 of column `col`
  bf16x8 whi[4][G::KS], wlo[PARTS == 3 ? 4 : 1][G::KS];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      const float* wp = d.w_hh + ((int64_t)dir * 4 * H + q * H + u) * H;
      float v[8], l[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kk * 32 + 8 * rg + e;
        v[e] = k < H ? wp[k] : 0.f;
        l[e] = v[e] - bf16_hi(v[e]);
      }
      whi[q][kk] = pack8(v);
      if constexpr (PARTS == 3) wlo[q][kk] = pack8(l);
    }
  const int64_t gx_ld = 2 * 4 * H, out_ld = 2 * H;
  // Rows past the batch (a last block with fewer than 4 RPL items) are clamped to item B - 1: they load ITS gx, compute ITS
  // recurrence bit for bit (rows of the tile are independent) and store the same values to the same addresses -- every load
  // and store of the loop is unconditional, which is what lets the compiler count them (a store behind a branch is "maybe
  // zero stores" to its wait-count pass, and the waits in front of the rotation then cover the step's stores as well).
  int64_t grow[RPL], orow[RPL];                          // per row r: offsets of (item, t = 0) in gx / out
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const int b = b0 + RPL * rg + r;                     // (tile row 4 rg + r)
    const int bc = b < B ? b : B - 1;
    grow[r] = (int64_t)bc * S * gx_ld + dir * 4 * H + u;
    orow[r] = (int64_t)bc * S * out_ld + dir * H + u;
  }
  int t = dir ? S - 1 : 0;
  const int dt = dir ? -1 : 1;
  float g0[4][RPL], g1[4][RPL], g2[4][RPL];              // gx of this step, the next, the one after: [gate][row]
  auto load_gx = [&](float (&g)[4][RPL], int tt) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < RPL; ++r) g[q][r] = d.gx[grow[r] + (int64_t)tt * gx_ld + q * H];
  };
  load_gx(g0, t);
  if (S > 1) load_gx(g1, t + dt);
  // The loads above must have LANDED at loop entry as far as the compiler's wait-count pass can tell: with loads pending at
  // the loop header it places `s_waitcnt vmcnt(3..0)` in front of the step's MFMAs for the first iteration's sake, and in
  // steady state those counts wait for the loads the step has just issued -- the whole memory latency in every step (first
  // build: 0.94 us per step).  With the entry state clean, the only waits in the loop sit in front of the rotation at the
  // bottom, one full step behind the loads they cover.
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < RPL; ++r) { asm volatile("" : "+v"(g0[q][r])); asm volatile("" : "+v"(g1[q][r])); }
  float c[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) c[r] = 0.f;
  lds_barrier();
  int cur = 0;
  for (int step = 0; step < S; ++step, t += dt) {
    if (step + 2 < S) load_gx(g2, t + 2 * dt);
    const uint16_t* hc = hb + cur * (MB * G::LDH);
    bf16x8 a[G::KS], al[PARTS == 3 ? G::KS : 1];
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      a[kk] = *reinterpret_cast<const bf16x8*>(hc + col * G::LDH + kk * 32 + 8 * rg);
      if constexpr (PARTS == 3) al[kk] = *reinterpret_cast<const bf16x8*>(hc + 2 * MB * G::LDH + col * G::LDH + kk * 32 + 8 * rg);
    }
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = r < RPL ? g0[q][r < RPL ? r : 0] : 0.f;
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (PARTS == 3) {
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk], wlo[q][kk], acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[kk], whi[q][kk], acc[q], 0, 0, 0);
        }
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk], whi[q][kk], acc[q], 0, 0, 0);
      }
    uint16_t* hn = hb + (cur ^ 1) * (MB * G::LDH);
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const float gi = fast_sigmoid(acc[0][r]), gf = fast_sigmoid(acc[1][r]);      // one exp2 + one rcp per gate
      const float gg = fast_tanh(acc[2][r]);
      const float go = fast_sigmoid(acc[3][r]);
      c[r] = gf * c[r] + gi * gg;
      const float h = go * fast_tanh(c[r]);
      const uint32_t hb16 = cvt_pk_bf16_rne(h, 0.f) & 0xffffu;          // (one v_cvt_pk_bf16_f32; f32_to_bf16_bits branches on NaN)
      hn[(4 * rg + r) * G::LDH + u] = (uint16_t)hb16;
      if constexpr (PARTS == 3) hn[2 * MB * G::LDH + (4 * rg + r) * G::LDH + u] = (uint16_t)cvt_pk_bf16_rne(h - __uint_as_float(hb16 << 16), 0.f);
      d.out[orow[r] + (int64_t)t * out_ld] = h;
      if constexpr (SAVE) {
        d.cell_out[orow[r] + (int64_t)t * out_ld] = c[r];
        float* gp = d.gates_out + grow[r] + (int64_t)t * gx_ld;
        gp[0] = gi; gp[H] = gf; gp[2 * H] = gg; gp[3 * H] = go;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < RPL; ++r) { g0[q][r] = g1[q][r]; g1[q][r] = g2[q][r]; }
    lds_barrier();
    cur ^= 1;
  }
}

struct LstmMfmaArgs { StylerLstmDesc d[4]; };

template <int PARTS, int RPL, bool SAVE>
__global__ __launch_bounds__(320) void lstm_mfma_fwd_kernel(LstmMfmaArgs a, int B, int S) {
  __shared__ __attribute__((aligned(16))) uint16_t hb[(PARTS == 3 ? 2 : 1) * 2 * MB * LstmGeom<80>::LDH];
  const StylerLstmDesc d = a.d[blockIdx.z];
  if (d.H == 80) lstm_mfma_fwd_body<80, PARTS, RPL, SAVE>(d, B, S, hb);
  else lstm_mfma_fwd_body<64, PARTS, RPL, SAVE>(d, B, S, hb);
}

// ---------------------------------------------------------------------------------------------------------------------
// BPTT.  dh_prev[row, u] = sum_j dgp[row, j] W_hh[j, u]: A = the step's gate gradients (bf16, LDS), B = column u of W_hh.
template <int H, int PARTS, int RPL>
__device__ __forceinline__ void lstm_mfma_bwd_body(const StylerLstmBwdDesc& d, int B, int S, uint16_t* sb) {
  using G = LstmGeom<H>;
  constexpr int NP = PARTS == 3 ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * (4 * RPL), dir = blockIdx.y;
  const int col = lane & 15, rg = lane >> 4;
  for (int i = tid; i < NP * 2 * MB * G::LDG / 2; i += blockDim.x) reinterpret_cast<uint32_t*>(sb)[i] = 0u;   // dgates behind the last step = 0
  if (wave >= G::NW) {
    lds_barrier();
    for (int step = 0; step < S; ++step) lds_barrier();
    return;
  }
  const int u = 16 * wave + col;
  bf16x8 whi[G::KSG], wlo[PARTS == 3 ? G::KSG : 1];
#pragma unroll
  for (int kk = 0; kk < G::KSG; ++kk) {
    const float* wp = d.w_hh + ((int64_t)dir * 4 * H + kk * 32 + 8 * rg) * H + u;       // rows j = 32 kk + 8 rg .. + 7, column u
    float v[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = wp[(int64_t)e * H]; l[e] = v[e] - bf16_hi(v[e]); }
    whi[kk] = pack8(v);
    if constexpr (PARTS == 3) wlo[kk] = pack8(l);
  }
  const int64_t g_ld = 2 * 4 * H, o_ld = 2 * H;
  int64_t grow[RPL], orow[RPL];                          // (rows past the batch: clamped duplicates of item B - 1, see the forward body)
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const int b = b0 + RPL * rg + r;
    const int bc = b < B ? b : B - 1;
    grow[r] = (int64_t)bc * S * g_ld + dir * 4 * H + u;
    orow[r] = (int64_t)bc * S * o_ld + dir * H + u;
  }
  // reverse of the forward processing order
  int t = dir ? 0 : S - 1;
  const int dt = dir ? 1 : -1;
  struct In { float g[4][RPL]; float dov[RPL]; float c[RPL]; };   // gates [gate][row], dout, cell of one step
  In i0, i1, i2;
  auto load_in = [&](In& x, int tt) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const float* gp = d.gates + grow[r] + (int64_t)tt * g_ld;
      x.g[0][r] = gp[0]; x.g[1][r] = gp[H]; x.g[2][r] = gp[2 * H]; x.g[3][r] = gp[3 * H];
      x.dov[r] = d.dout[orow[r] + (int64_t)tt * o_ld];
      x.c[r] = d.cell[orow[r] + (int64_t)tt * o_ld];
    }
  };
  load_in(i0, t);
  if (S > 1) load_in(i1, t + dt);
#pragma unroll
  for (int r = 0; r < RPL; ++r) {                      // (clean wait-count state at the loop header: see the forward body)
#pragma unroll
    for (int q = 0; q < 4; ++q) { asm volatile("" : "+v"(i0.g[q][r])); asm volatile("" : "+v"(i1.g[q][r])); }
    asm volatile("" : "+v"(i0.dov[r])); asm volatile("" : "+v"(i1.dov[r]));
    asm volatile("" : "+v"(i0.c[r])); asm volatile("" : "+v"(i1.c[r]));
  }
  float dc_next[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) dc_next[r] = 0.f;
  lds_barrier();
  int cur = 0;
  for (int step = 0; step < S; ++step, t += dt) {
    if (step + 2 < S) load_in(i2, t + 2 * dt);
    const uint16_t* sc = sb + cur * (MB * G::LDG);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < G::KSG; ++kk) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(sc + col * G::LDG + kk * 32 + 8 * rg);
      f32x4& acc = (kk & 1) ? acc1 : acc0;
      if constexpr (PARTS == 3) {
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(sc + 2 * MB * G::LDG + col * G::LDG + kk * 32 + 8 * rg);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wlo[kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, whi[kk], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, whi[kk], acc, 0, 0, 0);
    }
    uint16_t* sn = sb + (cur ^ 1) * (MB * G::LDG);
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const float gi = i0.g[0][r], gf = i0.g[1][r], gg = i0.g[2][r], go = i0.g[3][r];
      const float c_prev = step + 1 < S ? i1.c[r] : 0.f;   // previous step in forward processing order = the walk's next
      const float dh = i0.dov[r] + (acc0[r] + acc1[r]);
      const float tc = fast_tanh(i0.c[r]);
      const float d_o = dh * tc;
      const float dc = dc_next[r] + dh * go * (1.f - tc * tc);
      const float di = dc * gg, dg = dc * gi, df = dc * c_prev;
      dc_next[r] = dc * gf;
      const float p[4] = {di * gi * (1.f - gi), df * gf * (1.f - gf), dg * (1.f - gg * gg), d_o * go * (1.f - go)};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t pb = cvt_pk_bf16_rne(p[q], 0.f) & 0xffffu;
        sn[(4 * rg + r) * G::LDG + q * H + u] = (uint16_t)pb;
        if constexpr (PARTS == 3) sn[2 * MB * G::LDG + (4 * rg + r) * G::LDG + q * H + u] = (uint16_t)cvt_pk_bf16_rne(p[q] - __uint_as_float(pb << 16), 0.f);
      }
      float* gp = d.dgp + grow[r] + (int64_t)t * g_ld;
      gp[0] = p[0]; gp[H] = p[1]; gp[2 * H] = p[2]; gp[3 * H] = p[3];
    }
    i0 = i1; i1 = i2;
    lds_barrier();
    cur ^= 1;
  }
}

struct LstmMfmaBwdArgs { StylerLstmBwdDesc d[4]; };

template <int PARTS, int RPL>
__global__ __launch_bounds__(320) void lstm_mfma_bwd_kernel(LstmMfmaBwdArgs a, int B, int S) {
  __shared__ __attribute__((aligned(16))) uint16_t sb[(PARTS == 3 ? 2 : 1) * 2 * MB * LstmGeom<80>::LDG];
  const StylerLstmBwdDesc d = a.d[blockIdx.z];
  if (d.H == 80) lstm_mfma_bwd_body<80, PARTS, RPL>(d, B, S, sb);
  else lstm_mfma_bwd_body<64, PARTS, RPL>(d, B, S, sb);
}

}  // namespace lstm_mfma_detail
using namespace lstm_mfma_detail;

// Launched with RPL = 1 (4 items per block).  Same box, four LSTMs per launch at B = 96, S = 60 (tools/lstm_bench.py,
// profiles/r06_lstm_bench.txt): forward RPL 4 / 2 / 1 = 94 / 59 / 57 us, backward 106 / 74 / 50 us (lstm.hip: 87 / 103 us).

extern "C" int styler_lstm_bidir_multi_mfma(const StylerLstmDesc* descs, int count, int B, int S, int parts, void* stream) {
  if (!descs || count <= 0 || count > 4 || B <= 0 || S <= 0 || (parts != 1 && parts != 3)) return STYLER_EINVAL;
  LstmMfmaArgs a;
  for (int i = 0; i < count; ++i) {
    a.d[i] = descs[i];
    if (!descs[i].gx || !descs[i].w_hh || !descs[i].out || (descs[i].H != 64 && descs[i].H != 80)) return STYLER_EINVAL;
  }
  int save = 0;                                          // the tensors BPTT reads: all or none, for every member
  for (int i = 0; i < count; ++i) save += (descs[i].cell_out != nullptr) + (descs[i].gates_out != nullptr);
  if (save != 0 && save != 2 * count) return STYLER_EINVAL;
  const dim3 grid((B + 3) / 4, 2, count);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_FWD(P, SV) hipLaunchKernelGGL((lstm_mfma_fwd_kernel<P, 1, SV>), grid, dim3(320), 0, st, a, B, S)
  if (parts == 3) { if (save) LAUNCH_FWD(3, true); else LAUNCH_FWD(3, false); }
  else { if (save) LAUNCH_FWD(1, true); else LAUNCH_FWD(1, false); }
#undef LAUNCH_FWD
  return launch_status();
}

extern "C" int styler_lstm_bidir_bwd_multi_mfma(const StylerLstmBwdDesc* descs, int count, int B, int S, int parts, void* stream) {
  if (!descs || count <= 0 || count > 4 || B <= 0 || S <= 0 || (parts != 1 && parts != 3)) return STYLER_EINVAL;
  LstmMfmaBwdArgs a;
  for (int i = 0; i < count; ++i) {
    a.d[i] = descs[i];
    if (!descs[i].dout || !descs[i].gates || !descs[i].cell || !descs[i].w_hh || !descs[i].dgp ||
        (descs[i].H != 64 && descs[i].H != 80))
      return STYLER_EINVAL;
  }
  const dim3 grid((B + 3) / 4, 2, count);
  hipStream_t st = (hipStream_t)stream;
  if (parts == 3) hipLaunchKernelGGL((lstm_mfma_bwd_kernel<3, 1>), grid, dim3(320), 0, st, a, B, S);
  else hipLaunchKernelGGL((lstm_mfma_bwd_kernel<1, 1>), grid, dim3(320), 0, st, a, B, S);
  return launch_status();
}
