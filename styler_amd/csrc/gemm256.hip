// 256 x 256 implicit-GEMM Conv1d / Linear engine for gfx950: eight waves, LDS-DMA operand loads, counted vmcnt.
//
//   y[m, n] = act(scale[n] * sum_{j<kw, c<cin} x[row(m) + j - pad, c] * w[n, j*cin + c] + shift[n]) (+ res)
//
// Same contract as gemm_conv.hip (same GemmArgs, same epilogue features); it takes the launches whose activation operand
// already lives in HBM as bf16, with cin % 64 == 0 and enough 256 x 256 tiles to fill the chip more than once
// (styler_gemm256_try).  Replaces: nn.Conv1d / nn.Linear forward and dX (transformer/SubLayers.py:72-89,
// transformer/Layers.py:78-118, modules.py:103-161).
//
// Why a second engine.  The 128 x 128 kernel stages operands through registers (global -> VGPR -> ds_write) and meets a
// barrier per K step; at 3 blocks per CU that structure tops out at 0.8-1.0 PFLOP/s.  Here:
//   * block = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32 (v_mfma_f32_32x32x16_bf16): 24
//     ds_read_b128 feed 32 MFMAs per K step (the 64 x 64 wave tile of the other engine needs 32);
//   * operands go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.  The
//     convolution is implicit in the per-lane SOURCE address: K step (chunk cc, tap j) reads activation row m + j - pad,
//     channels cc*64 .. +64; a row that falls outside its own item ('same' zero padding, item boundaries inside the flat
//     rectangle, rows past M) gets an out-of-range offset and the DMA writes zeros;
//   * a K step (BK = 64) is four phases, one output quadrant (64 x 32 per wave, 8 MFMAs) each.  The operands of a step are
//     fetched as four 16 KB UNITS laid out by what a phase consumes -- UB0 / UB1 = the first / second 32 weight rows of
//     every wave column, UA0 / UA1 = the first / second 64 activation rows of both wave rows -- one unit per phase, one
//     step ahead, into the other of two LDS stages (128 KB).  A unit is awaited with `s_waitcnt vmcnt(4)`: two younger
//     units (2 DMA instructions per wave each) stay in flight across the barriers; vmcnt never drains inside the loop;
//   * fragment reads are balanced 8 / 4 / 8 / 4 over the phases: the weight fragments of phase 0 are read one phase early
//     (phase 3 of the step before) into the other of two register sets (251 VGPRs, no spill);
//   * the two wave rows run one barrier apart (wr == 1 waits once more before the loop): on every SIMD the wave of one row
//     issues its 8 MFMAs while the wave of the other row issues ds_reads and DMA -- the matrix pipe alternates between them.
//
// Split-K = 2 (launches with fewer data-carrying tiles than CUs and >= 64 K steps): blockIdx.y takes half of the K steps and
// writes raw fp32 partial tiles; gemm256_combine_kernel adds the halves in a fixed order and applies the epilogue.
//
// LDS image of a unit: 128 rows x 128 B (64 bf16 of K), written linearly by the DMA (lane l of piece p lands at
// p*1024 + l*16).  Fragment reads take 16 B per lane from 16 rows at once (ds_read_b128 lane groups pair rows r, r+12,
// r+20..): with the 16-byte chunk index XORed by (row >> 1) & 7 -- applied to the DMA's SOURCE address and to the read
// address, never to the DMA destination -- every lane group touches 16 distinct 16-byte slots.
//
// Ordering rules this file relies on (MI355X_MICROARCH.md, "Two waves per SIMD" item 7):
//   RAW  a unit is read one phase after the phase in which EVERY wave waited (vmcnt) for its own pieces and then met a barrier;
//   WAR  a unit is overwritten at least two phases after the phase of its last ds_read (here: four or more).
#include <cstdlib>
#include <type_traits>
#include "gemm_args.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int UNIT_BYTES = 128 * 128;               // 128 rows x 64 bf16
constexpr int STAGE_BYTES = 4 * UNIT_BYTES;         // UA0, UA1, UB0, UB1
constexpr int U_A0 = 0, U_A1 = UNIT_BYTES, U_B0 = 2 * UNIT_BYTES, U_B1 = 3 * UNIT_BYTES;
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;         // 128 KB
constexpr int CLD = 68;                             // epilogue staging row stride (floats): 64 + 4
constexpr uint32_t OOB = 0x80000000u;
constexpr int FIX_AUX = 0x11;                       // buffer cache policy sc0 | sc1: system scope (the split-K hand-over)               // beyond every descriptor's num_records (< 2^31)

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t lds_byte, uint32_t voff, char* smem) {
  // 64 lanes x 16 B -> LDS bytes [lds_byte, lds_byte + 1024), lane l at + 16 l (wave-uniform destination base)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + lds_byte), 16, voff, 0, 0, 0);
}

}  // namespace

// HT (round 6): MFMA row tiles per wave row -- the block covers H = 64 HT rows (256 or 192).  The LDS image, the DMA
// pieces and the phase / vmcnt schedule are those of the 256-row tile; with HT = 3 the rows 96..127 of each wave row do not
// exist (their DMA pieces carry the out-of-range offset: zero fill, no memory traffic) and phases 2 / 3 issue 4 MFMAs per
// wave instead of 8.  Why: tile quantisation -- M = 42 336 rows x 512 columns (PostNet) is 332 tiles of 256 x 256 = 1.30
// rounds of 256 CUs (paid as 2), but 442 tiles of 192 x 256 = 1.73 rounds of 3/4 the height (paid as 1.5); the 256-column
// AudioEncoder convolutions are 166 tiles (0.65 of a round) against 221 (0.86 of a 3/4-height round).
// FIX (round 6): a split-K = 2 launch whose second-arriving half adds the first one's accumulators and runs the epilogue (a
// template argument: the hand-over code must not cost the other instantiations a register -- they sit at 251 of 256).
template <bool Y16, int HT, bool FIX = false>
__global__ __launch_bounds__(512) void conv_gemm256_kernel(GemmArgs a) {
  static_assert(HT == 3 || HT == 4, "wave rows of 96 or 128 rows");
  constexpr int H = 64 * HT, HW = 32 * HT;                               // block rows, rows per wave row
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];       // the ONLY LDS object of the kernel

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 31, lh = lane >> 5;

  // ---- XCD-aware tile assignment (as gemm_conv.hip): workgroup b runs on XCD b % 8; an XCD walks the n-tiles of its
  // m-tile back to back, so the n-tiles that share an activation tile hit the same L2.
  int mtile, ntile;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, k = bid >> 3;
    mtile = (k / a.nt) * 8 + xcd;
    if (mtile >= a.mt) return;
    ntile = k % a.nt;
  }
  const int64_t M = (int64_t)a.B * a.L;
  const int64_t m0 = (int64_t)mtile * H;
  const int n0 = ntile * BN;
  const int kw = a.kw, pad = a.pad;
  const int ktot = kw * a.cin;
  const int ncc = a.cin / BK;
  // split-K (a.ksplit == 2: launches with fewer tiles than CUs and a long K, the FFN's k = 9 dX): blockIdx.y takes half of
  // the (chunk, tap) steps and stores its raw fp32 accumulators to a.part[blockIdx.y]; gemm256_combine_kernel adds the two
  // halves in a fixed order and applies the epilogue (deterministic: no atomics)
  const int all_steps = ncc * kw;
  const int split = a.ksplit > 1 ? (int)blockIdx.y : 0;
  const int step0 = split * (all_steps / a.ksplit);
  const int nsteps = (split + 1 == a.ksplit ? all_steps : (split + 1) * (all_steps / a.ksplit)) - step0;
  if (a.ksplit > 1 && !FIX) {                       // partial tiles: plain fp32 rows, the epilogue proper runs in the combine pass
    a.y = a.part + (int64_t)split * M * a.n;
    a.ldy = a.n;
    a.scale = a.shift = a.res = nullptr;
    a.mask = nullptr;
    a.act = STYLER_ACT_NONE;
    a.y3 = nullptr;                                   // (the split of the OUTPUT is written by the combine pass)
  }

  // ---- tiles made only of rows at or past their item's length: zeros (packed rows: nothing behind the data is read) ----
  if (a.len) {
    const uint32_t span = (uint32_t)((m0 + H < M ? m0 + H : M) - 1 - m0);
    const uint32_t b0 = (uint32_t)m0 / (uint32_t)a.L, t0 = (uint32_t)m0 - b0 * (uint32_t)a.L;
    if (t0 + span < (uint32_t)a.L && (int64_t)t0 >= a.len[b0]) {
      if (a.rowinfo) return;
      constexpr int QPR = BN / 4;
      for (int i = tid; i < H * QPR; i += 512) {
        const int r = i / QPR, c = n0 + (i - r * QPR) * 4;
        if (m0 + r < M && c < a.n) {
          if (Y16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.y) + (m0 + r) * a.ldy + c) = make_uint2(0u, 0u);
          else *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (m0 + r) * a.ldy + c) = make_float4(0.f, 0.f, 0.f, 0.f);
          if (!Y16 && a.y3) x3_store4(a.y3, m0 + r, c, a.n, a.y3parts, make_float4(0.f, 0.f, 0.f, 0.f));
        }
      }
      return;
    }
  }

  // ---- DMA source addresses.  A unit = 16 pieces of 1 KB (8 rows x 128 B); wave w issues pieces w and w + 8.  Lane l of
  // piece p: unit row i = 8 p + (l >> 3), physical 16-byte chunk l & 7 = logical chunk ^ ((i >> 1) & 7).
  // UA0 row i -> row (i & 63) of wave row i >> 6 = tile row (i >> 6) * HW + (i & 63), UA1: + 64 (rows at or past HW do not
  // exist: HT = 3).  UB0 row i -> weight row (i >> 5) * 64 + (i & 31), UB1: + 32.
  const __amdgpu_buffer_rsrc_t x_rs = [&] {
    int64_t rec = ((M - 1) * a.ldx + (a.x3n1 ? 2 * a.x3n1 * BK : a.cin)) * 2;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)(rec > 0x7fffffff ? 0x7fffffff : rec), 0x00020000);
  }();
  const __amdgpu_buffer_rsrc_t w_rs = [&] {
    const int64_t rec = (int64_t)a.n * ktot * 2;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)rec, 0x00020000);
  }();
  uint32_t va[2][2];                                  // [unit UA0 / UA1][piece]: byte offset of (row, logical chunk) at tap 0, chunk 0
  uint32_t tapmask[2][2];                             // bit j: row + j - pad lies inside the row's own item
  uint32_t vb[2][2];                                  // [unit UB0 / UB1][piece]
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int p = wave + 8 * q;
    const int i = 8 * p + (lane >> 3);
    const int cl = (lane & 7) ^ ((i >> 1) & 7);       // logical 16-byte chunk this lane fetches
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int wrow = (i & 63) + 64 * u;             // row inside its wave row
      const int64_t m = m0 + (i >> 6) * HW + wrow;
      uint32_t bits = 0;
      if (wrow < HW && m < M) {
        int t, rem;
        if (a.rowinfo) { const int2 ri = a.rowinfo[m]; t = ri.x; rem = ri.y; }
        else { t = (int)((uint32_t)m % (uint32_t)a.L); rem = a.L - 1 - t; }
        for (int j = 0; j < kw; ++j) {
          const int o = j - pad;
          if (o >= -t && o <= rem) bits |= 1u << j;
        }
      }
      tapmask[u][q] = bits;
      va[u][q] = (uint32_t)((m - pad) * a.ldx * 2 + cl * 16);        // wraps for the rows before row 0: masked by `bits`
      const int nrow = n0 + (i >> 5) * 64 + (i & 31) + 32 * u;
      vb[u][q] = nrow < a.n ? (uint32_t)((int64_t)nrow * ktot * 2 + cl * 16) : OOB;
    }
  }
  const uint32_t lds_piece0 = (uint32_t)wave * 1024u, lds_piece1 = (uint32_t)(wave + 8) * 1024u;

  // one unit of K step (cc, j) into stage `st`: 2 DMA instructions per wave
  auto issue_a = [&](int u, int cc, int j, uint32_t st) {
    const int ccs = (a.x3n1 && cc >= 2 * a.x3n1) ? cc - 2 * a.x3n1 : cc;      // compact bf16x3 rows [hi | lo]: third product = hi again
    const uint32_t sh = (uint32_t)(j * (int)a.ldx * 2 + ccs * (BK * 2));
    const uint32_t base = st + (u ? U_A1 : U_A0);
    dma16(x_rs, base + lds_piece0, ((tapmask[u][0] >> j) & 1u) ? va[u][0] + sh : OOB, smem);
    dma16(x_rs, base + lds_piece1, ((tapmask[u][1] >> j) & 1u) ? va[u][1] + sh : OOB, smem);
  };
  auto issue_b = [&](int u, int cc, int j, uint32_t st) {
    const uint32_t sh = (uint32_t)((j * a.cin + cc * BK) * 2);
    const uint32_t base = st + (u ? U_B1 : U_B0);
    dma16(w_rs, base + lds_piece0, vb[u][0] == OOB ? OOB : vb[u][0] + sh, smem);
    dma16(w_rs, base + lds_piece1, vb[u][1] == OOB ? OOB : vb[u][1] + sh, smem);
  };

  // ---- fragment read addresses: lane (li, lh), MFMA sub-step s (16 of the 64 k): logical chunk 2 s + lh of unit row
  // base + li; (row >> 1) & 7 == (li >> 1) & 7 because every base is a multiple of 32
  uint32_t ra[4], rb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t ch = (uint32_t)(((2 * s + lh) ^ ((li >> 1) & 7)) * 16);
    ra[s] = (uint32_t)((wr * 64 + li) * 128) + ch;      // + unit + i * 4096 (32 rows) + stage
    rb[s] = (uint32_t)((wc * 32 + li) * 128) + ch;
  }

  f32x16 acc[HT][2];
#pragma unroll
  for (int i = 0; i < HT; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

  // ---- prologue: the four units of step 0 into stage 0 (UB0, UA0 first: phase 0 reads them) ----
  const int cc0 = step0 / kw, j0 = step0 - cc0 * kw;   // K steps run chunk-major, taps inside a chunk
  issue_b(0, cc0, j0, 0u);
  issue_a(0, cc0, j0, 0u);
  issue_b(1, cc0, j0, 0u);
  issue_a(1, cc0, j0, 0u);
#define LDS_FRAG(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 fbA[4], fbB[4];                              // phase-0 weight fragments of the current / the next step
#pragma unroll
  for (int s = 0; s < 4; ++s) fbA[s] = LDS_FRAG(U_B0 + rb[s]);
  if (wr == 1) {                                      // the second wave row runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

#define PHASE_SYNC()                         \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

  // One K step, stage parity PAR (compile time).  Fragment reads per LOAD segment: 8 / 4 / 8 / 4 (the weight fragments of
  // phase 0 are read one phase early, in phase 3 of the step before, into the other of two register sets).
  // DMA issue order: L(t,0) UB0(t+1), L(t,1) UA0(t+1), L(t,2) UB1(t+1), L(t,3) UA1(t+1); every wait is vmcnt(4) = "all but
  // the two youngest units": L(t,0) -> UB1(t), L(t,1) -> UA1(t), L(t,2) -> UB0(t+1), L(t,3) -> UA0(t+1).
  int cc = cc0, j = j0;
  auto k_step = [&](auto par_tag, bf16x8 (&fbc)[4], bf16x8 (&fbn)[4], const bool more) {
    constexpr uint32_t st = decltype(par_tag)::value * STAGE_BYTES;
    constexpr uint32_t sn = STAGE_BYTES - st;           // the other stage: operands of step + 1
    int ccn = cc, jn = j + 1;
    if (jn == kw) { jn = 0; ccn = cc + 1; }
    bf16x8 fa[2][4], fb1[4];

    // ---------------- phase 0: rows 0..63 x cols 0..31 of the wave tile ----------------
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s) fa[i][s] = LDS_FRAG(st + U_A0 + i * 4096 + ra[s]);
    if (more) {
      issue_b(0, ccn, jn, sn);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // UB1(step) landed (UA1(step), UB0(step + 1) may be in flight)
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // UB1(step) landed (UA1(step) may be in flight)
    }
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fbc[s], acc[i][0], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------------- phase 1: rows 0..63 x cols 32..63 ----------------
#pragma unroll
    for (int s = 0; s < 4; ++s) fb1[s] = LDS_FRAG(st + U_B1 + rb[s]);
    if (more) {
      issue_a(0, ccn, jn, sn);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // UA1(step) landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb1[s], acc[i][1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------------- phase 2: rows 64..127 (HT = 3: 64..95) x cols 32..63 ----------------
#pragma unroll
    for (int i = 0; i < HT - 2; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s) fa[i][s] = LDS_FRAG(st + U_A1 + i * 4096 + ra[s]);
    if (more) {
      issue_b(1, ccn, jn, sn);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // UB0(step + 1) landed: phase 3 reads the next step's weight fragments
    }
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < HT - 2; ++i) acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb1[s], acc[2 + i][1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();

    // ---------------- phase 3: rows 64..127 x cols 0..31 (this step's phase-0 weight fragments, still in registers) ----------------
    if (more) {
#pragma unroll
      for (int s = 0; s < 4; ++s) fbn[s] = LDS_FRAG(sn + U_B0 + rb[s]);
      issue_a(1, ccn, jn, sn);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // UA0(step + 1) landed: phase 0 of the next step reads it
    }
    PHASE_SYNC();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < HT - 2; ++i) acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fbc[s], acc[2 + i][0], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PHASE_SYNC();
    cc = ccn; j = jn;
  };
  {
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
      k_step(std::integral_constant<uint32_t, 0>{}, fbA, fbB, true);
      k_step(std::integral_constant<uint32_t, 1>{}, fbB, fbA, step + 2 < nsteps);
    }
    if (step < nsteps) k_step(std::integral_constant<uint32_t, 0>{}, fbA, fbB, false);
  }
  if (wr == 0) {                                        // catch up with the second wave row
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
#undef LDS_FRAG
#undef PHASE_SYNC
  __syncthreads();                                      // every wave is done with the operand stages: LDS becomes the epilogue's

  // ---- split-K = 2 finished in the kernel (round 6; was: two partial tiles to HBM + gemm256_combine_kernel).  The two halves of
  // a tile draw a ticket.  The first stores its accumulators as they sit in the registers (32 x 16 bytes per lane, 1 KB per
  // wave-instruction; the reader is the same wave of the other block, so no layout change), releases a flag and leaves; the
  // second waits for the flag (the first block drew its ticket, so it is resident: the wait is bounded by its stores), adds the
  // image to its own accumulators -- a + b == b + a in fp32, so the result does not depend on which half came first -- and runs
  // the epilogue.  Both ints of the tile are zero again afterwards.
  if constexpr (FIX && HT == 4) {
    const int tile = mtile * a.nt + ntile;
    int* const c2 = a.cnt + 2 * tile;
    int* const role_p = reinterpret_cast<int*>(smem);
    if (tid == 0) *role_p = atomicAdd(c2, 1);
    __syncthreads();
    const int role = __builtin_amdgcn_readfirstlane(*role_p);
    __syncthreads();                                  // (the epilogue re-uses smem)
    const __amdgpu_buffer_rsrc_t p_rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (int64_t)tile * (BM * BN), 0, BM * BN * 4, 0x00020000);
    const uint32_t po = (uint32_t)((wave * 32 * 64 + lane) * 16);
    if (role == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = make_float4(acc[i][jj][4 * q], acc[i][jj][4 * q + 1], acc[i][jj][4 * q + 2], acc[i][jj][4 * q + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const i32x4*>(&v), p_rs, po + ((i * 2 + jj) * 4 + q) * 1024, 0, FIX_AUX);
          }
      // The image and the flag travel with the sc0 sc1 cache policy (written through to / read from the memory side, past
      // the per-XCD L2s): no agent-scope release fence, which on this chip writes back the WHOLE L2 of the XCD (measured:
      // +70 us per launch with __threadfence() here).  vmcnt(0) = the stores are acknowledged; then the flag.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(c2 + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(c2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {                       // eight loads (32 registers) in flight at a time: the accumulators hold 128
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(p_rs, po + ((i * 2 + jj) * 4 + q) * 1024, 0, FIX_AUX);
          const float4 o = *reinterpret_cast<const float4*>(&t);
          acc[i][jj][4 * q] += o.x; acc[i][jj][4 * q + 1] += o.y; acc[i][jj][4 * q + 2] += o.z; acc[i][jj][4 * q + 3] += o.w;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) { c2[0] = 0; c2[1] = 0; }
  }

  // ---- epilogue: per wave, one 32-row MFMA tile row (32 x 64) at a time through LDS -> coalesced 16-byte rows ----
  // C layout of v_mfma_f32_32x32x16: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  // Tail loads (residual, ReLU mask) are batched ahead of the stores, out-of-range rows / columns are dropped by the buffer
  // descriptors, the activation is a template argument of the tail (as in gemm_conv.hip).
  constexpr int LPR = 16;                               // lanes per output row (4 columns each)
  constexpr int RPP = 4;                                // rows per pass of a wave
  constexpr int NP = 32 / RPP;                          // passes per tile row = rows per lane per tile row
  constexpr int Y_ES = Y16 ? 2 : 4;
  constexpr int64_t REC_MAX = (int64_t)1 << 30;
  float* cst = reinterpret_cast<float*>(smem) + wave * (32 * CLD);
  const int lrow = lane / LPR;
  const int c4 = (lane % LPR) * 4;
  const int col = n0 + wc * 64 + c4;
  const bool col_ok = col < a.n;
  const int wrow0 = wr * HW + lrow;                     // tile-relative row of this lane's first row
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok) {
    if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + col);
    if (a.shift) sf = *reinterpret_cast<const float4*>(a.shift + col);
  }
  const bool res_first = a.act & STYLER_ACT_RES_FIRST;
  const int actc = a.act & 0xff;

  // bit (8 i + p) of `live`: the lane's row (tile row i, pass p) lies inside its item's length
  uint32_t live = 0xffffffffu;
  if (a.len) {
    live = 0u;
    if (a.B == 1) {
      const int64_t lim64 = a.len[0] - m0;
      const int lim = lim64 > H ? H : (lim64 < 0 ? 0 : (int)lim64);
#pragma unroll
      for (int q = 0; q < 32; ++q) live |= (uint32_t)(wrow0 + (q >> 3) * 32 + (q & 7) * RPP < lim) << q;
    } else {
      const uint32_t Lu = (uint32_t)a.L;
      const uint32_t b0 = (uint32_t)m0 / Lu, t0 = (uint32_t)m0 - b0 * Lu;
      for (int q = 0; q < 32; ++q) {
        const uint32_t nn = t0 + (uint32_t)(wrow0 + (q >> 3) * 32 + (q & 7) * RPP);
        const uint32_t qq = nn / Lu, r = nn - qq * Lu;
        uint32_t bi = b0 + qq;
        bi = bi < (uint32_t)a.B ? bi : (uint32_t)a.B - 1u;
        live |= (uint32_t)((int)r < reinterpret_cast<const int*>(a.len)[2 * bi]) << q;
      }
    }
  }

  auto tile_rsrc = [&](const void* base, int64_t ld, int es) {
    int64_t rec = ((M - m0 - 1) * ld + a.n) * es;
    rec = rec > REC_MAX ? REC_MAX : rec;
    const char* b = reinterpret_cast<const char*>(base) + m0 * ld * es;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b), 0, (int)rec, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t y_rs = tile_rsrc(a.y, a.ldy, Y_ES);
  const int r_es = a.res16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t r_rs = tile_rsrc(a.res ? (const void*)a.res : a.y, a.res ? a.ldres : a.ldy, a.res ? r_es : Y_ES);
  const int m_es = a.mask16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t m_rs = tile_rsrc(a.mask ? a.mask : a.y, a.mask ? a.ldmask : a.ldy, a.mask ? m_es : Y_ES);
  const uint32_t oy0 = col_ok ? (uint32_t)((wrow0 * (int)a.ldy + col) * Y_ES) : OOB;
  const uint32_t or0 = col_ok ? (uint32_t)((wrow0 * (int)a.ldres + col) * r_es) : OOB;
  const uint32_t om0 = col_ok ? (uint32_t)((wrow0 * (int)a.ldmask + col) * m_es) : OOB;
  const uint32_t ystep = (uint32_t)(RPP * (int)a.ldy * Y_ES), rstep = (uint32_t)(RPP * (int)a.ldres * r_es),
                 mstep = (uint32_t)(RPP * (int)a.ldmask * m_es);
  // round 5, bf16x3: the [hi | lo (| hi)] split of the fp32 output rows, stored next to them (GemmArgs.y3; rows of y3parts * n)
  const bool want3 = !Y16 && a.y3 != nullptr;
  const int ld3 = a.y3parts * a.n;
  const __amdgpu_buffer_rsrc_t y3_rs = [&] {
    int64_t rec = ((M - m0 - 1) * (int64_t)ld3 + ld3) * 2;
    rec = rec > REC_MAX ? REC_MAX : rec;
    const char* b = reinterpret_cast<const char*>(want3 ? a.y3 : reinterpret_cast<uint16_t*>(a.y)) + (want3 ? m0 * (int64_t)ld3 * 2 : 0);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b), 0, want3 ? (int)rec : 0, 0x00020000);
  }();
  const uint32_t o30 = (col_ok && want3) ? (uint32_t)((wrow0 * ld3 + col) * 2) : OOB;
  const uint32_t step3 = (uint32_t)(RPP * ld3 * 2), lo3 = (uint32_t)(a.n * 2);

  auto tail = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int i = 0; i < HT; ++i) {
      if (i) __syncthreads();
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) cst[((r & 3) + 8 * (r >> 2) + 4 * lh) * CLD + jj * 32 + li] = acc[i][jj][r];
      __syncthreads();
      i32x4 rr[NP], mk[NP];
      if (a.res) {
        if (a.res16) {
#pragma unroll
          for (int u = 0; u < NP; ++u) {
            const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r_rs, or0 + (i * 8 + u) * rstep, 0, 0);
            rr[u] = i32x4{(int)((uint32_t)t.x << 16), (int)((uint32_t)t.x & 0xffff0000u), (int)((uint32_t)t.y << 16),
                          (int)((uint32_t)t.y & 0xffff0000u)};
          }
        } else {
#pragma unroll
          for (int u = 0; u < NP; ++u) rr[u] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, or0 + (i * 8 + u) * rstep, 0, 0);
        }
      }
      if (a.mask) {
        if (a.mask16) {
#pragma unroll
          for (int u = 0; u < NP; ++u) {
            const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(m_rs, om0 + (i * 8 + u) * mstep, 0, 0);
            mk[u] = i32x4{t.x, t.y, 0, 0};
          }
        } else {
#pragma unroll
          for (int u = 0; u < NP; ++u) mk[u] = __builtin_amdgcn_raw_buffer_load_b128(m_rs, om0 + (i * 8 + u) * mstep, 0, 0);
        }
      }
      float4 v[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) v[u] = *reinterpret_cast<const float4*>(&cst[(u * RPP + lrow) * CLD + c4]);
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        float4 w = v[u];
        if (res_first && a.res) {
          const float4 q = *reinterpret_cast<const float4*>(&rr[u]);
          w.x += q.x; w.y += q.y; w.z += q.z; w.w += q.w;
        }
        w.x = apply_act(w.x * sc.x + sf.x, ACT); w.y = apply_act(w.y * sc.y + sf.y, ACT);
        w.z = apply_act(w.z * sc.z + sf.z, ACT); w.w = apply_act(w.w * sc.w + sf.w, ACT);
        if (a.mask) {
          if (a.mask16) {
            const uint32_t m0w = (uint32_t)mk[u].x, m1w = (uint32_t)mk[u].y;
            w.x = (int16_t)(m0w & 0xffffu) > 0 ? w.x : 0.f; w.y = (int16_t)(m0w >> 16) > 0 ? w.y : 0.f;
            w.z = (int16_t)(m1w & 0xffffu) > 0 ? w.z : 0.f; w.w = (int16_t)(m1w >> 16) > 0 ? w.w : 0.f;
          } else {
            const float4 q = *reinterpret_cast<const float4*>(&mk[u]);
            w.x = q.x > 0.f ? w.x : 0.f; w.y = q.y > 0.f ? w.y : 0.f;
            w.z = q.z > 0.f ? w.z : 0.f; w.w = q.w > 0.f ? w.w : 0.f;
          }
        }
        if (a.res && !res_first) {
          const float4 q = *reinterpret_cast<const float4*>(&rr[u]);
          w.x += q.x; w.y += q.y; w.z += q.z; w.w += q.w;
        }
        if (!((live >> (i * 8 + u)) & 1u)) w = make_float4(0.f, 0.f, 0.f, 0.f);
        // rows of this lane: wrow0 + 32 i + 4 u  ==  wrow0 + RPP * (8 i + u)
        if (Y16) {
          const i32x2 o = {(int)cvt_pk_bf16_rne(w.x, w.y), (int)cvt_pk_bf16_rne(w.z, w.w)};
          __builtin_amdgcn_raw_buffer_store_b64(o, y_rs, oy0 + (i * 8 + u) * ystep, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const i32x4*>(&w), y_rs, oy0 + (i * 8 + u) * ystep, 0, 0);
          if (want3) {
            uint2 h3, l3;
            x3_split4(w, h3, l3);
            const i32x2 hv = {(int)h3.x, (int)h3.y}, lv = {(int)l3.x, (int)l3.y};
            const uint32_t o3 = o30 + (i * 8 + u) * step3;
            __builtin_amdgcn_raw_buffer_store_b64(hv, y3_rs, o3, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(lv, y3_rs, o3 + lo3, 0, 0);
            if (a.y3parts == 3) __builtin_amdgcn_raw_buffer_store_b64(hv, y3_rs, o3 + 2 * lo3, 0, 0);
          }
        }
      }
    }
  };
  switch (actc) {
    case STYLER_ACT_RELU: tail(std::integral_constant<int, STYLER_ACT_RELU>{}); break;
    case STYLER_ACT_TANH: tail(std::integral_constant<int, STYLER_ACT_TANH>{}); break;
    case STYLER_ACT_LOGCLAMP: tail(std::integral_constant<int, STYLER_ACT_LOGCLAMP>{}); break;
    case STYLER_ACT_LEAKY: tail(std::integral_constant<int, STYLER_ACT_LEAKY>{}); break;
    case STYLER_ACT_CRELU: tail(std::integral_constant<int, STYLER_ACT_CRELU>{}); break;
    default: tail(std::integral_constant<int, STYLER_ACT_NONE>{}); break;
  }
}

// Second pass of a split-K launch: y = scale * (P0 + P1) + shift (+ res), one float4 per thread, fixed summation order.
// `nrows` (packed rows): rows at or past the device row counter are not touched (nothing behind the data is ever read).
template <bool Y16>
__global__ __launch_bounds__(256) void gemm256_combine_kernel(const float* __restrict__ part, int64_t M, int n, int ksplit,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ res, int64_t ldres, void* __restrict__ y,
                                                              int64_t ldy, const int64_t* __restrict__ nrows, int res16,
                                                              uint16_t* __restrict__ y3, int y3parts) {
  const int q = n >> 2;                                                      // float4 per row
  const int64_t lim = nrows ? (nrows[0] < M ? nrows[0] : M) : M;
  const int64_t total = lim * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    float4 v = *reinterpret_cast<const float4*>(part + r * n + c);
    for (int k = 1; k < ksplit; ++k) {
      const float4 w = *reinterpret_cast<const float4*>(part + (int64_t)k * M * n + r * n + c);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if (scale) { const float4 t = *reinterpret_cast<const float4*>(scale + c); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
    if (shift) { const float4 t = *reinterpret_cast<const float4*>(shift + c); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (res) { const float4 t = ldg4(res, r * ldres + c, res16 != 0); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (Y16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(y) + r * ldy + c) = make_uint2(cvt_pk_bf16_rne(v.x, v.y), cvt_pk_bf16_rne(v.z, v.w));
    else *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + r * ldy + c) = v;
    if (!Y16 && y3) x3_store4(y3, r, c, n, y3parts, v);     // round 5: the bf16x3 split of the output rows (GemmArgs.y3)
  }
}

// Eligibility: bf16 MFMA mode with the activation operand stored as bf16, whole 64-channel chunks, 16-byte aligned rows,
// every byte offset below 2^31, at least 8 K steps and at least `min_tiles` (384 = 1.5 per CU) 256 x 256 tiles that carry
// data.  Measured against the 128 x 128 engine on the same bf16 operands, stand-alone (tools/gemm256_bench.py,
// profiles/r03_gemm256_bench.txt): x1.15 at 424 tiles (FFN k = 9, M = 27 060), x1.02-1.06 at 166-848 tiles, x1.11-1.17 at
// 1000-8000 tiles (config 4), x1.55 on a square 4096^3 GEMM (1.26 PFLOP/s); x0.92-0.96 at 106 tiles.  Inside the training
// step (operands warm from the kernel before) the 166 / 332-tile launches of the AudioEncoder / PostNet lose on this engine:
// 10.79 ms per step at a bound of 384, 10.82 at 300, 10.85 at 160 (profiles/r03_g256_in_step.txt), hence 384.  STYLER_GEMM256=0 switches the engine off, STYLER_GEMM256_MIN_TILES
// overrides the bound (styler_gemm256_config at run time).
static int g_enabled = [] { const char* e = getenv("STYLER_GEMM256"); return e ? atoi(e) : 1; }();
static int g_min_tiles = [] { const char* e = getenv("STYLER_GEMM256_MIN_TILES"); return e ? atoi(e) : 384; }();

// Test-only policy overrides (round-3 advisor finding: the tile bound used to double as a mode switch -- 1 meant "take
// everything, never split", 0 "force split-K" -- so tuning STYLER_GEMM256_MIN_TILES silently changed the split policy):
//   g_split_mode: 0 = the policy below, 1 = never split-K, 2 = split-K wherever the epilogue allows it;
//   g_take_all:   1 = every launch the kernel CAN run takes this engine (no tile bound, no short-K guard).
static int g_split_mode = 0;
static int g_take_all = 0;

// Test / tuning hook: set the switch and the tile bound (-1 keeps a value); returns the previous state as
// enabled | min_tiles << 1.  The tile bound is only a bound.
extern "C" int styler_gemm256_config(int enabled, int min_tiles) {
  const int prev = (g_enabled ? 1 : 0) | (g_min_tiles << 1);
  if (enabled >= 0) g_enabled = enabled;
  if (min_tiles >= 0) g_min_tiles = min_tiles;
  return prev;
}
// split_mode / take_all as above (-1 keeps a value); returns the previous pair as split_mode | take_all << 2.
extern "C" int styler_gemm256_policy(int split_mode, int take_all) {
  const int prev = g_split_mode | (g_take_all << 2);
  if (split_mode >= 0 && split_mode <= 2) g_split_mode = split_mode;
  if (take_all >= 0) g_take_all = take_all ? 1 : 0;
  return prev;
}

// Round 6: the tile HEIGHT is a dispatch decision too (HT = 4: 256 rows, HT = 3: 192 rows, see the kernel).  The 192-row
// tile takes launches the 256-row bound turns away when they (a) have at least `g_min_tiles3` tiles of 192 x 256 and (b) fill
// the rounds they occupy to >= 80 % (442 tiles = 1.73 rounds: 0.86; 221 tiles: 0.86 of one round), and launches of the 256-row
// tile when rounds x (height + fixed cost) comes out lower.  STYLER_GEMM256_HT = 3 / 4 forces a height (0: this policy),
// STYLER_GEMM256_MIN_TILES3 = 0 switches the 192-row tile off.
static int g_ht = [] { const char* e = getenv("STYLER_GEMM256_HT"); return e ? atoi(e) : 0; }();
static int g_min_tiles3 = [] { const char* e = getenv("STYLER_GEMM256_MIN_TILES3"); return e ? atoi(e) : 300; }();
extern "C" int styler_gemm256_height(int ht, int min_tiles3) {       // test / tuning hook; returns the previous ht | min_tiles3 << 3
  const int prev = g_ht | (g_min_tiles3 << 3);
  if (ht == 0 || ht == 3 || ht == 4) g_ht = ht;
  if (min_tiles3 >= 0) g_min_tiles3 = min_tiles3;
  return prev;
}

static bool gemm256_eligible(int B, int L, int cin, int n, int kw, int64_t ldx, int x16, bool packed, int* mt_out, int* nt_out,
                             int* ht_out = nullptr) {
  const int enabled = g_enabled, min_tiles = g_take_all ? 1 : (g_min_tiles > 1 ? g_min_tiles : 1);
  if (!enabled || !x16) return false;
  if ((cin % BK) || (ldx & 7) || (n & 3) || kw > 9) return false;
  const int64_t M = (int64_t)B * L;
  const int nt = (n + BN - 1) / BN;
  if (!g_take_all && (cin / BK) * kw < 8) return false;             // short K: prologue + epilogue dominate a 1-block-per-CU tile
  if ((n % BN) > 0 && (n % BN) < 192) return false;                // a mostly empty last column tile wastes its MFMAs
  if (((M + 8) * ldx * 2) >= ((int64_t)1 << 31) || ((int64_t)n * kw * cin * 2) >= ((int64_t)1 << 31)) return false;
  // packed decoder rows: M is the row CAPACITY (B * T); the valid prefix is known on the device only and is ~64 % of it at
  // VCTK shapes -- tiles behind the data exit at once, so the bound is applied to 60 % of the m-tiles
  auto tiles_of = [&](int h, int* mt) {
    *mt = (int)((M + h - 1) / h);
    return (packed ? ((int64_t)*mt * 3 + 4) / 5 : (int64_t)*mt) * nt;
  };
  int mt4, mt3;
  const int64_t t4 = tiles_of(256, &mt4), t3 = tiles_of(192, &mt3);
  const bool ok4 = t4 >= min_tiles && g_ht != 3;
  const int64_t r3 = (t3 + 255) / 256;
  const bool ok3 = g_ht != 4 && (g_ht == 3 ? t3 >= min_tiles || g_take_all
                                           : g_min_tiles3 > 0 && t3 >= g_min_tiles3 && t3 * 5 >= r3 * 256 * 4);
  if (!ok4 && !ok3) return false;
  int ht = ok4 ? 4 : 3;
  if (ok4 && ok3 && !g_ht && r3 * 7 < ((t4 + 255) / 256) * 9) ht = 3;      // rounds x (height + 0.5)
  *mt_out = ht == 4 ? mt4 : mt3; *nt_out = nt;
  if (ht_out) *ht_out = ht;
  return true;
}

// Split-K = 2: launches whose data-carrying tiles are fewer than the CUs while K is long (the FFN's k = 9 dX at M = 27 060:
// 106 tiles x 144 K steps) run as two half-K launches in ONE grid (212 blocks) writing fp32 partial tiles, plus the combine
// pass.  Only plain epilogues (no activation, no ReLU mask, n == 256-multiple rows of float4), and only when the caller
// handed over a workspace for THIS call (styler_gemm_set_workspace).
static thread_local void* t_ws = nullptr;
static thread_local int64_t t_ws_bytes = 0;

// The next styler_conv_gemm / styler_conv_gemm_packed call of this host thread may use [ptr, ptr + bytes) as scratch (fp32
// split-K partial tiles); the registration is consumed by that call.  styler_conv_gemm_workspace_bytes tells how much a
// call wants (0: none).  The memory must stay valid until the call's kernels have run (stream order).
extern "C" int styler_gemm_set_workspace(void* ptr, int64_t bytes) {
  t_ws = ptr; t_ws_bytes = ptr ? bytes : 0;
  return 0;
}

static int gemm256_ksplit(int B, int L, int cin, int n, int kw, int64_t ldx, int x16, bool packed, int act, bool has_mask) {
  static const int split_on = [] { const char* e = getenv("STYLER_GEMM256_SPLITK"); return e ? atoi(e) : 1; }();
  if (!g_enabled || !split_on || !x16 || g_split_mode == 1) return 1;
  if ((cin % BK) || (ldx & 7) || (n % BN) || kw > 9 || (act & 0xff) != STYLER_ACT_NONE || (act & STYLER_ACT_RES_FIRST) || has_mask) return 1;
  const int64_t M = (int64_t)B * L;
  const int64_t mt = (M + BM - 1) / BM, nt = n / BN;
  const int64_t tiles = (packed ? (mt * 3 + 4) / 5 : mt) * nt;
  if (g_split_mode != 2 && (tiles < 96 || tiles > 160 || (cin / BK) * kw < 64)) return 1;
  if (((M + 8) * ldx * 2) >= ((int64_t)1 << 31) || ((int64_t)n * kw * cin * 2) >= ((int64_t)1 << 31)) return 1;
  return 2;
}

extern "C" int64_t styler_conv_gemm_workspace_bytes(int B, int L, int cin, int n, int kw, int act, int prec, int io_flags,
                                                    int64_t ldx, int packed, int has_mask) {
  if (prec != STYLER_PREC_BF16) return 0;
  int ks = gemm256_ksplit(B, L, cin, n, kw, ldx, io_flags & STYLER_IO_X_BF16, packed != 0, act, has_mask != 0);
  if (ks <= 1) {
    int mt, nt;                                    // (the 64 x 64 split-K only where the 256 x 256 engine does not take the launch)
    if (gemm256_eligible(B, L, cin, n, kw, ldx, io_flags & STYLER_IO_X_BF16, packed != 0, &mt, &nt)) return 0;
    ks = styler_gemm_small_ksplit(B, L, cin, n, kw, act, has_mask != 0);
  }
  return ks > 1 ? (int64_t)ks * B * L * n * 4 : 0;
}

// Round 6: with zeroed int counters registered next to the workspace (2 per 256 x 256 tile of the launch) a split-K launch of this
// engine is finished inside the kernel (no combine pass); consumed together with the workspace.  The kernel leaves them zero.
static thread_local int* t_cnt = nullptr;
static thread_local int64_t t_cnt_n = 0;
static int g_fixup = [] { const char* e = getenv("STYLER_GEMM256_FIXUP"); return e ? atoi(e) : 1; }();
extern "C" int styler_gemm_set_counters(void* ptr, int64_t count) {
  t_cnt = reinterpret_cast<int*>(ptr); t_cnt_n = ptr ? count : 0;
  return 0;
}
extern "C" int styler_gemm256_fixup(int enabled) {                   // test / A-B hook; returns the previous value
  const int prev = g_fixup;
  if (enabled >= 0) g_fixup = enabled ? 1 : 0;
  return prev;
}

void styler_gemm_take_workspace(void** ws, int64_t* bytes) {
  *ws = t_ws; *bytes = t_ws_bytes;
  t_ws = nullptr; t_ws_bytes = 0;
}

int styler_gemm_combine(const GemmArgs& a0, int ks, int y16, hipStream_t st) {
  const int64_t M = (int64_t)a0.B * a0.L;
  const int64_t quads = M * (a0.n >> 2);
  const unsigned cb = (unsigned)((quads + 255) / 256 > 4096 ? 4096 : (quads + 255) / 256);
  const int64_t* nrows = (a0.rowinfo && a0.B == 1) ? a0.len : nullptr;
  if (y16) hipLaunchKernelGGL(gemm256_combine_kernel<true>, dim3(cb), dim3(256), 0, st, a0.part, M, a0.n, ks, a0.scale, a0.shift, a0.res, a0.ldres, a0.y, a0.ldy, nrows, a0.res16, nullptr, 0);
  else hipLaunchKernelGGL(gemm256_combine_kernel<false>, dim3(cb), dim3(256), 0, st, a0.part, M, a0.n, ks, a0.scale, a0.shift, a0.res, a0.ldres, a0.y, a0.ldy, nrows, a0.res16, a0.y3, a0.y3parts);
  return launch_status();
}

int styler_gemm256_try(const GemmArgs& a0, int x16, int y16, hipStream_t st, void* ws, int64_t ws_bytes) {
  int* const cnt = t_cnt; const int64_t cnt_n = t_cnt_n;          // consumed by this call, whatever path it takes
  t_cnt = nullptr; t_cnt_n = 0;
  if (a0.trace) return 0;
  int mt, nt;
  const int64_t M = (int64_t)a0.B * a0.L;
  const int ks = gemm256_ksplit(a0.B, a0.L, a0.cin, a0.n, a0.kw, a0.ldx, x16, a0.rowinfo != nullptr, a0.act, a0.mask != nullptr);
  if (ks > 1 && ws && ws_bytes >= (int64_t)ks * M * a0.n * 4 && !((uintptr_t)ws & 15) && !(a0.ldy & 3) &&
      (!a0.len || (a0.rowinfo && a0.B == 1))) {                        // (length zeroing other than the packed row counter: not in the combine pass)
    GemmArgs a = a0;
    a.mt = (int)((M + BM - 1) / BM); a.nt = a0.n / BN;
    a.ksplit = ks; a.part = reinterpret_cast<float*>(ws);
    const dim3 grid((unsigned)(((a.mt + 7) / 8) * 8 * a.nt), (unsigned)ks);
    if (g_fixup && ks == 2 && cnt && cnt_n >= 2 * (int64_t)a.mt * a.nt && !((uintptr_t)cnt & 3) &&
        (int64_t)a.mt * a.nt * BM * BN * 4 <= ws_bytes) {          // finished in the kernel: one raw accumulator image per tile
      a.cnt = cnt;
      if (y16) hipLaunchKernelGGL((conv_gemm256_kernel<true, 4, true>), grid, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((conv_gemm256_kernel<false, 4, true>), grid, dim3(512), 0, st, a);
      const int rc = launch_status();
      return rc ? (rc < 0 ? rc : -rc) : 1;
    }
    hipLaunchKernelGGL((conv_gemm256_kernel<false, 4>), grid, dim3(512), 0, st, a);
    int rc = launch_status();
    if (!rc) rc = styler_gemm_combine(a, ks, y16, st);
    return rc ? (rc < 0 ? rc : -rc) : 1;
  }
  int ht = 4;
  if (!gemm256_eligible(a0.B, a0.L, a0.cin, a0.n, a0.kw, a0.ldx, x16, a0.rowinfo != nullptr, &mt, &nt, &ht)) return 0;
  GemmArgs a = a0;
  a.mt = mt; a.nt = nt;
  const dim3 grid((unsigned)(((mt + 7) / 8) * 8 * nt));
  if (ht == 3) {
    if (y16) hipLaunchKernelGGL((conv_gemm256_kernel<true, 3>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_gemm256_kernel<false, 3>), grid, dim3(512), 0, st, a);
  } else {
    if (y16) hipLaunchKernelGGL((conv_gemm256_kernel<true, 4>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_gemm256_kernel<false, 4>), grid, dim3(512), 0, st, a);
  }
  const int rc = launch_status();
  return rc ? (rc < 0 ? rc : -rc) : 1;
}

// Which engine / tile a styler_conv_gemm call with these arguments runs on: 0..3 = styler_conv_gemm_variant (bit 0: 128 x 128
// tile, bit 1: bf16 MFMA), 4 = the 256 x 256 LDS-DMA engine of this file.
extern "C" int styler_conv_gemm_engine2(int B, int L, int cin, int n, int kw, int prec, int io_flags, int64_t ldx, int packed,
                                        int act, int has_mask) {
  int mt, nt;
  if (prec == STYLER_PREC_BF16) {
    // a split-K launch runs on this engine too (it bypasses the tile bound); it needs the caller's workspace, which the
    // host layer supplies whenever styler_conv_gemm_workspace_bytes asks for one
    if (gemm256_ksplit(B, L, cin, n, kw, ldx, io_flags & STYLER_IO_X_BF16, packed != 0, act, has_mask != 0) > 1) return 4;
    if (gemm256_eligible(B, L, cin, n, kw, ldx, io_flags & STYLER_IO_X_BF16, packed != 0, &mt, &nt)) return 4;
  }
  return styler_conv_gemm_variant(B, L, cin, n, kw, prec);
}
extern "C" int styler_conv_gemm_engine(int B, int L, int cin, int n, int kw, int prec, int io_flags, int64_t ldx, int packed) {
  return styler_conv_gemm_engine2(B, L, cin, n, kw, prec, io_flags, ldx, packed, STYLER_ACT_NONE, 0);
}
