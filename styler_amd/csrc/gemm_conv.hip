// GEMM / Conv1d-as-implicit-GEMM on MFMA (gfx950).
//
//   y[m, n] = act(scale[n] * sum_{j<kw, c<cin} x[row(m) + j - kw/2, c] * w[n, j*cin + c] + shift[n]) (+ res)
//
// m = b*L + t indexes the padded rectangle [B, L]; rows shifted outside [0, L) of their own item
// contribute zero ('same' zero padding).  No im2col buffer exists, not even in LDS: for each channel
// chunk the block stages ONE haloed activation tile [(BM + kw - 1) rows x BK channels] and walks the kw
// taps over it by shifting the LDS row index, so an activation element is fetched from L2 once per
// chunk instead of kw times (and converted to bf16 once).  Only the weight tile [BN x BK] streams per
// (chunk, tap) step; it is double-buffered in LDS with its global loads issued one step ahead, so a
// step costs one barrier.
//
// Tile engine: 256 threads = 4 waves as 2x2, each wave owns TM x TN MFMA tiles of 32x32, block tile
// (64*TM) x (64*TN).  Two arithmetic modes share the skeleton:
//   * F32 : v_mfma_f32_32x32x2_f32, BK = 32 floats, exact fp32 (== an fmaf chain).
//   * BF16: v_mfma_f32_32x32x16_bf16, BK = 64; activations are converted fp32->bf16 (v_cvt_pk_bf16_f32)
//           while being staged into LDS, weights come from a bf16 shadow copy.
// K is permuted inside a chunk so that each lane's fragment is CONTIGUOUS in LDS (the k <-> lane map
// only has to agree between A and B): lane (i = l&31, h = l>>5) reads 16 floats (F32) or 8 bf16 per
// MFMA step with ds_read_b128.  LDS rows are padded to 36 dwords: 16 consecutive rows then hit 16
// distinct 16-byte slots of the 64-bank row (conflict-free ds_read_b128), for any tap shift.
//
// Block -> tile map is XCD-aware: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only); m-tiles are
// dealt round-robin to the XCDs with the n index fastest inside an XCD: the n-tiles that share an activation tile run
// on the same L2.
#include <cstdlib>
#include <type_traits>
#include "common.h"

#include "gemm_args.h"

// Round 4: in the occupancy-3 layout (unpadded 128-byte weight rows) the weight tile of a K step goes L2 -> LDS by DMA
// (buffer_load_dwordx4 ... lds) instead of through registers: no staging registers, no ds_write pass; the XOR swizzle of the
// 16-byte slots is applied to the SOURCE address (the DMA writes lane-linear).  -DSTYLER_GEMM_B_DMA=0: register staging.
#ifndef STYLER_GEMM_B_DMA
#define STYLER_GEMM_B_DMA 1
#endif
typedef __attribute__((address_space(3))) void cg_lds_void;

// Phase timestamps (constant 100 MHz counter, s_memrealtime) of every block of the launches that follow
// styler_gemm_set_trace(buf): [block, entry, first tile staged, main loop done, stores issued, stores acknowledged,
// hardware id, tile].  A measurement hook (tools/gemm_trace.py), off (null) by default.
static uint64_t* g_gemm_trace = nullptr;
extern "C" int styler_gemm_set_trace(void* buf) { g_gemm_trace = reinterpret_cast<uint64_t*>(buf); return 0; }

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  return cvt_pk_bf16_rne(lo, hi);
}

// A16 / Y16: the activation operand / the output live in HBM as bf16 (the FFN hidden tensor and its gradient in throughput
// mode: both are only ever consumed as bf16 MFMA operands or as a sign mask, so storing them rounded changes no result and
// halves the bytes of the widest tensors of the model).
//
// WM: waves along M.  2 (default): the 2 x 2 wave grid above.  4: a 4 x 1 grid, block tile (128*TM) x (32*TN) -- the
// narrow-output variant <1, 3, ..., WM = 4> = 128 x 96 for 64 < n <= 96 (the 80 mel channels: PostNet's last convolution,
// the dX of its first, mel_linear).  On the 64 x 64 tile those launches run two n-tiles, the second with 16 of 64 columns
// in use, and every activation row is staged twice; one 96-wide tile covers the row once (12 MFMAs per wave and step
// instead of 4).  Measured (M = 42 336, K = 2560, bf16 in): 52.3 -> 40.3 us alone, training step -0.04 ms; at 21 168 rows
// (166 blocks) the two tiles tie, so the variant takes launches of >= 32 768 rows.
// (the body is a device function: conv_gemm_kernel runs one problem per launch, conv_gemm_group_kernel up to eight small ones;
// TAG only keeps the two kernels' body specialisations distinct for hipcc's host pass)
template <int TM, int TN, bool BF16, bool KW1, bool OCC3, bool A16 = false, bool Y16 = false, int WM = 2, int TAG = 0>
__device__ __forceinline__ void conv_gemm_body(const GemmArgs& a, const int bid_in) {
  static_assert(BF16 || (!A16 && !Y16), "bf16 storage only with the bf16 MFMA");
  static_assert(WM == 2 || (WM == 4 && !OCC3), "wave grids: 2 x 2, or 4 x 1 (double-buffered layout only)");
  constexpr int WN = 4 / WM;
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int BK = BF16 ? 64 : 32;
  constexpr int LD = 36;                           // dwords per LDS row (32 data + 4 pad)
  constexpr int A_ROWS = BM + (KW1 ? 0 : 8);       // halo for kw <= 9
  constexpr int A_EPV = A16 ? 8 : 4;               // elements per 16-byte vector of the A operand in HBM
  constexpr int A_ES = A16 ? 2 : 4;                // bytes per element
  constexpr int A_V = BK / A_EPV;                  // 16-byte vectors per A row
  constexpr int A_RPP = 256 / A_V;                 // rows per pass
  constexpr int A_P = (A_ROWS + A_RPP - 1) / A_RPP;
  constexpr int B_V = 8;                           // 16-byte vectors per B row
  constexpr int B_RPP = 256 / B_V;
  constexpr int B_P = BN / B_RPP;
  constexpr int B_ES = BF16 ? 2 : 4;               // bytes per weight element
  constexpr int CLD = 32 * TN + 4;                 // epilogue staging row stride (floats)
  // OCC3 (occupancy-3 layout, 52.6 KB for the 128x128 bf16 tile => 3 blocks per CU): ONE activation buffer (re-staged
  // behind an extra barrier at chunk boundaries), weight rows unpadded (32 dwords) with the 16-byte slot index
  // XOR-swizzled by (row & 7) ^ ((row >> 3) & 3) -- ds_read_b128 is served in 16-lane groups that pair rows r and r + 8
  // (lanes 12-15 with 20-27 ...): with (row & 7) alone those pairs collided, 25 % of the LDS cycles of the k >= 5 convs --
  // epilogue staged in two halves.
  constexpr int NABUF = OCC3 ? 1 : 2;
  constexpr int LDB = OCC3 ? 32 : LD;
  constexpr int EPI_H = OCC3 ? 2 : 1;              // epilogue passes
  constexpr int SMEM_MAIN = NABUF * A_ROWS * LD + 2 * BN * LDB + LD;   // + one row of zeros
  constexpr int SMEM_EPI = 4 * 32 * TM * CLD / EPI_H;
  constexpr int SMEM = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;

  __shared__ __attribute__((aligned(16))) uint32_t smem[SMEM];
  uint32_t* const sA = smem;                       // NABUF x [A_ROWS][LD]
  uint32_t* const sB = smem + NABUF * A_ROWS * LD; // 2 x [BN][LDB]
  uint32_t* const sZ = sB + 2 * BN * LDB;          // [LD] zeros

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  uint64_t stamp[5];
  if (a.trace) stamp[0] = wall_clock64();

  // ---- XCD-aware tile assignment: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only).  M-tiles are
  // dealt round-robin to the XCDs and an XCD walks the n-tiles of its m-tile back to back, so the n-tiles that share an
  // activation tile run on the same L2 -- and the tiles that carry work are spread over all eight XCDs when only a
  // prefix of the rows is valid (packed decoder rows, masked tails).  The grid is padded to 8 * ceil(mt / 8) m-tiles.
  int tile;
  {
    const int bid = bid_in;
    const int xcd = bid & 7, k = bid >> 3;
    const int mtile = (k / a.nt) * 8 + xcd;
    if (mtile >= a.mt) return;
    tile = mtile * a.nt + k % a.nt;
  }
  const int64_t M = (int64_t)a.B * a.L;
  const int64_t m0 = (int64_t)(tile / a.nt) * BM;
  const int n0 = (tile % a.nt) * BN;
  const int kw = KW1 ? 1 : a.kw;
  const int pad = KW1 ? 0 : a.pad;
  const int ktot = kw * a.cin;
  // channel chunks; a split-K launch (a.ksplit > 1, conv_gemm_kernel below) gives blockIdx.y its share of them
  const int ncc_all = (a.cin + BK - 1) / BK;
  int cc0 = 0, cc1 = ncc_all;
  if (a.ksplit > 1) {
    const int per = (ncc_all + a.ksplit - 1) / a.ksplit;
    cc0 = (int)blockIdx.y * per;
    cc1 = cc0 + per < ncc_all ? cc0 + per : ncc_all;
    cc0 = cc0 < cc1 ? cc0 : cc1;
  }
  const int nsteps = (cc1 - cc0) * kw;

  // ---- tiles made only of rows at or past their item's length produce zeros: write them and leave (in the packed
  // decoder layout B = 1 and len[0] = the number of packed rows: every tile behind the data is skipped) ----
  if (a.len) {
    // (32-bit arithmetic: the host side rejects B * L >= 2^31)
    const uint32_t span = (uint32_t)((m0 + BM < M ? m0 + BM : M) - 1 - m0);
    const uint32_t b0 = (uint32_t)m0 / (uint32_t)a.L, t0 = (uint32_t)m0 - b0 * (uint32_t)a.L;
    if (t0 + span < (uint32_t)a.L && (int64_t)t0 >= a.len[b0]) {
      // packed rows: nothing behind the data is ever read (every consumer is bounded by the same row counter), so the
      // tile is not even zero-filled -- that fill was 36 % of the output bytes of every decoder GEMM at VCTK shapes
      if (a.rowinfo) return;
      constexpr int QPR = BN / 4;                    // float4 per tile row
      for (int i = tid; i < BM * QPR; i += 256) {
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

  // ---- operand fetch = raw buffer loads: (uniform descriptor) + (32-bit lane offset) ----
  // The descriptor of a load starts at the tile's first row inside the tensor and ends with the tensor (num_records):
  // rows past the end read as zeros in hardware; halo rows before row 0 get a negative offset, which wraps far above
  // num_records; channels past cin (partial last chunk) swap in the OOB marker.  No masks, no branches, no zero-fill
  // moves: one v_add per load.
  constexpr uint32_t OOB = 0x80000000u;
  constexpr int64_t REC_MAX = (int64_t)1 << 30;      // tile-relative offsets are < 5 MB; markers and wraps are > 2^30
  const int a_col = (tid % A_V) * A_EPV;
  const int a_r0 = tid / A_V;
  const int64_t mrow0 = m0 - pad;                    // first halo row of the tile (< 0 for the first tile)
  const int64_t mbase = mrow0 > 0 ? mrow0 : 0;
  const uint32_t va0 = (uint32_t)(((a_r0 + (int)(mrow0 - mbase)) * (int)a.ldx + a_col) * A_ES);
  const uint32_t a_pstep = (uint32_t)(A_RPP * (int)a.ldx * A_ES);

  const int b_col = (tid % B_V) * (BF16 ? 8 : 4);  // element offset inside the chunk
  const int b_r0 = tid / B_V;
  const uint32_t vb0 = (uint32_t)((b_r0 * ktot + b_col) * B_ES);
  const uint32_t b_pstep = (uint32_t)(B_RPP * ktot * B_ES);

  // LDS byte offsets (per lane), everything else in the fragment addresses is uniform or immediate
  const uint32_t fa_off = ((wm * TM * 32 + li) * LD + lh * (BF16 ? 4 : 16));
  const uint32_t fb_off = OCC3 ? (wn * TN * 32 + li) * LDB : ((wn * TN * 32 + li) * LD + lh * (BF16 ? 4 : 16));
  uint32_t fb_sw[4];                               // OCC3: swizzled dword offset of MFMA step s inside the row
#pragma unroll
  for (int sx = 0; sx < 4; ++sx) fb_sw[sx] = (uint32_t)(((sx * 2 + lh) ^ ((li & 7) ^ ((li >> 3) & 3))) * 4);
  const uint32_t sa_off = a_r0 * LD + (BF16 ? a_col / 2 : a_col);     // dwords (a bf16 pair per dword)
  const uint32_t sb_off = OCC3 ? b_r0 * LDB + (((tid % B_V) ^ ((b_r0 & 7) ^ ((b_r0 >> 3) & 3))) * 4) : b_r0 * LD + (tid % B_V) * 4;
  // weight tile by LDS-DMA (occupancy-3 layout): piece P = wave + 4 q covers tile rows 8 P .. 8 P + 7 (1 KB of LDS); lane l
  // lands at row 8 P + (l >> 3), physical 16-byte slot l & 7, which holds the LOGICAL chunk (l & 7) ^ swz(row)
  constexpr bool BDMA = OCC3 && BF16 && (STYLER_GEMM_B_DMA != 0);
  static_assert(!BDMA || BN == 128, "B-tile DMA: 16 pieces of 8 rows");
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t vbd[4];                                 // byte offset of (row, logical chunk) inside the (n0, tap 0, chunk 0) tile
  int lc8[4];                                      // first channel of the lane's logical chunk inside the 64-channel chunk
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 8 * (wave_u + 4 * q) + (lane >> 3);
    const int lc = (lane & 7) ^ ((row & 7) ^ ((row >> 3) & 3));
    lc8[q] = lc * 8;
    vbd[q] = (uint32_t)((row * ktot + lc * 8) * 2);
  }

  // per-lane tap validity for the wave's output rows: bit j set <=> row t + j - pad lies in [0, L)
  uint32_t tapmask[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t m = m0 + (wm * TM + i) * 32 + li;
    uint32_t bits = 0;
    if (!KW1 && m < M) {
      int t, rem;                                    // rows before / after this one inside its item
      if (a.rowinfo) { const int2 ri = a.rowinfo[m]; t = ri.x; rem = ri.y; }
      else { t = (int)((uint32_t)m % (uint32_t)a.L); rem = a.L - 1 - t; }
      for (int j = 0; j < kw; ++j) {
        const int o = j - pad;
        if (o >= -t && o <= rem) bits |= 1u << j;
      }
    }
    tapmask[i] = bits;
  }

  i32x4 ra[A_P];                                   // 16 raw bytes per pass: 4 floats, or 8 bf16 (A16)
  uint4 rb[B_P];

  auto load_a = [&](int cc) {
    const int c0 = cc * BK;
    // (compact bf16x3 activations, a.x3n1: the row holds [hi | lo]; the chunks of the third product read hi again)
    const int cs = (a.x3n1 && cc >= 2 * a.x3n1) ? c0 - 2 * a.x3n1 * BK : c0;
    const int ccols = a.x3n1 ? 2 * a.x3n1 * BK : a.cin;                // columns a row really has
    int64_t rec = ((M - mbase - 1) * a.ldx + (ccols - cs)) * A_ES;
    rec = rec > REC_MAX ? REC_MAX : rec;
    const char* abase = reinterpret_cast<const char*>(a.x) + (mbase * a.ldx + cs) * A_ES;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase), 0, (int)rec, 0x00020000);
    const uint32_t v = c0 + a_col < a.cin ? va0 : OOB;
#pragma unroll
    for (int p = 0; p < A_P; ++p) {
      ra[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v + p * a_pstep, 0, 0);
    }
  };
  auto store_a = [&](int buf) {
    uint32_t* dst = sA + (OCC3 ? 0 : buf) * A_ROWS * LD + sa_off;
#pragma unroll
    for (int p = 0; p < A_P; ++p) {
      if (a_r0 + p * A_RPP < A_ROWS) {
        if (A16) {
          *reinterpret_cast<i32x4*>(&dst[p * A_RPP * LD]) = ra[p];      // already bf16: no conversion
        } else if (BF16) {
          const float4 f = *reinterpret_cast<const float4*>(&ra[p]);
          *reinterpret_cast<uint2*>(&dst[p * A_RPP * LD]) = make_uint2(cvt_pk_bf16(f.x, f.y), cvt_pk_bf16(f.z, f.w));
        } else {
          *reinterpret_cast<i32x4*>(&dst[p * A_RPP * LD]) = ra[p];
        }
      }
    }
  };
  auto load_b = [&](int cc, int j) {
    const int c0 = cc * BK;
    int64_t rec = ((int64_t)(a.n - n0 - 1) * ktot + (a.cin - c0)) * B_ES;
    rec = rec > REC_MAX ? REC_MAX : rec;
    const char* base = reinterpret_cast<const char*>(a.w) + ((int64_t)n0 * ktot + (int64_t)j * a.cin + c0) * B_ES;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)rec, 0x00020000);
    const uint32_t v = c0 + b_col < a.cin ? vb0 : OOB;
#pragma unroll
    for (int p = 0; p < B_P; ++p) {
      const i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v + p * b_pstep, 0, 0);
      rb[p] = *reinterpret_cast<const uint4*>(&t);
    }
  };
  auto store_b = [&](int buf) {
    uint32_t* dst = sB + buf * BN * LDB + sb_off;      // B_RPP = 32 rows per pass: (row & 7) is pass-invariant
#pragma unroll
    for (int p = 0; p < B_P; ++p) *reinterpret_cast<uint4*>(&dst[p * B_RPP * LDB]) = rb[p];
  };
  auto dma_b = [&](int cc, int j, int buf) {         // the whole [BN x 64] weight tile of step (cc, j) -> sB[buf]
    const int c0 = cc * BK;
    int64_t rec = ((int64_t)(a.n - n0 - 1) * ktot + (a.cin - c0)) * B_ES;
    rec = rec > REC_MAX ? REC_MAX : rec;
    const char* base = reinterpret_cast<const char*>(a.w) + ((int64_t)n0 * ktot + (int64_t)j * a.cin + c0) * B_ES;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)rec, 0x00020000);
    char* dst = reinterpret_cast<char*>(sB + buf * BN * LDB) + wave_u * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (cg_lds_void*)(dst + q * 4096), 16, c0 + lc8[q] < a.cin ? vbd[q] : OOB, 0, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (tid < LD) sZ[tid] = 0u;
  const uint32_t* const zrow = sZ + lh * (BF16 ? 4 : 16);
  load_a(cc0);
  if constexpr (BDMA) dma_b(cc0, 0, 0); else load_b(cc0, 0);
  store_a(cc0 & 1);
  if constexpr (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else store_b(0);
  __syncthreads();
  if (a.trace) stamp[1] = wall_clock64();

#ifdef STYLER_GEMM_TRACE_STEPS                          // diagnosis builds (-DSTYLER_GEMM_TRACE_STEPS, loaded through STYLER_LIB):
  uint64_t ph[4] = {0, 0, 0, 0};                       // shader-clock cycles wave 0 spends in each phase of the main loop
#define PH_MARK(i, t0) { const uint64_t t1_ = __builtin_readcyclecounter(); ph[i] += t1_ - t0; t0 = t1_; }
#else
#define PH_MARK(i, t0)
#endif
  int cc = cc0, j = 0;
  for (int step = 0; step < nsteps; ++step) {
    // next step's coordinates; its global loads are issued now and land in LDS after this step's MFMAs
    int ccn = cc, jn = j + 1;
    if (jn == kw) { jn = 0; ccn = cc + 1; }
    const bool more = step + 1 < nsteps;
#ifdef STYLER_GEMM_TRACE_STEPS
    uint64_t tp = __builtin_readcyclecounter();
#endif
    if (more) {
      // (DMA: straight into the other weight buffer -- its last reader was step - 1, behind the barrier that ended it)
      if constexpr (BDMA) dma_b(ccn, jn, (step + 1) & 1); else load_b(ccn, jn);
      if (jn == 0) load_a(ccn);
    }
    PH_MARK(0, tp)

    const uint32_t* cA = sA + (OCC3 ? 0 : (cc & 1)) * A_ROWS * LD + j * LD + fa_off;
    const uint32_t* cB = sB + (step & 1) * BN * LDB + fb_off;
    // 'same' zero padding across item boundaries: a lane whose row t + j - pad falls outside [0, L)
    // reads its A fragment from a row of zeros instead (one address select per tile, no data selects)
    const uint32_t* pa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      pa[i] = cA + i * 32 * LD;
      if (!KW1) pa[i] = ((tapmask[i] >> j) & 1u) ? pa[i] : zrow;
    }
    if (BF16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(&pa[i][s * 8]);
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
          fb[jj] = OCC3 ? *reinterpret_cast<const bf16x8*>(&cB[jj * 32 * LDB + fb_sw[s]])
                        : *reinterpret_cast<const bf16x8*>(&cB[jj * 32 * LD + s * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
      }
    } else {
      f32x4 fa[TM][4], fb[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) fa[i][v] = *reinterpret_cast<const f32x4*>(&pa[i][v * 4]);
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int v = 0; v < 4; ++v) fb[jj][v] = *reinterpret_cast<const f32x4*>(&cB[jj * 32 * LD + v * 4]);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk >> 2][kk & 3], fb[jj][kk >> 2][kk & 3],
                                                              acc[i][jj], 0, 0, 0);
    }

#ifdef STYLER_GEMM_TRACE_STEPS
    asm volatile("s_nop 0" ::: "memory");
#endif
    PH_MARK(1, tp)
    if (more) {
      if constexpr (!BDMA) store_b((step + 1) & 1);
      if (jn == 0) {
        if (OCC3) __syncthreads();                     // every wave is done with the single activation buffer
        store_a(ccn & 1);
      }
      if constexpr (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces have landed
    }
    PH_MARK(2, tp)
    __syncthreads();
    PH_MARK(3, tp)
    cc = ccn; j = jn;
  }

  if (a.trace) stamp[2] = wall_clock64();
  // ---- epilogue: accumulators -> per-wave LDS tile -> coalesced float4 rows ----
  // C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  //
  // Everything the tail needs from memory (residual, ReLU mask, item lengths) is fetched in batches of up to 8 rows per
  // lane BEFORE the rows are finished and stored: a load placed next to its row's store waits on vmcnt(0), which also waits
  // for the previous row's store to be acknowledged -- one memory round trip per row, 16 rows per lane, was most of the
  // life of a short-K block.  Rows and columns outside the tensor are dropped by the buffer descriptors (rows past M are
  // past num_records; lanes whose columns are past n carry an out-of-range offset), so no lane branches.  The activation
  // is a template argument of the tail (one uniform switch per block instead of a branch ladder per element).
  constexpr int LPR = 8 * TN;                      // lanes per output row (float4 each)
  constexpr int RPP = 64 / LPR;                    // rows per pass
  constexpr int ROWS_H = 32 * TM / EPI_H;          // tile rows staged per epilogue pass
  constexpr int NP = ROWS_H / RPP;                 // rows per lane per pass
  constexpr int UB = OCC3 ? 4 : (NP > 8 ? 8 : NP); // rows per batch (the occupancy-3 kernels must stay within 168 registers)
  constexpr int NPT = NP * EPI_H;                  // rows per lane per tile
  constexpr int Y_ES = Y16 ? 2 : 4;
  float* cst = reinterpret_cast<float*>(smem) + wave * (ROWS_H * CLD);
  const int lrow = lane / LPR;
  const int c4 = (lane % LPR) * 4;
  const int col = n0 + wn * 32 * TN + c4;
  const bool col_ok = col < a.n && lane < RPP * LPR;  // (8 * TN lanes per row: with TN = 3 the last 16 lanes carry no row)
  const int wrow0 = wm * 32 * TM + lrow;           // tile-relative row of this lane's first row
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok) {
    if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + col);
    if (a.shift) sf = *reinterpret_cast<const float4*>(a.shift + col);
  }
  const bool res_first = a.act & STYLER_ACT_RES_FIRST;
  const int actc = a.act & 0xff;

  // bit p of `live`: the lane's p-th row lies inside its item's length (rows at or past it are written as zeros)
  uint32_t live = 0xffffffffu;
  if (a.len) {
    live = 0u;
    if (a.B == 1) {                                  // packed rows / a single item: one uniform bound
      int64_t lim64 = a.len[0] - m0;
      const int lim = lim64 > BM ? BM : (lim64 < 0 ? 0 : (int)lim64);
#pragma unroll
      for (int p = 0; p < NPT; ++p) live |= (uint32_t)(wrow0 + p * RPP < lim) << p;
    } else {
      const uint32_t Lu = (uint32_t)a.L;
      const uint32_t magic = 0xffffffffu / Lu;       // floor(n / L) = umulhi(n, magic) or that + 1 (one fix-up)
      const uint32_t b0 = (uint32_t)m0 / Lu, t0 = (uint32_t)m0 - b0 * Lu;
      uint32_t rem[NPT];
      int lv[NPT];
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const uint32_t nn = t0 + (uint32_t)(wrow0 + p * RPP);
        uint32_t q = __umulhi(nn, magic), r = nn - q * Lu;
        if (r >= Lu) { ++q; r -= Lu; }
        uint32_t bi = b0 + q;
        bi = bi < (uint32_t)a.B ? bi : (uint32_t)a.B - 1u;        // rows past M are dropped by the store anyway
        rem[p] = r;
        lv[p] = reinterpret_cast<const int*>(a.len)[2 * bi];      // low dword of the int64 length
      }
#pragma unroll
      for (int p = 0; p < NPT; ++p) live |= (uint32_t)((int)rem[p] < lv[p]) << p;
    }
  }

  // descriptors relative to the tile's first row: rows at or past M start at or past num_records
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
    for (int hh = 0; hh < EPI_H; ++hh) {
      if (hh) __syncthreads();
      static_assert(EPI_H == 1 || TM == 2, "two-pass epilogue stages one 32-row MFMA tile row per pass");
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (EPI_H == 2 && i != hh) continue;
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cst[((EPI_H == 2 ? 0 : i * 32) + (r & 3) + 8 * (r >> 2) + 4 * lh) * CLD + jj * 32 + li] = acc[i][jj][r];
      }
      __syncthreads();
#pragma unroll
      for (int pb = 0; pb < NP / UB; ++pb) {
        const int p0 = hh * NP + pb * UB;            // index of the batch's first row among the lane's rows
        i32x4 rr[UB], mk[UB];
        if (a.res) {
          if (a.res16) {                             // bf16 residual: 8 bytes per lane, widened to fp32 in place
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r_rs, or0 + (p0 + u) * rstep, 0, 0);
              rr[u] = i32x4{(int)((uint32_t)t.x << 16), (int)((uint32_t)t.x & 0xffff0000u), (int)((uint32_t)t.y << 16),
                            (int)((uint32_t)t.y & 0xffff0000u)};
            }
          } else {
#pragma unroll
            for (int u = 0; u < UB; ++u) rr[u] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, or0 + (p0 + u) * rstep, 0, 0);
          }
        }
        if (a.mask) {
          if (a.mask16) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(m_rs, om0 + (p0 + u) * mstep, 0, 0);
              mk[u] = i32x4{t.x, t.y, 0, 0};
            }
          } else {
#pragma unroll
            for (int u = 0; u < UB; ++u) mk[u] = __builtin_amdgcn_raw_buffer_load_b128(m_rs, om0 + (p0 + u) * mstep, 0, 0);
          }
        }
        float4 v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) v[u] = *reinterpret_cast<const float4*>(&cst[((pb * UB + u) * RPP + lrow) * CLD + c4]);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          float4 w = v[u];
          if (res_first && a.res) {                   // partial sums of a multi-call convolution: joined before the tail
            const float4 q = *reinterpret_cast<const float4*>(&rr[u]);
            w.x += q.x; w.y += q.y; w.z += q.z; w.w += q.w;
          }
          w.x = apply_act(w.x * sc.x + sf.x, ACT); w.y = apply_act(w.y * sc.y + sf.y, ACT);
          w.z = apply_act(w.z * sc.z + sf.z, ACT); w.w = apply_act(w.w * sc.w + sf.w, ACT);
          if (a.mask) {                               // dX of a ReLU layer: gradient only where the forward output was > 0
            if (a.mask16) {                           // bf16 mask: positive <=> sign clear and magnitude non-zero
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
          if (!((live >> (p0 + u)) & 1u)) w = make_float4(0.f, 0.f, 0.f, 0.f);
          if (Y16) {
            const i32x2 o = {(int)cvt_pk_bf16(w.x, w.y), (int)cvt_pk_bf16(w.z, w.w)};
            __builtin_amdgcn_raw_buffer_store_b64(o, y_rs, oy0 + (p0 + u) * ystep, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const i32x4*>(&w), y_rs, oy0 + (p0 + u) * ystep, 0, 0);
            if (want3) {
              uint2 h3, l3;
              x3_split4(w, h3, l3);
              const i32x2 hv = {(int)h3.x, (int)h3.y}, lv = {(int)l3.x, (int)l3.y};
              const uint32_t o3 = o30 + (p0 + u) * step3;
              __builtin_amdgcn_raw_buffer_store_b64(hv, y3_rs, o3, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b64(lv, y3_rs, o3 + lo3, 0, 0);
              if (a.y3parts == 3) __builtin_amdgcn_raw_buffer_store_b64(hv, y3_rs, o3 + 2 * lo3, 0, 0);
            }
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
  if (a.trace) {
    stamp[3] = wall_clock64();
    __builtin_amdgcn_s_waitcnt(0);                   // every store of this wave acknowledged
    stamp[4] = wall_clock64();
    if (tid == 0) {
      uint64_t* t = a.trace + (int64_t)bid_in * 8;
      t[0] = bid_in;
#pragma unroll
      for (int i = 0; i < 5; ++i) t[1 + i] = stamp[i];
      t[6] = __builtin_amdgcn_s_getreg((3 << 11) | 4);   // HW_ID (wave, simd, cu, sh, se)
      t[7] = (uint64_t)tile;
#ifdef STYLER_GEMM_TRACE_STEPS                          // words 6, 7: (issue | ds_read + MFMA) and (wait + ds_write | barrier) cycles
      t[6] = (ph[0] << 32) | (ph[1] & 0xffffffffu);
      t[7] = (ph[2] << 32) | (ph[3] & 0xffffffffu);
#endif
    }
  }
}

template <int TM, int TN, bool BF16, bool KW1, bool OCC3, bool A16 = false, bool Y16 = false, int WM = 2>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmArgs a) {
  // split-K (small-M, long-K launches, gemm_small_ksplit below): blockIdx.y owns a range of channel chunks and stores its raw
  // fp32 accumulators as a plain [M][n] partial tile; styler_gemm_combine adds them in a fixed order and applies the epilogue
  if (!Y16 && a.ksplit > 1) {
    a.y = a.part + (int64_t)blockIdx.y * a.B * a.L * a.n;
    a.ldy = a.n;
    a.scale = nullptr; a.shift = nullptr; a.res = nullptr; a.mask = nullptr; a.act = STYLER_ACT_NONE;
    a.y3 = nullptr;                                  // (the split of the OUTPUT is written by the combine pass)
  }
  conv_gemm_body<TM, TN, BF16, KW1, OCC3, A16, Y16, WM, 0>(a, blockIdx.x);
}

// Up to 8 small GEMMs (64 x 64 tile, k = 1, fp32 activations in and out, bf16 MFMA) in ONE launch: problem p owns blocks
// [start[p], start[p + 1]).  The S-domain of a training step is a chain of such launches of 5-8 us each (the four BiLSTM
// input projections of a layer, the first / second Linear of the four style MLPs, the three classifier projections: the
// members of a group are independent of each other) -- launch latency, not arithmetic.
struct GemmGroup {
  GemmArgs a[8];
  int start[9];
  int n;
};
__global__ __launch_bounds__(256) void conv_gemm_group_kernel(const GemmGroup g) {
  int p = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) p += (i < g.n && (int)blockIdx.x >= g.start[i]) ? 1 : 0;
  conv_gemm_body<1, 1, true, true, false, false, false, 2, 1>(g.a[p], (int)blockIdx.x - g.start[p]);
}
// ... with bf16 activations in (StylerGemmProblem.flags bit 0): the grouped launches of the bf16x3 arithmetic, whose members
// read split activations (GemmArgs.x3n1 for the compact form)
__global__ __launch_bounds__(256) void conv_gemm_group16_kernel(const GemmGroup g) {
  int p = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) p += (i < g.n && (int)blockIdx.x >= g.start[i]) ? 1 : 0;
  conv_gemm_body<1, 1, true, true, false, true, false, 2, 2>(g.a[p], (int)blockIdx.x - g.start[p]);
}

template <int TM, int TN, bool BF16, int WM = 2>
static int launch_gemm(GemmArgs a, hipStream_t st, int x16, int y16) {
  const int64_t M = (int64_t)a.B * a.L;
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * (4 / WM);
  a.mt = (int)((M + BM - 1) / BM);
  a.nt = (a.n + BN - 1) / BN;
  const dim3 grid((unsigned)(((a.mt + 7) / 8) * 8 * a.nt), (unsigned)(a.ksplit > 1 ? a.ksplit : 1));
  // occupancy-3 layout (one extra barrier per chunk): measured +12..15 % on the k = 9 / k = 5 convs; on k = 3 it lost 5 %
  // with the round-1 epilogue and is a small gain with the present one (train step 11.989 -> 11.972 ms same-box).
  // STYLER_GEMM_OCC3=0/1 overrides for experiments.
  static const int occ3_env = [] { const char* e = getenv("STYLER_GEMM_OCC3"); return e ? atoi(e) : -1; }();
  const bool occ3 = occ3_env >= 0 ? occ3_env != 0 : a.kw >= 3;
#define GEMM_LAUNCH(KW1_, OCC_, A_, Y_) \
  hipLaunchKernelGGL((conv_gemm_kernel<TM, TN, BF16, KW1_, OCC_, A_, Y_, WM>), grid, dim3(256), 0, st, a)
#define GEMM_IO(KW1_, OCC_)                                                        \
  do {                                                                              \
    if constexpr (BF16) {                                                           \
      if (x16 && y16) GEMM_LAUNCH(KW1_, OCC_, true, true);                          \
      else if (x16) GEMM_LAUNCH(KW1_, OCC_, true, false);                           \
      else if (y16) GEMM_LAUNCH(KW1_, OCC_, false, true);                           \
      else GEMM_LAUNCH(KW1_, OCC_, false, false);                                   \
    } else {                                                                        \
      GEMM_LAUNCH(KW1_, OCC_, false, false);                                        \
    }                                                                               \
  } while (0)
  if ((x16 || y16) && !BF16) return STYLER_EINVAL;
  // kw = 1: the occupancy-3 layout (one activation buffer, hence an extra barrier per step) pays when the launch has more
  // blocks than two per CU can hold at once (the QKV projection: 34.3 -> 30.1 us); with fewer blocks the barrier only costs
  // (output projection, k = 1 FFN: +2..6 %).  STYLER_GEMM_OCC3_K1=0/1 overrides.
  static const int occ3_k1_env = [] { const char* e = getenv("STYLER_GEMM_OCC3_K1"); return e ? atoi(e) : -1; }();
  const bool occ3_k1 = occ3_k1_env >= 0 ? occ3_k1_env != 0 : (int64_t)a.mt * a.nt > 512;
  if (a.kw == 1 && BF16 && TM == 2 && WM == 2 && occ3_k1) GEMM_IO(true, (BF16 && TM == 2 && WM == 2));
  else if (a.kw == 1) GEMM_IO(true, false);
  else if (BF16 && TM == 2 && WM == 2 && occ3) GEMM_IO(false, (BF16 && TM == 2 && WM == 2));
  else GEMM_IO(false, false);
#undef GEMM_IO
#undef GEMM_LAUNCH
  return launch_status();
}

extern "C" int styler_conv_gemm_group(const StylerGemmProblem* probs, int count, void* stream) {
  if (!probs || count <= 0 || count > 8) return STYLER_EINVAL;
  GemmGroup g;
  int start = 0;
  const int x16 = probs[0].flags & 1;
  for (int k = 0; k < count; ++k) {
    const StylerGemmProblem& p = probs[k];
    if (!p.x || !p.w || !p.y || p.B <= 0 || p.L <= 0 || p.cin <= 0 || p.n <= 0) return STYLER_EINVAL;
    if ((p.flags & 1) != x16 || ((p.flags & 2) && (!x16 || (p.cin % 192) || p.ldx < 2 * (int64_t)(p.cin / 3)))) return STYLER_EINVAL;
    if ((p.cin & 7) || (p.ldx & (x16 ? 7 : 3)) || ((uintptr_t)p.x & 15) || ((uintptr_t)p.w & 15)) return STYLER_EALIGN;
    if ((p.n & 3) || (p.ldy & 3) || ((uintptr_t)p.y & 15) || (p.res && ((p.ldres & 3) || ((uintptr_t)p.res & 15)))) return STYLER_EALIGN;
    if ((int64_t)p.B * p.L >= ((int64_t)1 << 31) || p.ldy >= (1 << 22) || p.ldres >= (1 << 22)) return STYLER_EINVAL;
    GemmArgs a{p.x, p.ldx, p.w, reinterpret_cast<const float*>(p.scale), reinterpret_cast<const float*>(p.shift),
               reinterpret_cast<const float*>(p.res), p.ldres, p.y, p.ldy, p.B, p.L, p.cin, p.n, 1, p.act, 0,
               reinterpret_cast<const int64_t*>(p.len), 0, 0, nullptr, nullptr, 0, 0, nullptr};
    const int64_t M = (int64_t)p.B * p.L;
    a.mt = (int)((M + 63) / 64);
    a.nt = (p.n + 63) / 64;
    if (p.flags & 2) a.x3n1 = p.cin / 192;
    g.a[k] = a;
    g.start[k] = start;
    start += ((a.mt + 7) / 8) * 8 * a.nt;
  }
  for (int k = count; k < 9; ++k) g.start[k] = start;
  for (int k = count; k < 8; ++k) g.a[k] = g.a[0];
  g.n = count;
  if (x16) hipLaunchKernelGGL(conv_gemm_group16_kernel, dim3((unsigned)start), dim3(256), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(conv_gemm_group_kernel, dim3((unsigned)start), dim3(256), 0, (hipStream_t)stream, g);
  return launch_status();
}

// Tile choice: the 128x128 tile needs >= ~1 block per CU to pay; otherwise 64x64 (4x the blocks).
// Returns bit0 = 128x128 tile (else 64x64), bit1 = bf16 MFMA (else fp32 MFMA).
extern "C" int styler_conv_gemm_variant(int B, int L, int cin, int n, int kw, int prec) {
  (void)cin; (void)kw;
  const int64_t M = (int64_t)B * L;
  const int64_t big_blocks = ((M + 127) / 128) * ((n + 127) / 128);
  static const int nmin = [] { const char* e = getenv("STYLER_GEMM_BIG_NMIN"); return e ? atoi(e) : 96; }();
  int big = (big_blocks >= 192 && n >= nmin) ? 1 : 0;
  static const int force = [] { const char* e = getenv("STYLER_GEMM_TILE"); return e ? atoi(e) : 0; }();
  if (force == 1) big = 0;
  if (force == 2) big = 1;
  return big | (prec == STYLER_PREC_BF16 ? 2 : 0);
}

// The narrow-output tile (128 x 96, conv_gemm_kernel<1, 3, ..., WM = 4>): on by default for launches of >= 32768 rows
// (STYLER_GEMM_N96=0 turns it off).  styler_gemm_n96_config(enabled, min_rows): -1 keeps a value; returns the previous
// setting as enabled | (min_rows << 1) -- the A/B switch of the parity tests (small ragged cases on the new tile).
static int g_n96_enabled = [] { const char* e = getenv("STYLER_GEMM_N96"); return (!e || atoi(e) != 0) ? 1 : 0; }();
static int g_n96_min_rows = 32768;
extern "C" int styler_gemm_n96_config(int enabled, int min_rows) {
  const int prev = g_n96_enabled | (g_n96_min_rows << 1);
  if (enabled >= 0) g_n96_enabled = enabled ? 1 : 0;
  if (min_rows >= 0) g_n96_min_rows = min_rows;
  return prev;
}

// Split-K for the 64 x 64 tile: a conv GEMM over few rows and a long contraction axis (the text encoder's FFN k = 9 dX:
// M = 2880, n = 256, K = 9 x 1024 -> 180 tiles x 144 steps) has fewer blocks than CUs and nothing to hide the load -> LDS ->
// barrier latency of a step behind (0.73 us per step, 105 us per launch at 129 TFLOP/s).  Its channel chunks are dealt to
// blockIdx.y so that about four blocks share a CU; the partial tiles are folded by the combine pass of gemm256.hip.
// Plain epilogues only (bias / scale, residual), like the 256 x 256 engine's split-K.
static int g_small_split = [] { const char* e = getenv("STYLER_GEMM_SMALL_SPLITK"); return e ? atoi(e) : 1; }();
// enabled: 0 / 1, -1 keeps; returns the previous value (the A/B switch of the parity test)
extern "C" int styler_gemm_small_split_config(int enabled) {
  const int prev = g_small_split;
  if (enabled >= 0) g_small_split = enabled ? 1 : 0;
  return prev;
}
int styler_gemm_small_ksplit(int B, int L, int cin, int n, int kw, int act, bool has_mask) {
  if (!g_small_split || kw < 3 || (cin & 63) || (n & 3) || (act & 0xff) != STYLER_ACT_NONE || (act & STYLER_ACT_RES_FIRST) || has_mask) return 1;
  const int64_t M = (int64_t)B * L;
  const int64_t tiles = ((M + 63) / 64) * ((n + 63) / 64);
  const int ncc = cin / 64;
  if (tiles > 256 || ncc * kw < 48) return 1;
  const int want = (int)(1024 / tiles);                              // about four blocks per CU
  if (want < 2) return 1;
  const int per = ncc / want > 0 ? ncc / want : 1;
  const int ks = (ncc + per - 1) / per;
  return ks > 1 ? ks : 1;
}

// Internal entry with an explicit left padding (pad = kw/2 is the 'same' conv of the model; pad = 0 with an
// even kw is the framing conv of the STFT, stft.hip).
int styler_conv_gemm_impl2(const float* x, int64_t ldx, const void* w, const float* scale, const float* shift,
                           const float* res, int64_t ldres, float* y, int64_t ldy, int B, int L, int cin, int n,
                           int kw, int pad, int act, int prec, const int64_t* len, const int32_t* rowinfo,
                           const float* mask, int64_t ldmask, int io_flags, void* stream);

int styler_conv_gemm_impl(const float* x, int64_t ldx, const void* w, const float* scale, const float* shift,
                          const float* res, int64_t ldres, float* y, int64_t ldy, int B, int L, int cin, int n,
                          int kw, int pad, int act, int prec, const int64_t* len, void* stream) {
  return styler_conv_gemm_impl2(x, ldx, w, scale, shift, res, ldres, y, ldy, B, L, cin, n, kw, pad, act, prec, len,
                                nullptr, nullptr, 0, 0, stream);
}

int styler_conv_gemm_impl2(const float* x, int64_t ldx, const void* w, const float* scale, const float* shift,
                           const float* res, int64_t ldres, float* y, int64_t ldy, int B, int L, int cin, int n,
                           int kw, int pad, int act, int prec, const int64_t* len, const int32_t* rowinfo,
                           const float* mask, int64_t ldmask, int io_flags, void* stream) {
  // the one-shot registrations of this host thread are consumed FIRST, whatever happens below: an argument error must not
  // leave them for the next call (round-5 advisor)
  void* ws = nullptr;
  int64_t ws_bytes = 0;
  styler_gemm_take_workspace(&ws, &ws_bytes);      // consumed by this call, whatever engine takes it
  uint16_t* y3_reg = nullptr;
  int y3_parts = 0;
  styler_take_x3_out(&y3_reg, &y3_parts);          // (likewise: the split of the fp32 output, styler_set_x3_out)
  const int x16 = io_flags & STYLER_IO_X_BF16, y16 = io_flags & STYLER_IO_Y_BF16, m16 = io_flags & STYLER_IO_MASK_BF16;
  const int r16 = io_flags & STYLER_IO_RES_BF16;
  if ((x16 || y16 || r16) && (prec != STYLER_PREC_BF16 || (y16 && (ldy & 3)) || (x16 && (ldx & 7)) || (r16 && (!res || (ldres & 3)))))
    return STYLER_EINVAL;
  if (!x || !w || !y || B <= 0 || L <= 0 || cin <= 0 || n <= 0 || kw <= 0 || kw > 9 || pad < 0 || pad >= kw)
    return STYLER_EINVAL;
  if ((cin & 3) || (ldx & 3) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return STYLER_EALIGN;
  if ((n & 3) || (ldy & 3) || ((uintptr_t)y & 15) || (res && ((ldres & 3) || ((uintptr_t)res & (r16 ? 7 : 15))))) return STYLER_EALIGN;
  if (prec == STYLER_PREC_BF16 && (cin & 7)) return STYLER_EALIGN;
  if (rowinfo && B != 1) return STYLER_EINVAL;
  if ((int64_t)B * L >= ((int64_t)1 << 31) || ldy >= (1 << 22) || ldres >= (1 << 22) || ldmask >= (1 << 22)) return STYLER_EINVAL;
  if (mask && ((ldmask & 3) || ((uintptr_t)mask & 15))) return STYLER_EALIGN;
  GemmArgs a{x, ldx, w, scale, shift, res, ldres, y, ldy, B, L, cin, n, kw, act, pad, len, 0, 0,
             reinterpret_cast<const int2*>(rowinfo), mask, ldmask, m16 ? 1 : 0, g_gemm_trace};
  a.res16 = r16 ? 1 : 0;
  if (io_flags & STYLER_IO_X3A) {                  // compact bf16x3 activation rows [hi | lo]: whole 64-channel chunks per part
    if (prec != STYLER_PREC_BF16 || !x16 || (cin % 192) || ldx < 2 * (int64_t)(cin / 3)) return STYLER_EINVAL;
    a.x3n1 = cin / 192;
  }
  hipStream_t st = (hipStream_t)stream;
  a.y3 = y3_reg; a.y3parts = y3_parts;
  if (a.y3 && (y16 || prec != STYLER_PREC_BF16 || (a.y3parts != 2 && a.y3parts != 3) || ((uintptr_t)a.y3 & 7) ||
               (int64_t)B * L * a.y3parts * n * 2 >= ((int64_t)1 << 31)))
    return STYLER_EINVAL;
  if (prec == STYLER_PREC_BF16) {                  // large launches on bf16 activations: the 256 x 256 LDS-DMA engine (gemm256.hip)
    const int r = styler_gemm256_try(a, x16, y16, st, ws, ws_bytes);
    if (r) return r < 0 ? r : 0;
    const int ks = styler_gemm_small_ksplit(B, L, cin, n, kw, act, mask != nullptr);
    if (ks > 1 && ws && ws_bytes >= (int64_t)ks * B * L * n * 4 && !((uintptr_t)ws & 15) && (!len || (rowinfo && B == 1))) {
      a.ksplit = ks; a.part = reinterpret_cast<float*>(ws);
      const int rc = launch_gemm<1, 1, true>(a, st, x16, 0);
      if (rc) return rc;
      return styler_gemm_combine(a, ks, y16, st);
    }
  }
  const bool big = styler_conv_gemm_variant(B, L, cin, n, kw, prec) & 1;
  // narrow outputs (64 < n <= 96) over many rows: one 128 x 96 tile per row block (see the kernel's WM note)
  if (g_n96_enabled && prec == STYLER_PREC_BF16 && !big && n > 64 && n <= 96 && (int64_t)B * L >= g_n96_min_rows)
    return launch_gemm<1, 3, true, 4>(a, st, x16, y16);
  if (prec == STYLER_PREC_BF16) return big ? launch_gemm<2, 2, true>(a, st, x16, y16) : launch_gemm<1, 1, true>(a, st, x16, y16);
  return big ? launch_gemm<2, 2, false>(a, st, x16, y16) : launch_gemm<1, 1, false>(a, st, x16, y16);
}

extern "C" int styler_conv_gemm(const float* x, int64_t ldx, const void* w, const float* scale,
                                const float* shift, const float* res, int64_t ldres, float* y,
                                int64_t ldy, int B, int L, int cin, int n, int kw, int act, int prec,
                                const int64_t* len, const float* mask, int64_t ldmask, int io_flags, void* stream) {
  if (!(kw & 1)) return STYLER_EINVAL;
  return styler_conv_gemm_impl2(x, ldx, w, scale, shift, res, ldres, y, ldy, B, L, cin, n, kw, kw / 2, act, prec, len,
                                nullptr, mask, ldmask, io_flags, stream);
}

extern "C" int styler_conv_gemm_pad(const float* x, int64_t ldx, const void* w, const float* scale,
                                    const float* shift, const float* res, int64_t ldres, float* y, int64_t ldy,
                                    int B, int L, int cin, int n, int kw, int pad, int act, int prec, void* stream) {
  return styler_conv_gemm_impl2(x, ldx, w, scale, shift, res, ldres, y, ldy, B, L, cin, n, kw, pad, act, prec, nullptr,
                                nullptr, nullptr, 0, 0, stream);
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

// bf16 -> fp32 (exact): the way back of the optional bf16 gradient all-reduce (training.TrainState, STYLER_ALLREDUCE_BF16)
__global__ void cast_from_bf16_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, int64_t count) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < count; i += stride) {
    if (i + 3 < count) {
      const uint2 u = *reinterpret_cast<const uint2*>(src + i);
      *reinterpret_cast<float4*>(dst + i) = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                                        __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    } else {
      for (int64_t k = i; k < count; ++k) dst[k] = __uint_as_float((uint32_t)src[k] << 16);
    }
  }
}

extern "C" int styler_cast_from_bf16(const uint16_t* src, float* dst, int64_t count, void* stream) {
  if (!src || !dst || count < 0) return STYLER_EINVAL;
  if (count == 0) return 0;
  if (((uintptr_t)dst & 15) || ((uintptr_t)src & 7)) return STYLER_EALIGN;
  int64_t blocks = (count / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_from_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, count);
  return launch_status();
}

template <typename OutT>
__global__ void repack_conv_kernel(const float* __restrict__ src, OutT* __restrict__ dst, int n, int cin, int kw,
                                   int to_kernel) {
  const int64_t total = (int64_t)n * cin * kw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // i enumerates the DESTINATION
    float v;
    if (to_kernel) {  // dst [n, kw, cin] <- src [n, cin, kw]
      const int c = i % cin; const int j = (i / cin) % kw; const int64_t o = i / ((int64_t)cin * kw);
      v = src[(o * cin + c) * kw + j];
    } else {          // dst [n, cin, kw] <- src [n, kw, cin]
      const int j = i % kw; const int c = (i / kw) % cin; const int64_t o = i / ((int64_t)cin * kw);
      v = src[(o * kw + j) * cin + c];
    }
    if constexpr (sizeof(OutT) == 2) dst[i] = (OutT)f32_to_bf16_bits(v); else dst[i] = v;
  }
}

extern "C" int styler_repack_conv_weight(const float* src, void* dst, int n, int cin, int kw, int to_kernel_layout,
                                         int out_bf16, void* stream) {
  if (!src || !dst || n <= 0 || cin <= 0 || kw <= 0) return STYLER_EINVAL;
  const int64_t total = (int64_t)n * cin * kw;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (out_bf16)
    hipLaunchKernelGGL(repack_conv_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<uint16_t*>(dst), n, cin, kw, to_kernel_layout);
  else
    hipLaunchKernelGGL(repack_conv_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<float*>(dst), n, cin, kw, to_kernel_layout);
  return launch_status();
}

// Packed-rows variant (the decoder runs on the valid frames only): B = 1, L = row capacity, rowinfo = the
// (t, len - 1 - t) pairs of styler_pack_plan, len = its row counter (tiles behind the data are skipped).
extern "C" int styler_conv_gemm_packed(const float* x, int64_t ldx, const void* w, const float* scale,
                                       const float* shift, const float* res, int64_t ldres, float* y, int64_t ldy,
                                       int rows, int cin, int n, int kw, int act, int prec, const int64_t* nrows,
                                       const int32_t* rowinfo, const float* mask, int64_t ldmask, int io_flags,
                                       void* stream) {
  if (!(kw & 1) || !nrows || !rowinfo) return STYLER_EINVAL;
  return styler_conv_gemm_impl2(x, ldx, w, scale, shift, res, ldres, y, ldy, 1, rows, cin, n, kw, kw / 2, act, prec,
                                nrows, rowinfo, mask, ldmask, io_flags, stream);
}

extern "C" int styler_abi_version(void) { return 1; }
