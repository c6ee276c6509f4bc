// Linear(K -> 256) + bias + dropout + residual + LayerNorm(256) + pad mask as ONE kernel (round 6).
//
//   o[m, :]  = a[m, :] W^T + bias                         a: bf16 rows [rows, K], W: bf16 [256, K] (nn.Linear layout)
//   s[m, :]  = dropout(o[m, :], p) + res[m, :]            (the pre-norm sum: what styler_layernorm_bwd reads)
//   y[m, :]  = LayerNorm_256(s[m, :]) * gamma + beta      rows t >= len[b]: y = 0 (masked_fill), s not written
//
// Replaces the tail of both sublayers of an FFT block in throughput (bf16) mode: the output projection `fc` of
// MultiHeadAttention followed by dropout + residual + LayerNorm (transformer/SubLayers.py:55-61) and the second convolution
// `w_2` (k = 1) of PositionwiseFeedForward followed by the same (SubLayers.py:86-89), plus the masked_fill of Layers.py:29,32.
// Before: styler_conv_gemm wrote the fp32 projection (1 KB per row), styler_add_layernorm read it back with the residual
// and wrote y and the sum.  A 128-row x 256-column tile owns whole rows, so the normalisation runs on the accumulators: the
// fp32 projection never exists in HBM, one launch instead of two.  Same dropout stream (common.h: one keyed hash per four
// consecutive elements of the [rows, 256] tensor), same statistics (of the bf16-rounded sum when the sum is stored as bf16)
// as add_layernorm_kernel -- layernorm_bwd_kernel regenerates the mask and the statistics from what this kernel stores.
//
// Engine: 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 64 x 64 = 2 x 2 MFMA tiles (v_mfma_f32_32x32x16_bf16).
// Operands go HBM / L2 -> LDS by DMA (buffer_load_dwordx4 ... lds) in K steps of 64: 16 KB of activations + 32 KB of weights
// per step into one of THREE stages (144 KB), two steps ahead of the MFMAs; the LDS image, its XOR swizzle (applied to the
// DMA's source address and to the fragment read address, never to the DMA destination) and the fragment layout are those of
// gemm256.hip.  One barrier per K step:
//   RAW  a stage is read after every wave waited (vmcnt) for its own pieces of that step and then met the barrier;
//   WAR  a stage is overwritten by the DMA issued behind the barrier of step t, and was last read in step t - 1.
// The kernel is HBM-bound (K = 1024: 2 KB of activations + 0.5 KB of residual in, 1 KB out per row against 0.5 MFLOP).
#include <cstdlib>
#include "gemm_args.h"

namespace {

constexpr int LL_BM = 128, LL_N = 256, LL_BK = 64;
constexpr int LL_A_BYTES = LL_BM * 128;             // 128 rows x 64 bf16
constexpr int LL_B_BYTES = LL_N * 128;              // 256 weight rows x 64 bf16
constexpr int LL_STAGE = LL_A_BYTES + LL_B_BYTES;   // 48 KB
constexpr int LL_NSTAGE = 3;
constexpr int LL_SMEM = LL_NSTAGE * LL_STAGE;       // 144 KB
constexpr int LL_CLD = 264;                         // epilogue staging row stride (floats): 4 rows apart = 32 banks apart
static_assert(LL_BM * LL_CLD * 4 <= LL_SMEM, "the fp32 tile is staged in the operand stages");
constexpr uint32_t LL_OOB = 0x80000000u;            // beyond every descriptor's num_records (< 2^31)

typedef __attribute__((address_space(3))) void ll_lds_void;

struct LinLnArgs {
  const uint16_t* a; int64_t lda; int K;
  const uint16_t* w;
  const float* bias;
  const void* res; int64_t ldres;
  const float* gamma; const float* beta;
  void* y; int64_t ldy;
  void* sum; int64_t ldsum;
  uint16_t* y16; int64_t ldy16;
  int64_t rows; int L;
  const int64_t* len;
  float drop_p; uint64_t drop_seed; const uint64_t* epoch;
  int io;
  uint64_t* trace;                                   // styler_linear_ln_set_trace: 8 words per block (phase timestamps), or null
};

__device__ __forceinline__ void ll_dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t lds_byte, uint32_t voff, char* smem) {
  // 64 lanes x 16 B -> LDS bytes [lds_byte, lds_byte + 1024), lane l at + 16 l (wave-uniform destination base)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (ll_lds_void*)(smem + lds_byte), 16, voff, 0, 0, 0);
}

// wave_sum (common.h) of four independent values, level by level: the same partners in the same order per value (the same
// bits), but no instruction depends on the one in front of it -- a lone chain pays the DPP / permlane wait states at every level.
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j]), false, false);
    v[j] = __uint_as_float(h[0]) + __uint_as_float(h[1]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[j]), __float_as_uint(v[j]), false, false);
    v[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] += dpp_f32<0x128>(v[j]);              // row_ror:8
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] += dpp_f32<0x124>(v[j]);              // row_ror:4
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] += dpp_f32<0x4E>(v[j]);               // quad_perm [2,3,0,1]
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] += dpp_f32<0xB1>(v[j]);               // quad_perm [1,0,3,2]
}

}  // namespace

// IOC: the storage formats of res / y / sum as a compile-time constant -- 0: all fp32 (the encoder's stream), 7: all bf16 (the
// packed decoder's stream), -1: read from a.io at run time (any other combination of the STYLER_LN_* flags).
template <int IOC>
__global__ __launch_bounds__(512) void linear_ln_kernel(const LinLnArgs a) {
  __shared__ __attribute__((aligned(1024))) char smem[LL_SMEM];          // the ONLY LDS object of the kernel

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * LL_BM;
  const int io = IOC >= 0 ? IOC : a.io;
  const bool res16 = io & STYLER_LN_RES_BF16, yb16 = io & STYLER_LN_Y_BF16, sum16 = io & STYLER_LN_SUM_BF16;
  const bool one_item = (int64_t)a.L >= a.rows;                          // packed rows: len[0] = the valid row count
  uint64_t stamp[6];
  if (a.trace) stamp[0] = wall_clock64();

  // ---- row masks of this wave's 16 epilogue rows (wave w owns tile rows 16 w .. 16 w + 15): lane u < 16 looks up its row's
  // length with ONE vector load, the wave shares the verdicts as a ballot (the kernel's pointers are not restrict-qualified,
  // so a per-row `len[b]` would be a vector load + wait each: a chain of 16 memory latencies) ----
  constexpr int RPW = LL_BM / 8;                                         // epilogue rows per wave
  uint32_t live, inside;                                                 // bit u: row u of this wave is unmasked / exists
  int64_t nvalid = a.rows;                                               // packed rows: the valid row count
  const uint64_t epoch_word = a.epoch ? *a.epoch : 0ull;                 // (the dropout step counter: same wait as the lengths)
  {
    const int64_t m = m0 + wave * RPW + (lane & (RPW - 1));
    const bool in = m < a.rows;
    bool ok = in;
    if (a.len) {
      uint32_t b = 0, t = (uint32_t)(in ? m : 0);
      if (!one_item) { b = t / (uint32_t)a.L; t -= b * (uint32_t)a.L; }
      const int64_t lb = a.len[b];
      ok = in && (int64_t)t < lb;
      if (one_item) nvalid = lb;                                         // (every lane loaded len[0])
    }
    inside = (uint32_t)__ballot(in) & 0xffffu;
    live = (uint32_t)__ballot(ok) & 0xffffu;
  }
  // ---- tiles behind the data of a packed tensor: zeros (what add_layernorm_kernel writes for masked rows), nothing read ----
  if (a.len && one_item && m0 >= nvalid) {
    for (int i = tid; i < LL_BM * 64; i += 512) {
      const int64_t m = m0 + (i >> 6);
      const int c = (i & 63) * 4;
      if (m < a.rows) {
        stg4(a.y, m * a.ldy + c, make_float4(0.f, 0.f, 0.f, 0.f), yb16);
        if (a.y16) *reinterpret_cast<uint2*>(a.y16 + m * a.ldy16 + c) = make_uint2(0u, 0u);
      }
    }
    return;
  }

  if (a.trace) stamp[1] = wall_clock64();
  // ---- the epilogue's residual rows, fetched FIRST (ahead of the operand DMA in the in-order vmcnt queue, so every wait of
  // the K loop covers them) as RAW loads -- converted in the epilogue, a conversion here would put a wait behind every load.
  // They sit in registers while the K loop runs: a block is alone on its CU (144 KB of LDS), nothing else would hide them.
  const int c4 = lane * 4;
  const int r_es = res16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t r_rs = [&] {                              // rows past the tensor: out of range, zeros
    const int64_t rec = a.res ? ((a.rows - 1) * a.ldres + LL_N) * r_es : 0;          // < 2^31 (checked by the host)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : (const void*)a.gamma), 0, (int)rec, 0x00020000);
  }();
  i32x4 rraw[RPW];
  {
    const uint32_t o0 = (uint32_t)(((m0 + wave * RPW) * a.ldres + c4) * r_es), ostep = (uint32_t)(a.ldres * r_es);
    if (res16) {
#pragma unroll
      for (int u = 0; u < RPW; ++u) {
        const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r_rs, o0 + u * ostep, 0, 0);
        rraw[u] = i32x4{t.x, t.y, 0, 0};
      }
    } else {
#pragma unroll
      for (int u = 0; u < RPW; ++u) rraw[u] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, o0 + u * ostep, 0, 0);
    }
  }
  // ... and the per-channel parameters and the dropout key (the step counter is a device word), for the same reason
  const float4 bs = a.bias ? *reinterpret_cast<const float4*>(a.bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 g = *reinterpret_cast<const float4*>(a.gamma + c4);
  const float4 bt = *reinterpret_cast<const float4*>(a.beta + c4);
  const uint2 key = dropout_key(a.drop_seed + epoch_word * 0xD6E8FEB86659FD93ull);      // == mix_drop_epoch(seed, epoch)
  const uint32_t thr = dropout_thr16(a.drop_p);
  const float dsc = 1.f / (1.f - a.drop_p);

  // ---- DMA source addresses.  Activation unit: 16 pieces of 1 KB (8 rows x 128 B), wave w issues pieces w and w + 8;
  // weight unit: 32 pieces, wave w issues w, w + 8, w + 16, w + 24.  Lane l of piece p: unit row i = 8 p + (l >> 3),
  // physical 16-byte chunk l & 7 = logical chunk ^ ((i >> 1) & 7).
  const __amdgpu_buffer_rsrc_t a_rs = [&] {
    const int64_t rec = ((a.rows - 1) * a.lda + a.K) * 2;                // < 2^31 (checked by the host)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.a), 0, (int)rec, 0x00020000);
  }();
  const __amdgpu_buffer_rsrc_t w_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w), 0, (int)((int64_t)LL_N * a.K * 2), 0x00020000);
  uint32_t va[2], vb[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = 8 * (wave + 8 * q) + (lane >> 3);
    const int cl = (lane & 7) ^ ((i >> 1) & 7);
    const int64_t m = m0 + i;
    va[q] = m < a.rows ? (uint32_t)(m * a.lda * 2 + cl * 16) : LL_OOB;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 8 * (wave + 8 * q) + (lane >> 3);
    const int cl = (lane & 7) ^ ((i >> 1) & 7);
    vb[q] = (uint32_t)(i * a.K * 2 + cl * 16);
  }
  auto issue = [&](const int t, const uint32_t st) {                     // the operands of K step t into stage byte offset st
    const uint32_t sh = (uint32_t)t * (LL_BK * 2);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      ll_dma16(a_rs, st + (uint32_t)(wave + 8 * q) * 1024u, va[q] == LL_OOB ? LL_OOB : va[q] + sh, smem);
#pragma unroll
    for (int q = 0; q < 4; ++q) ll_dma16(w_rs, st + LL_A_BYTES + (uint32_t)(wave + 8 * q) * 1024u, vb[q] + sh, smem);
  };

  // ---- fragment read addresses: lane (li, lh), MFMA sub-step s (16 of the 64 k): logical chunk 2 s + lh of unit row
  // base + li; (row >> 1) & 7 == (li >> 1) & 7 because every base is a multiple of 32
  uint32_t ra[4], rb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t ch = (uint32_t)(((2 * s + lh) ^ ((li >> 1) & 7)) * 16);
    ra[s] = (uint32_t)((wr * 64 + li) * 128) + ch;                       // + i * 4096 (32 rows) + stage
    rb[s] = (uint32_t)(LL_A_BYTES + (wc * 64 + li) * 128) + ch;          // + jj * 4096 + stage
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

#define LL_SYNC()                            \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)
#define LL_FRAG(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))

  const int nsteps = a.K / LL_BK;
  issue(0, 0u);
  if (nsteps > 1) issue(1, (uint32_t)LL_STAGE);
  uint32_t st = 0u, st2 = 2u * LL_STAGE;                                 // stage of step t / of step t + 2
  for (int t = 0; t < nsteps; ++t) {
    if (t + 1 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // step t landed (this wave's pieces); t + 1 in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LL_SYNC();
    if (a.trace && t == 0) stamp[2] = wall_clock64();
    if (t + 2 < nsteps) issue(t + 2, st2);
    // fragment reads one MFMA sub-step (16 of the 64 k) ahead of the MFMAs that consume them: 4 ds_read_b128 in flight
    // under every group of 4 MFMAs (the compiler counts lgkmcnt per group)
    bf16x8 fa[2][2], fb[2][2];                                           // [sub-step parity][tile]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[0][i] = LL_FRAG(st + i * 4096 + ra[0]);
      fb[0][i] = LL_FRAG(st + i * 4096 + rb[0]);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[(s + 1) & 1][i] = LL_FRAG(st + i * 4096 + ra[s + 1]);
          fb[(s + 1) & 1][i] = LL_FRAG(st + i * 4096 + rb[s + 1]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][jj], acc[i][jj], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    st = st + LL_STAGE == (uint32_t)LL_SMEM ? 0u : st + LL_STAGE;
    st2 = st2 + LL_STAGE == (uint32_t)LL_SMEM ? 0u : st2 + LL_STAGE;
  }
#undef LL_FRAG
  LL_SYNC();                                                             // every wave is done with the operand stages
#undef LL_SYNC
  if (a.trace) stamp[3] = wall_clock64();

  // ---- epilogue: the fp32 tile through LDS (C layout of v_mfma_f32_32x32x16: col = lane & 31, row = (r & 3) + 8 (r >> 2)
  // + 4 (lane >> 5)), then one wave per row exactly as add_layernorm_kernel: lane = 4 consecutive channels ----
  float* const cst = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        cst[(wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LL_CLD + wc * 64 + jj * 32 + li] = acc[i][jj][r];
  __syncthreads();
  if (a.trace) stamp[4] = wall_clock64();

  // Rows in groups of four, branch-free: the four rows of a group are independent dependency chains in ONE basic block (a
  // row is ~85 vector instructions, most of them dependent -- one row at a time ran at the latency of that chain, 0.5 us per
  // row).  Masking is done by the stores: a buffer store whose offset lies past its descriptor is dropped.
  auto out_rsrc = [&](void* base, int64_t ld, int es) {
    const int64_t rec = base ? ((a.rows - 1) * ld + LL_N) * es : 0;      // < 2^31 (checked by the host); null: every store dropped
    return __builtin_amdgcn_make_buffer_rsrc(base ? base : (void*)smem, 0, (int)rec, 0x00020000);
  };
  const int y_es = yb16 ? 2 : 4, s_es = sum16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t y_rs = out_rsrc(a.y, a.ldy, y_es), s_rs = out_rsrc(a.sum, a.ldsum, s_es),
                               y16_rs = out_rsrc(a.y16, a.ldy16, 2);
  auto store4 = [&](const __amdgpu_buffer_rsrc_t& rs, const uint32_t off, const float4 v, const bool as16) {
    if (as16) {
      const i32x2 o = {(int)cvt_pk_bf16_rne(v.x, v.y), (int)cvt_pk_bf16_rne(v.z, v.w)};
      __builtin_amdgcn_raw_buffer_store_b64(o, rs, off, 0, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const i32x4*>(&v), rs, off, 0, 0);
    }
  };
  constexpr int RG = 4;
#pragma unroll
  for (int u0 = 0; u0 < RPW; u0 += RG) {
    float4 v[RG];
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      const int u = u0 + j, r = wave * RPW + u;
      const int64_t m = m0 + r;
      const uint32_t dead = (((live >> u) & 1u) ^ 1u) << 31;              // 0x80000000 on a masked row: the offset leaves the descriptor
      float4 t = *reinterpret_cast<const float4*>(&cst[r * LL_CLD + c4]);
      t.x += bs.x; t.y += bs.y; t.z += bs.z; t.w += bs.w;
      t = dropout_apply4(t, key, (uint64_t)m * 256 + c4, thr, dsc);      // (no dropout: thr = 0 keeps all, scale 1)
      const float4 rq = res16 ? raw4_f32(make_uint2((uint32_t)rraw[u].x, (uint32_t)rraw[u].y))
                              : make_float4(__int_as_float(rraw[u].x), __int_as_float(rraw[u].y), __int_as_float(rraw[u].z),
                                            __int_as_float(rraw[u].w));
      t.x += rq.x; t.y += rq.y; t.z += rq.z; t.w += rq.w;
      // the pre-norm sum, on unmasked rows only; a bf16 sum is what the backward will see: normalise THAT value
      store4(s_rs, (uint32_t)((m * a.ldsum + c4) * s_es) | dead, t, sum16);
      if (sum16) t = raw4_f32(make_uint2(cvt_pk_bf16_rne(t.x, t.y), cvt_pk_bf16_rne(t.z, t.w)));
      v[j] = t;
    }
    float mean[RG], rstd[RG];
#pragma unroll
    for (int j = 0; j < RG; ++j) mean[j] = v[j].x + v[j].y + v[j].z + v[j].w;
    wave_sum4(mean);
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      mean[j] *= 1.f / 256.f;
      v[j].x -= mean[j]; v[j].y -= mean[j]; v[j].z -= mean[j]; v[j].w -= mean[j];
      rstd[j] = v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    }
    wave_sum4(rstd);
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      // 1 / sqrt(var + eps): v_rsq_f32 + one Newton step (<= 1 ulp; add_layernorm_kernel's sqrt + IEEE division is ~35
      // instructions per row -- the backward recomputes its own statistics from the stored sum either way)
      const float x = rstd[j] * (1.f / 256.f) + 1e-5f;
      const float y0 = __builtin_amdgcn_rsqf(x);
      rstd[j] = y0 * (1.5f - 0.5f * x * y0 * y0);
    }
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      const int u = u0 + j;
      const int64_t m = m0 + wave * RPW + u;
      const uint32_t outside = (((inside >> u) & 1u) ^ 1u) << 31;         // a row past the tensor: no store
      const uint32_t keep = 0u - ((live >> u) & 1u);                      // all ones on an unmasked row
      const float rs = rstd[j];
      float4 o = make_float4(v[j].x * rs * g.x + bt.x, v[j].y * rs * g.y + bt.y, v[j].z * rs * g.z + bt.z,
                             v[j].w * rs * g.w + bt.w);
      o = make_float4(__uint_as_float(__float_as_uint(o.x) & keep), __uint_as_float(__float_as_uint(o.y) & keep),
                      __uint_as_float(__float_as_uint(o.z) & keep), __uint_as_float(__float_as_uint(o.w) & keep));   // masked_fill
      store4(y_rs, (uint32_t)((m * a.ldy + c4) * y_es) | outside, o, yb16);
      store4(y16_rs, (uint32_t)((m * a.ldy16 + c4) * 2) | outside, o, true);
    }
  }
  if (a.trace) {                                                         // [block, entry, masks known, step 0 landed, K loop done, tile staged, rows issued, stores acknowledged]
    stamp[5] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t done = wall_clock64();
    if (tid == 0) {
      uint64_t* t = a.trace + (int64_t)blockIdx.x * 8;
      t[0] = blockIdx.x;
      for (int i = 0; i < 6; ++i) t[1 + i] = stamp[i];
      t[7] = done;
    }
  }
}

// Phase timestamps (100 MHz counter) of every block of the launches that follow styler_linear_ln_set_trace(buf): a measurement
// hook (tools/linear_ln_trace.py), off (null) by default.
static uint64_t* g_ll_trace = nullptr;
extern "C" int styler_linear_ln_set_trace(void* buf) { g_ll_trace = reinterpret_cast<uint64_t*>(buf); return 0; }

// Shapes the kernel takes (styler_linear_ln returns STYLER_EINVAL / STYLER_EALIGN for anything else; the host layer asks
// first and keeps the two-launch path otherwise).
extern "C" int styler_linear_ln_ok(int64_t rows, int K, int n, int64_t lda) {
  if (rows <= 0 || n != LL_N || K < LL_BK || (K % LL_BK) || (lda & 7) || lda < K) return 0;
  if (((rows - 1) * lda + K) * 2 >= ((int64_t)1 << 31) || rows >= ((int64_t)1 << 31)) return 0;
  return 1;
}

extern "C" int styler_linear_ln(const void* a, int64_t lda, int K, const void* w, const float* bias, const void* res,
                                int64_t ldres, const float* gamma, const float* beta, void* y, int64_t ldy, void* sum_out,
                                int64_t ldsum, uint16_t* y16, int64_t ldy16, int B, int L, const int64_t* len, float drop_p,
                                uint64_t drop_seed, int io_flags, void* stream) {
  if (!a || !w || !gamma || !beta || !y || B <= 0 || L <= 0) return STYLER_EINVAL;
  const int64_t rows = (int64_t)B * L;
  if (!styler_linear_ln_ok(rows, K, LL_N, lda)) return STYLER_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f) return STYLER_EINVAL;
  if (((uintptr_t)a & 15) || ((uintptr_t)w & 15) || (bias && ((uintptr_t)bias & 15)) || ((uintptr_t)gamma & 15) ||
      ((uintptr_t)beta & 15))
    return STYLER_EALIGN;
  if ((res && (ldres & 3)) || (ldy & 3) || (sum_out && (ldsum & 3)) || (y16 && ((ldy16 & 3) || ((uintptr_t)y16 & 7))))
    return STYLER_EALIGN;
  if (res && ((rows - 1) * ldres + LL_N) * ((io_flags & STYLER_LN_RES_BF16) ? 2 : 4) >= ((int64_t)1 << 31)) return STYLER_EINVAL;
  {
    const int y_es = (io_flags & STYLER_LN_Y_BF16) ? 2 : 4, s_es = (io_flags & STYLER_LN_SUM_BF16) ? 2 : 4;
    const int64_t lim = (int64_t)1 << 31;
    if (((rows - 1) * ldy + LL_N) * y_es >= lim || (sum_out && ((rows - 1) * ldsum + LL_N) * s_es >= lim) ||
        (y16 && ((rows - 1) * ldy16 + LL_N) * 2 >= lim))
      return STYLER_EINVAL;
  }
  LinLnArgs k;
  k.a = reinterpret_cast<const uint16_t*>(a); k.lda = lda; k.K = K;
  k.w = reinterpret_cast<const uint16_t*>(w);
  k.bias = bias;
  k.res = res; k.ldres = ldres;
  k.gamma = gamma; k.beta = beta;
  k.y = y; k.ldy = ldy;
  k.sum = sum_out; k.ldsum = ldsum;
  k.y16 = y16; k.ldy16 = ldy16;
  k.rows = rows; k.L = L;
  k.len = len;
  k.drop_p = drop_p; k.drop_seed = drop_seed; k.epoch = g_styler_drop_epoch;
  k.io = io_flags;
  k.trace = g_ll_trace;
  const dim3 grid((unsigned)((rows + LL_BM - 1) / LL_BM));
  const int io3 = io_flags & (STYLER_LN_RES_BF16 | STYLER_LN_Y_BF16 | STYLER_LN_SUM_BF16);
  if (io3 == 7) hipLaunchKernelGGL(linear_ln_kernel<7>, grid, dim3(512), 0, (hipStream_t)stream, k);
  else if (io3 == 0) hipLaunchKernelGGL(linear_ln_kernel<0>, grid, dim3(512), 0, (hipStream_t)stream, k);
  else hipLaunchKernelGGL(linear_ln_kernel<-1>, grid, dim3(512), 0, (hipStream_t)stream, k);
  return launch_status();
}
