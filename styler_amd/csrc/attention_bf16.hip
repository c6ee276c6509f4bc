// bf16-operand attention (throughput mode): forward + backward on v_mfma_f32_32x32x16_bf16, fp32 softmax / accumulate.
// Same decomposition and MFMA orientations as attention.hip / attention_bwd.hip (a lane owns one query -- or, in the
// dK/dV kernel, one key -- column; P / dS stay in the accumulator registers that feed the next product).
//
// The bf16 MFMA consumes 8 consecutive k per lane.  For the products whose k axis is the head dimension (Q K^T,
// dO V^T) the operands are row-major [row][d] tiles (ds_read_b128) or registers.  For the products whose k axis is
// the key (or query) axis (P V, dS K, P^T dO, dS^T Q) the register operand is the packed P / dS accumulator, whose lane
// holds rows (r&3) + 8*(r>>2) + 4*h of the 32-row block; MFMA step s2 takes registers 8*s2 .. 8*s2+7, i.e. the rows
// {16*s2 + 4h + 0..3} and {16*s2 + 8 + 4h + 0..3}.  The k <-> row map only has to agree between the two operands, so the
// LDS operand is the SAME row-major tile read through gfx950's LDS transpose read (ds_read_b64_tr_b16: a 16-lane group
// addresses a [4 rows][16 features] block and every lane receives the 4 rows of ITS feature): two reads at exactly
// those two row runs -- no transposed copy of any tile exists, not in LDS and not in registers.
//
// Tiles are fetched with raw buffer loads against a per-item descriptor (rows at or past the item's length are past
// num_records and read as zeros): no lane branches, so all loads of a tile are in flight together and ONE wait precedes
// the LDS stores.  (Conditional loads compiled to a branch + s_waitcnt vmcnt(0) per load: 5 serial memory round trips
// per tile in the forward, 8 -- plus scratch traffic -- in the dK/dV kernel.)
#include "common.h"

#define AD 64
#define ALD 36            // dwords per LDS row (64 bf16 + 16 B pad)
// Register budgets are stated, not left to the compiler: without the second __launch_bounds__ argument hipcc took 296
// registers for the dK/dV kernel (one wave per SIMD) and 192 for the dQ kernel (two) where 168 / 168 do without spills --
// dQ + dK/dV of a decoder layer 97 -> 81 us from that alone.
#ifndef STYLER_ATTN_DKV_WAVES
#define STYLER_ATTN_DKV_WAVES 3
#endif
#ifndef STYLER_ATTN_DQ_WAVES
#define STYLER_ATTN_DQ_WAVES 3
#endif
#ifndef STYLER_ATTN_FWD_WAVES
#define STYLER_ATTN_FWD_WAVES 3
#endif
#ifndef STYLER_ATTN_FWD_PREFETCH
#define STYLER_ATTN_FWD_PREFETCH 0
#endif
#ifndef STYLER_ATTN_DQ_PREFETCH
#define STYLER_ATTN_DQ_PREFETCH 0
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

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

// Q16 kernels (the fused q | k | v tensor stored as bf16, STYLER_IO_X_BF16): 8 consecutive bf16 as they are.  The
// 1 / sqrt(d_k) * log2(e) scale that the fp32-input kernels multiply into q (or k) BEFORE the rounding to bf16 cannot be
// applied to a stored bf16 value without a second rounding, so these kernels multiply the raw scores instead -- inside the
// exponent's argument, exp2(fma(s, c, -m c)): a v_fma where the other form has a v_sub, no extra instruction, and every
// operand is still rounded exactly once.
__device__ __forceinline__ bf16x8 load8_raw16(const void* base, int64_t idx) {
  const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + idx);
  return *reinterpret_cast<const bf16x8*>(&v);
}
#define ATTN_SCALE_LOG2 (0.125f * 1.44269504088896f)

// accumulator registers 8*s2 .. 8*s2+7 -> bf16x8 (the B operand of MFMA step s2)
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s2) {
  uint4 v = make_uint4(cvtpk(a[s2 * 8 + 0], a[s2 * 8 + 1]), cvtpk(a[s2 * 8 + 2], a[s2 * 8 + 3]),
                       cvtpk(a[s2 * 8 + 4], a[s2 * 8 + 5]), cvtpk(a[s2 * 8 + 6], a[s2 * 8 + 7]));
  return *reinterpret_cast<bf16x8*>(&v);
}

// Descriptor of one item's [nrows][64] slice of a row-major tensor (`base` = the item's first row at the head's first
// column, ld in floats): rows >= nrows are out of range and read as zeros.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const float* base, int64_t ld, int nrows) {
  int64_t rec = ((int64_t)(nrows - 1) * ld + AD) * 4;
  rec = rec > 0x7fffffff ? 0x7fffffff : rec;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)rec, 0x00020000);
}

// the same for a bf16 tensor (ld in elements)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc16(const void* base, int64_t idx, int64_t ld, int nrows) {
  int64_t rec = ((int64_t)(nrows - 1) * ld + AD) * 2;
  rec = rec > 0x7fffffff ? 0x7fffffff : rec;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(reinterpret_cast<const uint16_t*>(base) + idx), 0, (int)rec, 0x00020000);
}
// 64 rows x 64 bf16 starting at row0: thread tid fetches rows (tid >> 3) + 32 p, columns (tid & 7) * 8 .. + 7 (16 bytes)
__device__ __forceinline__ void load_rows16(uint4 (&v)[2], __amdgpu_buffer_rsrc_t rs, int ld, int row0, int tid) {
  const uint32_t off = (uint32_t)(((row0 + (tid >> 3)) * ld + (tid & 7) * 8) * 2);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (uint32_t)(p * 32 * ld * 2), 0, 0);
    v[p] = *reinterpret_cast<const uint4*>(&t);
  }
}
// ... -> the same bf16 [64][ALD] tile, no conversion
__device__ __forceinline__ void store_rows16(uint32_t* dst, const uint4 (&v)[2], int tid) {
  uint32_t* d = dst + (tid >> 3) * ALD + (tid & 7) * 4;
#pragma unroll
  for (int p = 0; p < 2; ++p) *reinterpret_cast<uint4*>(&d[p * 32 * ALD]) = v[p];
}
// 64 rows x 64 floats starting at row0: thread tid fetches rows (tid >> 4) + 16 p, columns (tid & 15) * 4 .. + 3
__device__ __forceinline__ void load_rows(float4 (&v)[4], __amdgpu_buffer_rsrc_t rs, int ld, int row0, int tid) {
  const uint32_t off = (uint32_t)(((row0 + (tid >> 4)) * ld + (tid & 15) * 4) * 4);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (uint32_t)(p * 16 * ld * 4), 0, 0);
    v[p] = *reinterpret_cast<const float4*>(&t);
  }
}
// ... -> bf16 [64][ALD]
__device__ __forceinline__ void store_rows(uint32_t* dst, const float4 (&v)[4], int tid) {
  uint32_t* d = dst + (tid >> 4) * ALD + (tid & 15) * 2;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    *reinterpret_cast<uint2*>(&d[p * 16 * ALD]) = make_uint2(cvtpk(v[p].x, v[p].y), cvtpk(v[p].z, v[p].w));
}
// registers of one staged tile in either storage format
template <bool Q16> struct TileRegs { float4 v[4]; };
template <> struct TileRegs<true> { uint4 v[2]; };
template <bool Q16>
__device__ __forceinline__ void tile_load(TileRegs<Q16>& r, __amdgpu_buffer_rsrc_t rs, int ld, int row0, int tid) {
  if constexpr (Q16) load_rows16(r.v, rs, ld, row0, tid); else load_rows(r.v, rs, ld, row0, tid);
}
template <bool Q16>
__device__ __forceinline__ void tile_store(uint32_t* dst, const TileRegs<Q16>& r, int tid) {
  if constexpr (Q16) store_rows16(dst, r.v, tid); else store_rows(dst, r.v, tid);
}


// A operand of a product over the ROW axis of a row-major tile: feature d = dt*32 + (lane & 31), the two 4-row runs
// {16*s2 + 4*lh + 0..3} and {.. + 8} of 32-row block `blk`.  tq = lane & 15, tc = (lane >> 4) & 1 (the 16-lane group's
// feature half).  Lane tq addresses row (tq >> 2), features (tq & 3) * 4 .. + 3 of its group's [4][16] block.
__device__ __forceinline__ bf16x8 fragT(const uint32_t* tile, int dt, int tq, int tc, int blk, int s2, int lh) {
  const uint32_t* p = &tile[(blk * 32 + s2 * 16 + lh * 4 + (tq >> 2)) * ALD + dt * 16 + tc * 8 + (tq & 3) * 2];
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<uint32_t*>(p)));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<uint32_t*>(p + 8 * ALD)));
  const uint2 ua = *reinterpret_cast<const uint2*>(&a), ub = *reinterpret_cast<const uint2*>(&b);
  uint4 v = make_uint4(ua.x, ua.y, ub.x, ub.y);
  return *reinterpret_cast<bf16x8*>(&v);
}

// the same 64 features as bf16 (dqkv stored as bf16: its consumers -- the QKV dX GEMM and the weight gradients -- round
// it to bf16 while staging, so nothing changes but the bytes)
__device__ __forceinline__ void store_accT16(uint16_t* op, const f32x16& a0, const f32x16& a1, int lh, float scale) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int d = 8 * g + 4 * lh;
    *reinterpret_cast<uint2*>(op + d) = make_uint2(cvtpk(a0[g * 4 + 0] * scale, a0[g * 4 + 1] * scale),
                                                   cvtpk(a0[g * 4 + 2] * scale, a0[g * 4 + 3] * scale));
    *reinterpret_cast<uint2*>(op + 32 + d) = make_uint2(cvtpk(a1[g * 4 + 0] * scale, a1[g * 4 + 1] * scale),
                                                        cvtpk(a1[g * 4 + 2] * scale, a1[g * 4 + 3] * scale));
  }
}
// 32 zeros at element offset `off` of a dqkv row (fp32 or bf16 storage)
__device__ __forceinline__ void zero32(void* base, int64_t off, int out16) {
  if (out16) {
    uint16_t* op = reinterpret_cast<uint16_t*>(base) + off;
#pragma unroll
    for (int d = 0; d < 32; d += 8) *reinterpret_cast<uint4*>(op + d) = make_uint4(0u, 0u, 0u, 0u);
  } else {
    float* op = reinterpret_cast<float*>(base) + off;
#pragma unroll
    for (int d = 0; d < 32; d += 4) *reinterpret_cast<float4*>(op + d) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// round 5, bf16x3: the [hi | lo] split of the same 64 features (the values store_accT writes, split as styler_split3_bf16 does);
// op3 = the hi block's address of these features, the lo block sits lo_off elements behind it (styler_set_x3_out)
__device__ __forceinline__ void store_accT_x3(uint16_t* op3, int lo_off, const f32x16& a0, const f32x16& a1, int lh, float scale) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int d = 8 * g + 4 * lh;
    uint2 h, l;
    x3_split4(make_float4(a0[g * 4 + 0] * scale, a0[g * 4 + 1] * scale, a0[g * 4 + 2] * scale, a0[g * 4 + 3] * scale), h, l);
    *reinterpret_cast<uint2*>(op3 + d) = h;
    *reinterpret_cast<uint2*>(op3 + lo_off + d) = l;
    x3_split4(make_float4(a1[g * 4 + 0] * scale, a1[g * 4 + 1] * scale, a1[g * 4 + 2] * scale, a1[g * 4 + 3] * scale), h, l);
    *reinterpret_cast<uint2*>(op3 + 32 + d) = h;
    *reinterpret_cast<uint2*>(op3 + lo_off + 32 + d) = l;
  }
}
__device__ __forceinline__ void zero32_x3(uint16_t* op3, int lo_off) {          // 32 zeros in both blocks
#pragma unroll
  for (int d = 0; d < 32; d += 8) {
    *reinterpret_cast<uint4*>(op3 + d) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(op3 + lo_off + d) = make_uint4(0u, 0u, 0u, 0u);
  }
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

// XCD-aware block order.  Workgroups go to the eight XCDs round-robin in launch order, and every XCD has its own L2: in
// the natural (x = row block, y = head, z = item) order the 2-4 row blocks of one (item, head) -- which read the SAME
// K / V (or Q / dO) rows -- landed on different XCDs and each fetched those rows over the fabric again (L2 hit 29 %,
// 180 MB fetched per forward launch for 83 MB of q/k/v).  The grid is 1-D, a multiple of 8, and launch slot n runs
// logical block (n % 8) * (grid / 8) + n / 8: the logical blocks an XCD sees are consecutive, so the row blocks of a
// pair run back to back on one L2.
#ifndef STYLER_ATTN_XCD
#define STYLER_ATTN_XCD 1
#endif
__device__ __forceinline__ bool attn_block(int L, int B, int& bx, int& head, int& b) {
  const int gx = (L + 127) >> 7, total = gx * 4 * B;
  const int n = blockIdx.x;
#if STYLER_ATTN_XCD
  const int v = (n & 7) * (int)(gridDim.x >> 3) + (n >> 3);
#else
  const int v = n;
#endif
  if (v >= total) return false;
  bx = v % gx;
  const int t = v / gx;
  head = t & 3;
  b = t >> 2;
  return true;
}
static inline dim3 attn_grid(int L, int B) { return dim3((unsigned)((((L + 127) / 128) * 4 * B + 7) / 8 * 8)); }

// ------------------------------------------------------------------------------------------------- forward
// LAZY (round 4).  PMC on the config-4 launch (profiles/r04_attn_c4_pmc.txt): 13.2 VALU instructions per MFMA, the VALU pipe busy
// 64 % and the matrix pipe 33 % of the time, and the two add up to the kernel's duration -- at d_k = 64 a 64-key tile is 16
// MFMAs (512 cycles) against 34 quarter-rate v_exp_f32 and ~160 other vector instructions that compete with the MFMAs for
// the SIMD's issue slots.  LAZY trims the part of that which is not needed on every tile: the running maximum is only
// moved when a tile's maximum exceeds it by more than 8 in the log2 domain (p <= 256: exact in fp32, harmless in bf16; o and
// l carry the same stale maximum, o / l does not change), so the rescaling of the 32 output accumulators leaves the common
// path behind a wave-uniform branch.
template <bool Q16, bool LAZY = false>
__global__ __launch_bounds__(256, STYLER_ATTN_FWD_WAVES) void attention_fwd_bf16_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 float* __restrict__ lse, int B, int L,
                                                                 const int64_t* __restrict__ len,
                                                                 const int* __restrict__ cu, int out16) {
  __shared__ __attribute__((aligned(16))) uint32_t sK[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sV[64 * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tq = lane & 15, tc = (lane >> 4) & 1;
  int bx, head, b;
  if (!attn_block(L, B, bx, head, b)) return;
  const int q0 = bx * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;     // packed rows (pack.hip): items back to back
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;                                      // rows this item owns in memory
  if (Lr <= 0) return;
  const int q = q0 + li, qc = q < Lr ? q : Lr - 1;
  // Query rows at or past the item's length are don't-care (every caller zeroes them after the following
  // LayerNorm, Layers.py:29): blocks made only of such rows write zeros and leave.
  if (bx * 128 >= klen) {
    if (q < Lr) {
      zero32(out, (rowbase + q) * 256 + head * AD + lh * 32, out16);
      if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = 0.f;
    }
    return;
  }

  // scores are kept in the log2 domain: q is pre-scaled by log2(e) / sqrt(d_k), so that p = exp2(s - m) is one
  // v_exp_f32 (no range fix-ups: p underflowing to zero is exactly what softmax wants)
  bf16x8 qf[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    if constexpr (Q16) qf[st] = load8_raw16(qkv, (rowbase + qc) * 768 + head * AD + st * 16 + lh * 8);
    else qf[st] = load8_bf16(qkv + (rowbase + qc) * 768 + head * AD + st * 16 + lh * 8, ATTN_SCALE_LOG2);
  }
  constexpr float SC = Q16 ? ATTN_SCALE_LOG2 : 1.f;    // Q16: the scale lives in the exponent's fma (see load8_raw16)

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const __amdgpu_buffer_rsrc_t krs = Q16 ? rows_rsrc16(qkv, rowbase * 768 + 256 + head * AD, 768, Lr)
                                          : rows_rsrc(qkv + rowbase * 768 + 256 + head * AD, 768, Lr);
  const __amdgpu_buffer_rsrc_t vrs = Q16 ? rows_rsrc16(qkv, rowbase * 768 + 512 + head * AD, 768, Lr)
                                          : rows_rsrc(qkv + rowbase * 768 + 512 + head * AD, 768, Lr);
  const int ntiles = (klen + 63) / 64;
  TileRegs<Q16> rk, rv;
  if (STYLER_ATTN_FWD_PREFETCH) { tile_load<Q16>(rk, krs, 768, 0, tid); tile_load<Q16>(rv, vrs, 768, 0, tid); }
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * 64;
    // (a register prefetch of the next tile across the MFMAs was measured in round 1: occupancy 3 -> 2 waves per SIMD
    // lost more than it hid.  The loads are issued ahead of the barrier instead: they fly while the block's other waves
    // finish the previous tile.)
    if (!STYLER_ATTN_FWD_PREFETCH) { tile_load<Q16>(rk, krs, 768, k0, tid); tile_load<Q16>(rv, vrs, 768, k0, tid); }
    __syncthreads();
    tile_store<Q16>(sK, rk, tid);
    tile_store<Q16>(sV, rv, tid);
    if (STYLER_ATTN_FWD_PREFETCH && kt + 1 < ntiles) { tile_load<Q16>(rk, krs, 768, k0 + 64, tid); tile_load<Q16>(rv, vrs, 768, k0 + 64, tid); }
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
      if constexpr (LAZY) {
        constexpr float TH = 8.f / SC;                 // 8 in the log2 domain, in units of the stored scores
        const bool move = mb > m_run + TH;             // (the first tile: m_run = -1e30 -> every lane moves)
        if (__builtin_amdgcn_ballot_w64(move) != 0) {
          const float m_new = move ? mb : m_run;
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * SC);
          m_run = m_new;
          l_run *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        const float mneg = -m_run * SC;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], SC, mneg)); rs += s[r]; }
        rs += __shfl_xor(rs, 32, 64);
        l_run += rs;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pb = pack_acc(s, s2);
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 0, tq, tc, kb, s2, lh), pb, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 1, tq, tc, kb, s2, lh), pb, o1, 0, 0, 0);
        }
        continue;
      }
      const float m_new = fmaxf(m_run, mb);
      const float alpha = __builtin_amdgcn_exp2f(Q16 ? (m_run - m_new) * SC : m_run - m_new);
      float rs = 0.f;
      if constexpr (Q16) {
        const float mneg = -m_new * SC;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], SC, mneg)); rs += s[r]; }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); rs += s[r]; }
      }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack_acc(s, s2);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 0, tq, tc, kb, s2, lh), pb, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 1, tq, tc, kb, s2, lh), pb, o1, 0, 0, 0);
      }
    }
  }
  if (q < Lr) {
    // out16: the attention output stored as bf16 (its readers -- the output projection, that projection's weight gradient
    // and the backward's delta -- take it rounded to bf16 or, the delta, accept it)
    if (out16) store_accT16(reinterpret_cast<uint16_t*>(out) + (rowbase + q) * 256 + head * AD, o0, o1, lh, 1.f / l_run);
    else store_accT(out + (rowbase + q) * 256 + head * AD, o0, o1, lh, 1.f / l_run);
    if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = (m_run * SC + log2f(l_run)) * 0.693147180559945f;   // natural log
  }
}

// Round 5: the LAZY forward with the softmax bookkeeping taken off the LDS pipe and out of the per-32-key dependency chain
// (VERDICT r04 item 8).  V is a bit set:
//   1  s_setprio 1 around the two MFMA clusters (a wave in its matrix phase wins the issue slot over its SIMD partners' VALU);
//   2  the two half-wave exchanges of a key block -- row maximum and row sum, each a ds_bpermute + s_waitcnt lgkmcnt(0) round
//      trip in the middle of the chain S -> max -> exp2 -> P V -- become one v_permlane32_swap (VALU, no wait) for the maximum
//      and NONE for the sum: both lane halves rescale by the same alpha, so each keeps the sum of its own 16 keys per block
//      and the halves are added once, after the last tile;
//   4  one softmax step per 64-key tile instead of per 32-key block: 8 independent S MFMAs in flight, one maximum / one
//      move test per tile, then 32 exp2 and 8 P V MFMAs;
//   8  the next tile's K / V rows are loaded into registers across the current tile's steps (16 more live registers).
__device__ __forceinline__ float xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <bool Q16, int V>
__global__ __launch_bounds__(256, STYLER_ATTN_FWD_WAVES) void attention_fwd_bf16_v_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 float* __restrict__ lse, int B, int L,
                                                                 const int64_t* __restrict__ len,
                                                                 const int* __restrict__ cu, int out16) {
  constexpr bool PRIO = V & 1, SWAP = V & 2, T64 = V & 4, PF = V & 8;
  __shared__ __attribute__((aligned(16))) uint32_t sK[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sV[64 * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tq = lane & 15, tc = (lane >> 4) & 1;
  int bx, head, b;
  if (!attn_block(L, B, bx, head, b)) return;
  const int q0 = bx * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;
  if (Lr <= 0) return;
  const int q = q0 + li, qc = q < Lr ? q : Lr - 1;
  if (bx * 128 >= klen) {
    if (q < Lr) {
      zero32(out, (rowbase + q) * 256 + head * AD + lh * 32, out16);
      if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = 0.f;
    }
    return;
  }
  bf16x8 qf[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    if constexpr (Q16) qf[st] = load8_raw16(qkv, (rowbase + qc) * 768 + head * AD + st * 16 + lh * 8);
    else qf[st] = load8_bf16(qkv + (rowbase + qc) * 768 + head * AD + st * 16 + lh * 8, ATTN_SCALE_LOG2);
  }
  constexpr float SC = Q16 ? ATTN_SCALE_LOG2 : 1.f;
  constexpr float TH = 8.f / SC;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;               // SWAP: l_run is this lane half's share of the row sum

  const __amdgpu_buffer_rsrc_t krs = Q16 ? rows_rsrc16(qkv, rowbase * 768 + 256 + head * AD, 768, Lr)
                                          : rows_rsrc(qkv + rowbase * 768 + 256 + head * AD, 768, Lr);
  const __amdgpu_buffer_rsrc_t vrs = Q16 ? rows_rsrc16(qkv, rowbase * 768 + 512 + head * AD, 768, Lr)
                                          : rows_rsrc(qkv + rowbase * 768 + 512 + head * AD, 768, Lr);
  const int ntiles = (klen + 63) / 64;
  TileRegs<Q16> rk, rv;
  // the tile's K and V rows -> LDS.  Loads are issued ahead of the barrier; PF: one tile ahead, across the tile's MFMAs
  auto stage = [&](int k0, int knext) {
    if constexpr (!PF) { tile_load<Q16>(rk, krs, 768, k0, tid); tile_load<Q16>(rv, vrs, 768, k0, tid); }
    __syncthreads();
    tile_store<Q16>(sK, rk, tid);
    tile_store<Q16>(sV, rv, tid);
    if constexpr (PF) if (knext >= 0) { tile_load<Q16>(rk, krs, 768, knext, tid); tile_load<Q16>(rv, vrs, 768, knext, tid); }
    __syncthreads();
  };
  if constexpr (PF) { tile_load<Q16>(rk, krs, 768, 0, tid); tile_load<Q16>(rv, vrs, 768, 0, tid); }
  // one softmax step over the 32-key block kb of the tile
  auto step32 = [&](int k0, int kb) {
    bf16x8 kf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) kf[st] = *reinterpret_cast<const bf16x8*>(&sK[(kb * 32 + li) * ALD + st * 8 + lh * 4]);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int st = 0; st < 4; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[st], s, 0, 0, 0);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    if (k0 + kb * 32 + 32 > klen) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= klen) s[r] = -INFINITY;
      }
    }
    float mb = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) mb = fmaxf(mb, s[r]);
    mb = SWAP ? xhalf_max(mb) : fmaxf(mb, __shfl_xor(mb, 32, 64));
    const bool move = mb > m_run + TH;
    if (__builtin_amdgcn_ballot_w64(move) != 0) {
      const float m_new = move ? mb : m_run;
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * SC);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    const float mneg = -m_run * SC;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], SC, mneg)); rs += s[r]; }
    if constexpr (!SWAP) rs += __shfl_xor(rs, 32, 64);
    l_run += rs;
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pb = pack_acc(s, s2);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 0, tq, tc, kb, s2, lh), pb, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 1, tq, tc, kb, s2, lh), pb, o1, 0, 0, 0);
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto step64 = [&](int k0) {
    // both key blocks in one step (all keys of block 0 are valid here: only block 1 can cross the length)
    bf16x8 kf[8];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      kf[st] = *reinterpret_cast<const bf16x8*>(&sK[li * ALD + st * 8 + lh * 4]);
      kf[4 + st] = *reinterpret_cast<const bf16x8*>(&sK[(32 + li) * ALD + st * 8 + lh * 4]);
    }
    f32x16 sa, sb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[st], sa, 0, 0, 0);
      sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[4 + st], qf[st], sb, 0, 0, 0);
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    if (k0 + 64 > klen) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= klen) sb[r] = -INFINITY;
      }
    }
    float mb = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) mb = fmaxf(mb, sa[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mb = fmaxf(mb, sb[r]);
    mb = SWAP ? xhalf_max(mb) : fmaxf(mb, __shfl_xor(mb, 32, 64));
    const bool move = mb > m_run + TH;
    if (__builtin_amdgcn_ballot_w64(move) != 0) {
      const float m_new = move ? mb : m_run;
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * SC);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    const float mneg = -m_run * SC;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = __builtin_amdgcn_exp2f(fmaf(sa[r], SC, mneg)); rs += sa[r]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { sb[r] = __builtin_amdgcn_exp2f(fmaf(sb[r], SC, mneg)); rs += sb[r]; }
    if constexpr (!SWAP) rs += __shfl_xor(rs, 32, 64);
    l_run += rs;
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pa = pack_acc(sa, s2);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 0, tq, tc, 0, s2, lh), pa, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 1, tq, tc, 0, s2, lh), pa, o1, 0, 0, 0);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pb = pack_acc(sb, s2);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 0, tq, tc, 1, s2, lh), pb, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sV, 1, tq, tc, 1, s2, lh), pb, o1, 0, 0, 0);
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  // tiles whose second key block holds data (k0 + 32 < klen), then -- klen % 64 in 1..32 -- a last tile of one block
  const int n2 = (klen + 31) >> 6;
  for (int kt = 0; kt < n2; ++kt) {
    stage(kt * 64, kt + 1 < ntiles ? kt * 64 + 64 : -1);
    if constexpr (T64) step64(kt * 64);
    else { step32(kt * 64, 0); step32(kt * 64, 1); }
  }
  if (n2 < ntiles) { stage(n2 * 64, -1); step32(n2 * 64, 0); }
  if constexpr (SWAP) l_run = xhalf_sum(l_run);
  if (q < Lr) {
    if (out16) store_accT16(reinterpret_cast<uint16_t*>(out) + (rowbase + q) * 256 + head * AD, o0, o1, lh, 1.f / l_run);
    else store_accT(out + (rowbase + q) * 256 + head * AD, o0, o1, lh, 1.f / l_run);
    if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = (m_run * SC + log2f(l_run)) * 0.693147180559945f;
  }
}

// ------------------------------------------------------------------------------------------------- dQ
template <bool Q16>
__global__ __launch_bounds__(256, STYLER_ATTN_DQ_WAVES) void attention_bwd_dq_bf16_kernel(const float* __restrict__ qkv,
                                                                    const float* __restrict__ o,
                                                                    const float* __restrict__ dout,
                                                                    const float* __restrict__ lse,
                                                                    void* __restrict__ dqkv, float* __restrict__ delta,
                                                                    int B, int L, const int64_t* __restrict__ len,
                                                                    const int* __restrict__ cu, int out16, int o16, int do16) {
  __shared__ __attribute__((aligned(16))) uint32_t sK[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sV[64 * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tq = lane & 15, tc = (lane >> 4) & 1;
  int bx, head, b;
  if (!attn_block(L, B, bx, head, b)) return;
  const int q0 = bx * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;
  if (Lr <= 0) return;
  const int q = q0 + li, qc = q < Lr ? q : Lr - 1;

  // query rows at or past the item's length carry no gradient (they are zeroed after the LayerNorm that follows):
  // blocks made only of such rows write dQ = 0, delta = 0 and leave
  if (bx * 128 >= klen) {
    if (q < Lr) {
      zero32(dqkv, (rowbase + q) * 768 + head * AD + lh * 32, out16);
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
    if constexpr (Q16) qf[st] = load8_raw16(qkv, (rowbase + qc) * 768 + off);    // (scale: in the exponent below)
    else qf[st] = load8_bf16(qkv + (rowbase + qc) * 768 + off, 0.125f * LOG2E);  // log2-domain scores, see forward
    // o16 / do16: the forward's output / the incoming gradient stored as bf16 (8 elements = one 16-byte load each)
    const int64_t ro = (rowbase + qc) * 256 + off;
    float dv8[8], ov8[8];
    if (do16) {
      const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(dout) + ro);
      dof[st] = *reinterpret_cast<const bf16x8*>(&r);
      const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { dv8[2 * e] = __uint_as_float(w[e] << 16); dv8[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    } else {
      const float* dp = dout + ro;
      dof[st] = load8_bf16(dp, 1.0f);
#pragma unroll
      for (int e = 0; e < 8; ++e) dv8[e] = dp[e];
    }
    if (o16) {
      const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(o) + ro);
      const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { ov8[2 * e] = __uint_as_float(w[e] << 16); ov8[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    } else {
      const float* op = o + ro;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov8[e] = op[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += dv8[e] * ov8[e];
  }
  dl += __shfl_xor(dl, 32, 64);
  const float my_lse = lse[((int64_t)b * 4 + head) * L + qc] * LOG2E;
  if (q < Lr && lh == 0) delta[((int64_t)b * 4 + head) * L + q] = dl;

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const __amdgpu_buffer_rsrc_t krs = Q16 ? rows_rsrc16(qkv, rowbase * 768 + 256 + head * AD, 768, Lr)
                                          : rows_rsrc(qkv + rowbase * 768 + 256 + head * AD, 768, Lr);
  const __amdgpu_buffer_rsrc_t vrs = Q16 ? rows_rsrc16(qkv, rowbase * 768 + 512 + head * AD, 768, Lr)
                                          : rows_rsrc(qkv + rowbase * 768 + 512 + head * AD, 768, Lr);
  const int ntiles = (klen + 63) / 64;
  TileRegs<Q16> rk, rv;
  if (STYLER_ATTN_DQ_PREFETCH) { tile_load<Q16>(rk, krs, 768, 0, tid); tile_load<Q16>(rv, vrs, 768, 0, tid); }
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * 64;
    if (!STYLER_ATTN_DQ_PREFETCH) { tile_load<Q16>(rk, krs, 768, k0, tid); tile_load<Q16>(rv, vrs, 768, k0, tid); }
    __syncthreads();
    tile_store<Q16>(sK, rk, tid);
    tile_store<Q16>(sV, rv, tid);
    if (STYLER_ATTN_DQ_PREFETCH && kt + 1 < ntiles) { tile_load<Q16>(rk, krs, 768, k0 + 64, tid); tile_load<Q16>(rv, vrs, 768, k0 + 64, tid); }
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
      for (int r = 0; r < 16; ++r)
        s[r] = __builtin_amdgcn_exp2f(Q16 ? fmaf(s[r], 0.125f * LOG2E, -my_lse) : s[r] - my_lse) * (dp[r] - dl);
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
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sK, 0, tq, tc, kb, s2, lh), dsb, dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sK, 1, tq, tc, kb, s2, lh), dsb, dq1, 0, 0, 0);
      }
    }
  }
  if (q < Lr) {
    const int64_t off = (rowbase + q) * 768 + head * AD;
    if (out16) store_accT16(reinterpret_cast<uint16_t*>(dqkv) + off, dq0, dq1, lh, 0.125f);
    else store_accT(reinterpret_cast<float*>(dqkv) + off, dq0, dq1, lh, 0.125f);
  }
}

// ------------------------------------------------------------------------------------------------- dK, dV
template <bool Q16, bool DO16>
__global__ __launch_bounds__(256, STYLER_ATTN_DKV_WAVES) void attention_bwd_dkv_bf16_kernel(const float* __restrict__ qkv,
                                                                     const float* __restrict__ dout,
                                                                     const float* __restrict__ lse,
                                                                     const float* __restrict__ delta,
                                                                     void* __restrict__ dqkv, int B, int L,
                                                                     const int64_t* __restrict__ len,
                                                                     const int* __restrict__ cu, int out16) {
  __shared__ __attribute__((aligned(16))) uint32_t sQ[64 * ALD];
  __shared__ __attribute__((aligned(16))) uint32_t sDO[64 * ALD];
  __shared__ __attribute__((aligned(16))) float sLse[64], sDl[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tq = lane & 15, tc = (lane >> 4) & 1;
  int bx, head, b;
  if (!attn_block(L, B, bx, head, b)) return;
  const int key0 = bx * 128 + wave * 32;
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
    if constexpr (Q16) {
      kf[st] = load8_raw16(qkv, (rowbase + keyc) * 768 + 256 + head * AD + st * 16 + lh * 8);
      vf[st] = load8_raw16(qkv, (rowbase + keyc) * 768 + 512 + head * AD + st * 16 + lh * 8);
    } else {
      kf[st] = load8_bf16(qkv + (rowbase + keyc) * 768 + 256 + head * AD + st * 16 + lh * 8, 0.125f * LOG2E);
      vf[st] = load8_bf16(qkv + (rowbase + keyc) * 768 + 512 + head * AD + st * 16 + lh * 8, 1.0f);
    }
  }
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }

  // A lane owns one key column, so an invalid key only pollutes its own dK / dV, which are written as zeros below.
  // Query rows at or past klen have dO = 0 and delta = 0 (no gradient reaches them): they add nothing to any dK / dV,
  // so the query loop stops at klen; rows of the last tile past the item are fetched as zeros (lse = +huge there).
  const __amdgpu_buffer_rsrc_t qrs = Q16 ? rows_rsrc16(qkv, rowbase * 768 + head * AD, 768, Lr)
                                          : rows_rsrc(qkv + rowbase * 768 + head * AD, 768, Lr);
  const __amdgpu_buffer_rsrc_t drs = DO16 ? rows_rsrc16(dout, rowbase * 256 + head * AD, 256, Lr)
                                           : rows_rsrc(dout + rowbase * 256 + head * AD, 256, Lr);
  const float* lse_row = lse + ((int64_t)b * 4 + head) * L;
  const float* dl_row = delta + ((int64_t)b * 4 + head) * L;
  const bool block_live = bx * 128 < klen;
  const int ntiles = block_live ? (klen + 63) / 64 : 0;
  TileRegs<Q16> rq;
  TileRegs<DO16> rdo;
  float r_lse = 0.f, r_dl = 0.f;                       // tid < 64: row tid of the tile
  auto fetch = [&](int qb) {
    if (tid < 64) {                                    // issued first: their wait must not cover the tile loads below
      const int qq = qb + tid, qi = qq < klen ? qq : klen - 1;
      r_lse = lse_row[qi]; r_dl = dl_row[qi];
    }
    tile_load<Q16>(rq, qrs, 768, qb, tid);
    tile_load<DO16>(rdo, drs, 256, qb, tid);
  };
#ifndef STYLER_ATTN_DKV_PREFETCH                       // 1: next tile's loads in registers across the MFMAs (no better at 2-3 waves
#define STYLER_ATTN_DKV_PREFETCH 0                     // per SIMD: 84.4 vs 80.7 us), 0: loads issued ahead of the barrier
#endif
  if (STYLER_ATTN_DKV_PREFETCH && ntiles > 0) fetch(0);
  for (int qt = 0; qt < ntiles; ++qt) {
    const int qb = qt * 64;
    if (!STYLER_ATTN_DKV_PREFETCH) fetch(qb);          // (variant without the register prefetch: loads ahead of the barrier)
    __syncthreads();
    tile_store<Q16>(sQ, rq, tid);
    tile_store<DO16>(sDO, rdo, tid);
    if (tid < 64) {
      // rows at or past klen: lse = +huge makes p exactly 0 (whatever dO / the forward's lse hold there)
      const bool okq = qb + tid < klen;
      sLse[tid] = okq ? r_lse * LOG2E : 1e30f;
      sDl[tid] = okq ? r_dl : 0.f;
    }
    if (STYLER_ATTN_DKV_PREFETCH && qt + 1 < ntiles) fetch(qb + 64);   // next tile: in flight while this one is multiplied
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
      for (int g = 0; g < 4; ++g) {                    // the lane's 16 query rows are four runs of 4: two ds_read_b128 per run
        const int ql = qk * 32 + 8 * g + 4 * lh;
        const float4 l4 = *reinterpret_cast<const float4*>(&sLse[ql]);
        const float4 d4 = *reinterpret_cast<const float4*>(&sDl[ql]);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = g * 4 + e;
          const float p = __builtin_amdgcn_exp2f(Q16 ? fmaf(s[r], 0.125f * LOG2E, -lv[e]) : s[r] - lv[e]);
          s[r] = p;
          dp[r] = p * (dp[r] - dvv[e]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pb = pack_acc(s, s2), dsb = pack_acc(dp, s2);
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sDO, 0, tq, tc, qk, s2, lh), pb, dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sDO, 1, tq, tc, qk, s2, lh), pb, dv1, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sQ, 0, tq, tc, qk, s2, lh), dsb, dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragT(sQ, 1, tq, tc, qk, s2, lh), dsb, dk1, 0, 0, 0);
      }
    }
  }
  if (key < Lr) {
    // dK = dS^T (Q / sqrt(d_k)): the staged Q tiles are unscaled, so the 1/8 is applied here; invalid keys get zeros
    // (their accumulators may hold anything, inf included: written as literal zeros, never multiplied by 0)
    const int64_t off = (rowbase + key) * 768 + head * AD;
    if (key_ok) {
      if (out16) {
        store_accT16(reinterpret_cast<uint16_t*>(dqkv) + off + 256, dk0, dk1, lh, 0.125f);
        store_accT16(reinterpret_cast<uint16_t*>(dqkv) + off + 512, dv0, dv1, lh, 1.0f);
      } else {
        store_accT(reinterpret_cast<float*>(dqkv) + off + 256, dk0, dk1, lh, 0.125f);
        store_accT(reinterpret_cast<float*>(dqkv) + off + 512, dv0, dv1, lh, 1.0f);
      }
    } else {
      zero32(dqkv, off + 256 + lh * 32, out16);
      zero32(dqkv, off + 512 + lh * 32, out16);
    }
  }
}

// ================================================================================================= bf16x3
// The bf16x3 arithmetic (STYLER_PREC_BF16X3) for the three kernels above: every MFMA operand is carried as
// hi + lo (hi = bf16(v), lo = bf16(v - hi): 16 mantissa bits) and every product a b runs as a_hi b_hi + a_hi b_lo + a_lo b_hi
// on v_mfma_f32_32x32x16_bf16 into the same fp32 accumulator.  Storage is fp32 on both sides; a staged tile becomes two
// bf16 LDS tiles (hi and lo), a register operand two fragments, P / dS are split while they are packed.  Softmax, lse and
// delta stay fp32 as in every mode.  Same decomposition, masks and early exits as the bf16 kernels.
__device__ __forceinline__ void split_pk(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = cvtpk(a, b);
  lo = cvtpk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
// 8 consecutive floats at p (scaled) -> hi / lo bf16x8
__device__ __forceinline__ void load8_x3(const float* p, float scale, bf16x8& hi, bf16x8& lo) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  uint4 h, l;
  split_pk(a.x * scale, a.y * scale, h.x, l.x);
  split_pk(a.z * scale, a.w * scale, h.y, l.y);
  split_pk(b.x * scale, b.y * scale, h.z, l.z);
  split_pk(b.z * scale, b.w * scale, h.w, l.w);
  hi = *reinterpret_cast<bf16x8*>(&h);
  lo = *reinterpret_cast<bf16x8*>(&l);
}
__device__ __forceinline__ void pack_acc_x3(const f32x16& a, int s2, bf16x8& hi, bf16x8& lo) {
  uint4 h, l;
  split_pk(a[s2 * 8 + 0], a[s2 * 8 + 1], h.x, l.x);
  split_pk(a[s2 * 8 + 2], a[s2 * 8 + 3], h.y, l.y);
  split_pk(a[s2 * 8 + 4], a[s2 * 8 + 5], h.z, l.z);
  split_pk(a[s2 * 8 + 6], a[s2 * 8 + 7], h.w, l.w);
  hi = *reinterpret_cast<bf16x8*>(&h);
  lo = *reinterpret_cast<bf16x8*>(&l);
}
// a staged fp32 tile (load_rows) -> the hi and the lo bf16 [64][ALD] tiles
__device__ __forceinline__ void store_rows_x3(uint32_t* dh, uint32_t* dl, const float4 (&v)[4], int tid) {
  const int o = (tid >> 4) * ALD + (tid & 15) * 2;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    uint2 h, l;
    split_pk(v[p].x, v[p].y, h.x, l.x);
    split_pk(v[p].z, v[p].w, h.y, l.y);
    *reinterpret_cast<uint2*>(&dh[o + p * 16 * ALD]) = h;
    *reinterpret_cast<uint2*>(&dl[o + p * 16 * ALD]) = l;
  }
}
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// c += a b with both operands split (three products, the lo lo one dropped)
#define MFMA_X3(ah, al, bh, bl, c) { c = MFMA_BF16(ah, bh, c); c = MFMA_BF16(ah, bl, c); c = MFMA_BF16(al, bh, c); }

__global__ __launch_bounds__(256, 2) void attention_fwd_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                  float* __restrict__ lse, int B, int L,
                                                                  const int64_t* __restrict__ len, const int* __restrict__ cu,
                                                                  uint16_t* __restrict__ y3) {
  // y3 (round 5, styler_set_x3_out): the [hi(256) | lo(256)] split of the output rows, written next to them
  __shared__ __attribute__((aligned(16))) uint32_t sKh[64 * ALD], sKl[64 * ALD], sVh[64 * ALD], sVl[64 * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tq = lane & 15, tc = (lane >> 4) & 1;
  int bx, head, b;
  if (!attn_block(L, B, bx, head, b)) return;
  const int q0 = bx * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;
  if (Lr <= 0) return;
  const int q = q0 + li, qc = q < Lr ? q : Lr - 1;
  if (bx * 128 >= klen) {
    if (q < Lr) {
      zero32(out, (rowbase + q) * 256 + head * AD + lh * 32, 0);
      if (y3) zero32_x3(y3 + (rowbase + q) * 512 + head * AD + lh * 32, 256);
      if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = 0.f;
    }
    return;
  }
  bf16x8 qh[4], ql[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) load8_x3(qkv + (rowbase + qc) * 768 + head * AD + st * 16 + lh * 8, ATTN_SCALE_LOG2, qh[st], ql[st]);
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  const __amdgpu_buffer_rsrc_t krs = rows_rsrc(qkv + rowbase * 768 + 256 + head * AD, 768, Lr);
  const __amdgpu_buffer_rsrc_t vrs = rows_rsrc(qkv + rowbase * 768 + 512 + head * AD, 768, Lr);
  const int ntiles = (klen + 63) / 64;
  float4 rk[4], rv[4];
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * 64;
    load_rows(rk, krs, 768, k0, tid);
    load_rows(rv, vrs, 768, k0, tid);
    __syncthreads();
    store_rows_x3(sKh, sKl, rk, tid);
    store_rows_x3(sVh, sVl, rv, tid);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (k0 + kb * 32 >= klen) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int o = (kb * 32 + li) * ALD + st * 8 + lh * 4;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(&sKh[o]);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(&sKl[o]);
        MFMA_X3(kh, kl, qh[st], ql[st], s);
      }
      if (k0 + kb * 32 + 32 > klen) {
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
        bf16x8 ph, pl;
        pack_acc_x3(s, s2, ph, pl);
        { const bf16x8 vh = fragT(sVh, 0, tq, tc, kb, s2, lh), vl = fragT(sVl, 0, tq, tc, kb, s2, lh); MFMA_X3(vh, vl, ph, pl, o0); }
        { const bf16x8 vh = fragT(sVh, 1, tq, tc, kb, s2, lh), vl = fragT(sVl, 1, tq, tc, kb, s2, lh); MFMA_X3(vh, vl, ph, pl, o1); }
      }
    }
  }
  if (q < Lr) {
    store_accT(out + (rowbase + q) * 256 + head * AD, o0, o1, lh, 1.f / l_run);
    if (y3) store_accT_x3(y3 + (rowbase + q) * 512 + head * AD, 256, o0, o1, lh, 1.f / l_run);
    if (lse && lh == 0) lse[((int64_t)b * 4 + head) * L + q] = (m_run + log2f(l_run)) * 0.693147180559945f;
  }
}

__global__ __launch_bounds__(256, 2) void attention_bwd_dq_x3_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                                     const float* __restrict__ dout, const float* __restrict__ lse,
                                                                     float* __restrict__ dqkv, float* __restrict__ delta,
                                                                     int B, int L, const int64_t* __restrict__ len,
                                                                     const int* __restrict__ cu, uint16_t* __restrict__ y3) {
  // y3 (round 5, styler_set_x3_out): the [hi(768) | lo(768)] split of the dqkv rows (this kernel: the dq third)
  __shared__ __attribute__((aligned(16))) uint32_t sKh[64 * ALD], sKl[64 * ALD], sVh[64 * ALD], sVl[64 * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tq = lane & 15, tc = (lane >> 4) & 1;
  int bx, head, b;
  if (!attn_block(L, B, bx, head, b)) return;
  const int q0 = bx * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;
  if (Lr <= 0) return;
  const int q = q0 + li, qc = q < Lr ? q : Lr - 1;
  if (bx * 128 >= klen) {
    if (q < Lr) {
      zero32(dqkv, (rowbase + q) * 768 + head * AD + lh * 32, 0);
      if (y3) zero32_x3(y3 + (rowbase + q) * 1536 + head * AD + lh * 32, 768);
      if (lh == 0) delta[((int64_t)b * 4 + head) * L + q] = 0.f;
    }
    return;
  }
  constexpr float LOG2E = 1.44269504088896f;
  bf16x8 qh[4], ql[4], dh[4], dlo[4];
  float dl = 0.f;
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int off = head * AD + st * 16 + lh * 8;
    load8_x3(qkv + (rowbase + qc) * 768 + off, 0.125f * LOG2E, qh[st], ql[st]);
    const float* dp = dout + (rowbase + qc) * 256 + off;
    const float* op = o + (rowbase + qc) * 256 + off;
    load8_x3(dp, 1.0f, dh[st], dlo[st]);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += dp[e] * op[e];
  }
  dl += __shfl_xor(dl, 32, 64);
  const float my_lse = lse[((int64_t)b * 4 + head) * L + qc] * LOG2E;
  if (q < Lr && lh == 0) delta[((int64_t)b * 4 + head) * L + q] = dl;

  f32x16 dq0, dq1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const __amdgpu_buffer_rsrc_t krs = rows_rsrc(qkv + rowbase * 768 + 256 + head * AD, 768, Lr);
  const __amdgpu_buffer_rsrc_t vrs = rows_rsrc(qkv + rowbase * 768 + 512 + head * AD, 768, Lr);
  const int ntiles = (klen + 63) / 64;
  float4 rk[4], rv[4];
  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * 64;
    load_rows(rk, krs, 768, k0, tid);
    load_rows(rv, vrs, 768, k0, tid);
    __syncthreads();
    store_rows_x3(sKh, sKl, rk, tid);
    store_rows_x3(sVh, sVl, rv, tid);
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (k0 + kb * 32 >= klen) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int of = (kb * 32 + li) * ALD + st * 8 + lh * 4;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(&sKh[of]), kl = *reinterpret_cast<const bf16x8*>(&sKl[of]);
        const bf16x8 vh = *reinterpret_cast<const bf16x8*>(&sVh[of]), vl = *reinterpret_cast<const bf16x8*>(&sVl[of]);
        MFMA_X3(kh, kl, qh[st], ql[st], s);
        MFMA_X3(vh, vl, dh[st], dlo[st], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - my_lse) * (dp[r] - dl);
      if (k0 + kb * 32 + 32 > klen) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= klen) s[r] = 0.f;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 sh, sl;
        pack_acc_x3(s, s2, sh, sl);
        { const bf16x8 kh = fragT(sKh, 0, tq, tc, kb, s2, lh), kl = fragT(sKl, 0, tq, tc, kb, s2, lh); MFMA_X3(kh, kl, sh, sl, dq0); }
        { const bf16x8 kh = fragT(sKh, 1, tq, tc, kb, s2, lh), kl = fragT(sKl, 1, tq, tc, kb, s2, lh); MFMA_X3(kh, kl, sh, sl, dq1); }
      }
    }
  }
  if (q < Lr) {
    store_accT(dqkv + (rowbase + q) * 768 + head * AD, dq0, dq1, lh, 0.125f);
    if (y3) store_accT_x3(y3 + (rowbase + q) * 1536 + head * AD, 768, dq0, dq1, lh, 0.125f);
  }
}

__global__ __launch_bounds__(256, 2) void attention_bwd_dkv_x3_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                                      float* __restrict__ dqkv, int B, int L,
                                                                      const int64_t* __restrict__ len, const int* __restrict__ cu,
                                                                      uint16_t* __restrict__ y3) {
  __shared__ __attribute__((aligned(16))) uint32_t sQh[64 * ALD], sQl[64 * ALD], sDh[64 * ALD], sDl2[64 * ALD];
  __shared__ __attribute__((aligned(16))) float sLse[64], sDl[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tq = lane & 15, tc = (lane >> 4) & 1;
  int bx, head, b;
  if (!attn_block(L, B, bx, head, b)) return;
  const int key0 = bx * 128 + wave * 32;
  const int64_t rowbase = cu ? (int64_t)cu[b] : (int64_t)b * L;
  int klen = len ? (int)len[b] : L;
  if (klen > L) klen = L;
  const int Lr = cu ? klen : L;
  if (Lr <= 0) return;
  const int key = key0 + li, keyc = key < Lr ? key : Lr - 1;
  const bool key_ok = key < klen;
  constexpr float LOG2E = 1.44269504088896f;
  bf16x8 kh[4], kl[4], vh[4], vl[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    load8_x3(qkv + (rowbase + keyc) * 768 + 256 + head * AD + st * 16 + lh * 8, 0.125f * LOG2E, kh[st], kl[st]);
    load8_x3(qkv + (rowbase + keyc) * 768 + 512 + head * AD + st * 16 + lh * 8, 1.0f, vh[st], vl[st]);
  }
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  const __amdgpu_buffer_rsrc_t qrs = rows_rsrc(qkv + rowbase * 768 + head * AD, 768, Lr);
  const __amdgpu_buffer_rsrc_t drs = rows_rsrc(dout + rowbase * 256 + head * AD, 256, Lr);
  const float* lse_row = lse + ((int64_t)b * 4 + head) * L;
  const float* dl_row = delta + ((int64_t)b * 4 + head) * L;
  const int ntiles = bx * 128 < klen ? (klen + 63) / 64 : 0;
  float4 rq[4], rdo[4];
  float r_lse = 0.f, r_dl = 0.f;
  for (int qt = 0; qt < ntiles; ++qt) {
    const int qb = qt * 64;
    if (tid < 64) {
      const int qq = qb + tid, qi = qq < klen ? qq : klen - 1;
      r_lse = lse_row[qi]; r_dl = dl_row[qi];
    }
    load_rows(rq, qrs, 768, qb, tid);
    load_rows(rdo, drs, 256, qb, tid);
    __syncthreads();
    store_rows_x3(sQh, sQl, rq, tid);
    store_rows_x3(sDh, sDl2, rdo, tid);
    if (tid < 64) {
      const bool okq = qb + tid < klen;
      sLse[tid] = okq ? r_lse * LOG2E : 1e30f;
      sDl[tid] = okq ? r_dl : 0.f;
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
        const int of = (qk * 32 + li) * ALD + st * 8 + lh * 4;
        const bf16x8 qvh = *reinterpret_cast<const bf16x8*>(&sQh[of]), qvl = *reinterpret_cast<const bf16x8*>(&sQl[of]);
        const bf16x8 dvh = *reinterpret_cast<const bf16x8*>(&sDh[of]), dvl = *reinterpret_cast<const bf16x8*>(&sDl2[of]);
        MFMA_X3(qvh, qvl, kh[st], kl[st], s);
        MFMA_X3(dvh, dvl, vh[st], vl[st], dp);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ql_ = qk * 32 + 8 * g + 4 * lh;
        const float4 l4 = *reinterpret_cast<const float4*>(&sLse[ql_]);
        const float4 d4 = *reinterpret_cast<const float4*>(&sDl[ql_]);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = g * 4 + e;
          const float p = __builtin_amdgcn_exp2f(s[r] - lv[e]);
          s[r] = p;
          dp[r] = p * (dp[r] - dvv[e]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 ph, pl, sh, sl;
        pack_acc_x3(s, s2, ph, pl);
        pack_acc_x3(dp, s2, sh, sl);
        { const bf16x8 ah = fragT(sDh, 0, tq, tc, qk, s2, lh), al = fragT(sDl2, 0, tq, tc, qk, s2, lh); MFMA_X3(ah, al, ph, pl, dv0); }
        { const bf16x8 ah = fragT(sDh, 1, tq, tc, qk, s2, lh), al = fragT(sDl2, 1, tq, tc, qk, s2, lh); MFMA_X3(ah, al, ph, pl, dv1); }
        { const bf16x8 ah = fragT(sQh, 0, tq, tc, qk, s2, lh), al = fragT(sQl, 0, tq, tc, qk, s2, lh); MFMA_X3(ah, al, sh, sl, dk0); }
        { const bf16x8 ah = fragT(sQh, 1, tq, tc, qk, s2, lh), al = fragT(sQl, 1, tq, tc, qk, s2, lh); MFMA_X3(ah, al, sh, sl, dk1); }
      }
    }
  }
  if (key < Lr) {
    const int64_t off = (rowbase + key) * 768 + head * AD;
    if (key_ok) {
      store_accT(dqkv + off + 256, dk0, dk1, lh, 0.125f);
      store_accT(dqkv + off + 512, dv0, dv1, lh, 1.0f);
    } else {
      zero32(dqkv, off + 256 + lh * 32, 0);
      zero32(dqkv, off + 512 + lh * 32, 0);
    }
    if (y3) {                                          // round 5: the split of the dk / dv thirds of the row
      uint16_t* const o3 = y3 + (rowbase + key) * 1536 + head * AD;
      if (key_ok) {
        store_accT_x3(o3 + 256, 768, dk0, dk1, lh, 0.125f);
        store_accT_x3(o3 + 512, 768, dv0, dv1, lh, 1.0f);
      } else {
        zero32_x3(o3 + 256 + lh * 32, 768);
        zero32_x3(o3 + 512 + lh * 32, 768);
      }
    }
  }
}

// fp32 tensors on both sides; same arguments as styler_attention_fwd / styler_attention_bwd
extern "C" int styler_attention_fwd_x3(const float* qkv, float* out, float* lse, int B, int L, const int64_t* len,
                                       const int32_t* cu, void* stream) {
  uint16_t* y3 = nullptr;
  int y3parts = 0;
  styler_take_x3_out(&y3, &y3parts);               // (the [hi | lo] split of the output rows, for the out-projection GEMM)
  if (y3 && (y3parts != 2 || ((uintptr_t)y3 & 15))) return STYLER_EINVAL;
  if (!qkv || !out || B <= 0 || L <= 0 || (cu && !len)) return STYLER_EINVAL;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return STYLER_EALIGN;
  hipLaunchKernelGGL(attention_fwd_x3_kernel, attn_grid(L, B), dim3(256), 0, (hipStream_t)stream, qkv, out, lse, B, L, len, cu, y3);
  return launch_status();
}
extern "C" int styler_attention_bwd_x3(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                       float* delta_ws, int B, int L, const int64_t* len, const int32_t* cu, void* stream) {
  uint16_t* y3 = nullptr;
  int y3parts = 0;
  styler_take_x3_out(&y3, &y3parts);               // (the [hi | lo] split of the dqkv rows, for the QKV dX GEMM and weight gradients)
  if (y3 && (y3parts != 2 || ((uintptr_t)y3 & 15))) return STYLER_EINVAL;
  if (!qkv || !out || !dout || !lse || !dqkv || !delta_ws || B <= 0 || L <= 0) return STYLER_EINVAL;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) return STYLER_EALIGN;
  const dim3 grid = attn_grid(L, B);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attention_bwd_dq_x3_kernel, grid, dim3(256), 0, st, qkv, out, dout, lse, dqkv, delta_ws, B, L, len, cu, y3);
  hipLaunchKernelGGL(attention_bwd_dkv_x3_kernel, grid, dim3(256), 0, st, qkv, dout, lse, delta_ws, dqkv, B, L, len, cu, y3);
  return launch_status();
}

// io_flags & STYLER_IO_X_BF16: qkv is stored as bf16 ([rows][768] elements; throughput mode writes it that way from the QKV
// GEMM's epilogue -- its only readers are these three kernels, which round it to bf16 anyway).  io_flags & STYLER_IO_Y_BF16:
// `out` is written as bf16.
extern "C" int styler_attention_fwd_bf16_io(const void* qkv, void* out, float* lse, int B, int L, const int64_t* len,
                                            const int32_t* cu, int io_flags, void* stream) {
  if (!qkv || !out || B <= 0 || L <= 0 || (cu && !len)) return STYLER_EINVAL;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return STYLER_EALIGN;
  const float* q = reinterpret_cast<const float*>(qkv);
  float* o = reinterpret_cast<float*>(out);
  const int out16 = (io_flags & STYLER_IO_Y_BF16) ? 1 : 0;
  // STYLER_ATTN_LAZY=0: the eager-rescaling form of the forward (see the kernel's LAZY note)
  static const int lazy = [] { const char* e = getenv("STYLER_ATTN_LAZY"); return e ? atoi(e) : 1; }();
#define FWD_LAUNCH(Q_, L_) hipLaunchKernelGGL((attention_fwd_bf16_kernel<Q_, L_>), attn_grid(L, B), dim3(256), 0, (hipStream_t)stream, q, o, lse, B, L, len, cu, out16)
  // the round-5 form of the lazy forward (attention_fwd_bf16_v_kernel), variant 7 = setprio + permlane swap / deferred row sum
  // + 64-key softmax step: 888-901 -> 854 us on the config-4 launch, 19.0 -> 17.6 us on the config-3 decoder's, same box
  // (profiles/r05_attn_variants.txt, which also has the variants not compiled in any more: 1, 2, 4 alone ~1 %; 14 = 6 +
  // register prefetch 869; a V tile at a row stride of 48 dwords -- conflict-free ds_read_b64_tr_b16 -- no change).
  // STYLER_ATTN_FWD_V=0: the round-4 kernel; 3, 6, 15: 7 without the 64-key step / without setprio / with the prefetch.
  static const int fv = [] { const char* e = getenv("STYLER_ATTN_FWD_V"); return e ? atoi(e) : 7; }();
#define FWDV_LAUNCH(Q_, V_) hipLaunchKernelGGL((attention_fwd_bf16_v_kernel<Q_, V_>), attn_grid(L, B), dim3(256), 0, (hipStream_t)stream, q, o, lse, B, L, len, cu, out16)
#define FWDV_CASE(V_) case V_: if (io_flags & STYLER_IO_X_BF16) FWDV_LAUNCH(true, V_); else FWDV_LAUNCH(false, V_); return launch_status();
  if (lazy && fv > 0) {
    switch (fv) { FWDV_CASE(3) FWDV_CASE(6) FWDV_CASE(7) FWDV_CASE(15) default: break; }
  }
#undef FWDV_CASE
#undef FWDV_LAUNCH
  if (io_flags & STYLER_IO_X_BF16) { if (lazy) FWD_LAUNCH(true, true); else FWD_LAUNCH(true, false); }
  else { if (lazy) FWD_LAUNCH(false, true); else FWD_LAUNCH(false, false); }
#undef FWD_LAUNCH
  return launch_status();
}

extern "C" int styler_attention_fwd_bf16(const float* qkv, float* out, float* lse, int B, int L, const int64_t* len,
                                         const int32_t* cu, void* stream) {
  return styler_attention_fwd_bf16_io(qkv, out, lse, B, L, len, cu, 0, stream);
}

// io_flags: STYLER_IO_X_BF16 = qkv stored as bf16, STYLER_IO_Y_BF16 = dqkv written as bf16, STYLER_IO_MASK_BF16 = `out` (the
// forward's output) stored as bf16, STYLER_IO_RES_BF16 = `dout` stored as bf16 (pass the bf16 pointers in their places).
extern "C" int styler_attention_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse,
                                         void* dqkv, float* delta_ws, int B, int L, const int64_t* len,
                                         const int32_t* cu, int io_flags, void* stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || !delta_ws || B <= 0 || L <= 0) return STYLER_EINVAL;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) return STYLER_EALIGN;
  const dim3 grid = attn_grid(L, B);
  hipStream_t st = (hipStream_t)stream;
  const int out16 = (io_flags & STYLER_IO_Y_BF16) ? 1 : 0;
  const int o16 = (io_flags & STYLER_IO_MASK_BF16) ? 1 : 0, do16 = (io_flags & STYLER_IO_RES_BF16) ? 1 : 0;
  const bool q16 = (io_flags & STYLER_IO_X_BF16) != 0;
  if (q16) hipLaunchKernelGGL(attention_bwd_dq_bf16_kernel<true>, grid, dim3(256), 0, st, qkv, out, dout, lse, dqkv, delta_ws, B, L, len, cu, out16, o16, do16);
  else hipLaunchKernelGGL(attention_bwd_dq_bf16_kernel<false>, grid, dim3(256), 0, st, qkv, out, dout, lse, dqkv, delta_ws, B, L, len, cu, out16, o16, do16);
#define DKV_LAUNCH(Q_, D_) hipLaunchKernelGGL((attention_bwd_dkv_bf16_kernel<Q_, D_>), grid, dim3(256), 0, st, qkv, dout, lse, delta_ws, dqkv, B, L, len, cu, out16)
  if (q16 && do16) DKV_LAUNCH(true, true);
  else if (q16) DKV_LAUNCH(true, false);
  else if (do16) DKV_LAUNCH(false, true);
  else DKV_LAUNCH(false, false);
#undef DKV_LAUNCH
  return launch_status();
}
