// Shared device helpers for the STYLER gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/styler_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define WAVE 64

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// Wave-wide sum without the LDS crossbar: the xor butterfly 32, 16, 8, 4, 2, 1 as gfx950's half / row swaps
// (v_permlane32_swap, v_permlane16_swap: with both operands = v the two results are v and its partner half / row) and
// four DPP adds (row_ror:8, row_ror:4 -- after the first three steps a value depends on lane & 7 only, so the rotation IS
// the xor -- and the two quad permutes).  Same partners in the same order as six `v += __shfl_xor(v, o)` steps, i.e. the
// same bits in every lane, at six VALU-rate instructions instead of six dependent ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef STYLER_WAVE_SUM_SHFL                            // A/B builds only: the ds_bpermute butterfly
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(h[0]) + __uint_as_float(h[1]);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  v += dpp_f32<0x128>(v);                              // row_ror:8
  v += dpp_f32<0x124>(v);                              // row_ror:4
  v += dpp_f32<0x4E>(v);                               // quad_perm [2,3,0,1]
  v += dpp_f32<0xB1>(v);                               // quad_perm [1,0,3,2]
  return v;
#endif
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// fp32 -> bf16 bits, round to nearest even (NaN kept quiet)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
// Two floats -> packed bf16 pair (round-to-nearest-even), one v_cvt_pk_bf16_f32.  Written as a vector conversion, NOT
// inline asm: the compiler must see the instruction to place the wait state the hardware needs between a
// transcendental (v_exp_f32) and a consumer of its result -- an asm block right behind exp2 read garbage.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16_rne(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<const uint32_t*>(&b);
}

// Four consecutive elements at element index `idx` of a tensor stored as fp32 or (IS16) bf16, as a raw load and a separate
// conversion: loops issue the raw loads of a batch first and convert afterwards (a conversion next to its load puts a wait
// behind every load).
template <bool IS16> struct Raw4 { typedef float4 T; };
template <> struct Raw4<true> { typedef uint2 T; };
template <bool IS16>
__device__ __forceinline__ typename Raw4<IS16>::T raw4_load(const void* base, int64_t idx) {
  if constexpr (IS16) return *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
  else return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
}
// Pin a batch of loaded values: an empty asm that "modifies" them sits between the load loop and the consume loop, so the
// compiler cannot sink a load to its use (it does that to shorten live ranges, and then waits behind every load).
__device__ __forceinline__ void pin_loaded(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void pin_loaded(uint2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ float4 raw4_f32(const float4& v) { return v; }
__device__ __forceinline__ float4 raw4_f32(const uint2& u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}

// Four consecutive elements at element index `idx` of a tensor stored as fp32 or (is16) bf16 -- load as float4 / store a
// float4 (round to nearest even).  For the row-wise kernels whose tensors may live in either format (round 3: the decoder's
// residual stream is bf16 in throughput mode); `is16` is uniform, the branch costs nothing next to the memory access.
__device__ __forceinline__ float4 ldg4(const void* base, int64_t idx, bool is16) {
  if (is16) return raw4_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx));
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
}
__device__ __forceinline__ void stg4(void* base, int64_t idx, const float4& v, bool is16) {
  if (is16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + idx) = make_uint2(cvt_pk_bf16_rne(v.x, v.y), cvt_pk_bf16_rne(v.z, v.w));
  else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = v;
}

// bf16x3 producer-side split (round 5): the (hi, lo) bf16 pairs of four consecutive fp32 values, bit for bit what
// split3_bf16_kernel (misc.hip) writes for them -- hi = bf16_rne(v), lo = bf16_rne(v - float(hi)).  Producers (GEMM epilogues,
// norms, attention) store the split next to their fp32 output, so no separate pass re-reads it (1.5 ms of the 21 ms step).
__device__ __forceinline__ void x3_split4(const float4 v, uint2& hi, uint2& lo) {
  const uint32_t h01 = cvt_pk_bf16_rne(v.x, v.y), h23 = cvt_pk_bf16_rne(v.z, v.w);
  hi = make_uint2(h01, h23);
  lo = make_uint2(cvt_pk_bf16_rne(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u)),
                  cvt_pk_bf16_rne(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u)));
}
// ... and their store into row `row` of a split tensor [rows, parts * C] (parts = 2: [hi | lo]; 3: [hi | lo | hi]), channel c
__device__ __forceinline__ void x3_store4(uint16_t* __restrict__ y3, int64_t row, int c, int C, int parts, const float4 v) {
  uint2 hi, lo;
  x3_split4(v, hi, lo);
  uint16_t* yr = y3 + row * (int64_t)(parts * C) + c;
  *reinterpret_cast<uint2*>(yr) = hi;
  *reinterpret_cast<uint2*>(yr + C) = lo;
  if (parts == 3) *reinterpret_cast<uint2*>(yr + 2 * C) = hi;
}

// hands over (and clears) the split output the host thread registered with styler_set_x3_out (misc.hip) for its NEXT producer
// call: every entry point that can fill one takes it first thing, so a registration never leaks to a later call
void styler_take_x3_out(uint16_t** y3, int* parts);

// v - float(bf16(v)): the low part of the hi + lo split of the bf16x3 arithmetic (exact in fp32)
__device__ __forceinline__ float bf16_lo_part(float v) { return v - __uint_as_float(f32_to_bf16_bits(v) << 16); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
}

// tanh(x) = sign(x) (1 - t) / (1 + t), t = exp2(-2 |x| log2 e): one v_exp_f32 + one v_rcp_f32 instead of the library's
// branchy ~30-instruction tanhf (the BatchNorm + tanh kernels of the PostNet evaluate it for every element in forward, and
// twice more in backward).  Absolute error vs double tanh <= 2.5e-7 over [-12, 12] and around 0
// (tests/test_hip_parity.py::test_fast_tanh_accuracy holds that bound on the device).
__device__ __forceinline__ float fast_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(x));
  const float r = (1.f - t) * __builtin_amdgcn_rcpf(1.f + t);
  return copysignf(r, x);
}

// sigmoid(x) = 1 / (1 + exp2(-x log2 e)): v_exp_f32 + v_rcp_f32 (expf + an IEEE division are ~25 instructions; the LSTM
// kernels are bound by their VALU instruction count).  Absolute error <= 3e-7.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == STYLER_ACT_RELU) return fmaxf(v, 0.f);
  if (act == STYLER_ACT_TANH) return fast_tanh(v);
  if (act == STYLER_ACT_LOGCLAMP) return logf(fmaxf(v, 1e-5f));
  if (act == STYLER_ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
  if (act == STYLER_ACT_CRELU) return fminf(fmaxf(v, 0.f), 20.f);
  return v;
}

// Device address of a step counter that is mixed into every dropout seed (styler_set_dropout_counter), or null.
// Seeds are host values baked into a captured hipGraph; the counter (incremented on the device once per step) is
// what makes every replay draw fresh masks.  Forward and backward of one step read the same value.
extern const uint64_t* g_styler_drop_epoch;
__device__ __forceinline__ uint64_t mix_drop_epoch(uint64_t seed, const uint64_t* epoch) {
  return epoch ? seed + *epoch * 0xD6E8FEB86659FD93ull : seed;
}

// Counter-based dropout stream: keep element e of a tensor iff dropout_hash32(seed, e) >= p * 2^32.  A keyed 32-bit
// mixer: the key schedule (splitmix64 of the seed) is uniform -- scalar ALU, hoisted out of element loops -- and the
// per-element part is two 32-bit multiplies with the key entering before the first and between the two (the 64-bit
// splitmix per element it replaces was ~50 issue slots per element: the BatchNorm backward kernels were VALU-bound on it).
__device__ __forceinline__ uint2 dropout_key(uint64_t seed) {
  uint64_t k = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
  k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
  k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
  k ^= k >> 31;
  return make_uint2((uint32_t)k, (uint32_t)(k >> 32));
}
// the per-element part with the key (and, where the caller has it as a scalar, the high index word) already in hand
__device__ __forceinline__ uint32_t dropout_hash32_keyed(uint2 key, uint32_t idx_lo, uint32_t idx_hi) {
  uint32_t h = idx_lo ^ key.x;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15;
  h += key.y + idx_hi * 0x9E3779B1u;
  h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t dropout_hash32(uint64_t seed, uint64_t idx) {
  return dropout_hash32_keyed(dropout_key(seed), (uint32_t)idx, (uint32_t)(idx >> 32));
}

// Round 6 -- THE dropout stream of every kernel: one keyed hash per GROUP of four consecutive elements.  Every site that
// draws masks handles elements e .. e + 3 (e % 4 == 0) in one lane, and the per-element mixer above was the largest single
// item of the row kernels' instruction count (layernorm_bwd: 80 of 175 per row pair; the BatchNorm backward was VALU-bound
// on it).  The group's first element index goes through the same mixer up to the last multiply; two finishers give 64 bits
// = four 16-bit draws; element j is kept iff draw_j >= p * 2^16 (p = 0.1: 6554 / 65536 = 0.100006 -- the keep probability
// differs from 1 - p by 6e-6, below fp32 resolution of the 1 / (1 - p) scale).  14 integer instructions per four elements
// instead of 40.  Forward and backward of a site regenerate the same draws from (seed, step counter, element index).
__device__ __forceinline__ uint2 dropout_word4(uint2 key, uint32_t e_lo, uint32_t e_hi) {
  uint32_t h = e_lo ^ key.x;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15;
  h += key.y + e_hi * 0x9E3779B1u;
  uint32_t a = h * 0x846ca68bu, b = (h ^ 0x68E31DA4u) * 0xB5297A4Du;
  a ^= a >> 16; b ^= b >> 15;
  return make_uint2(a, b);
}
__device__ __forceinline__ uint32_t dropout_thr16(float p) { return (uint32_t)(p * 65536.f + 0.5f); }
// keep * scale (or 0) for the four elements of a group
__device__ __forceinline__ float4 dropout_scale4(uint2 w, uint32_t thr16, float sc) {
  return make_float4((w.x & 0xffffu) >= thr16 ? sc : 0.f, (w.x >> 16) >= thr16 ? sc : 0.f,
                     (w.y & 0xffffu) >= thr16 ? sc : 0.f, (w.y >> 16) >= thr16 ? sc : 0.f);
}
// v -> dropout(v) for the four elements of a group: SELECTS, not products with a 0 / scale factor -- a consumer that forms
// `v - float(bf16(v))` (the bf16x3 split) would otherwise contract the product into an fma and see an unrounded v
__device__ __forceinline__ float4 dropout_select4(float4 v, uint2 w, uint32_t thr16, float sc) {
  return make_float4((w.x & 0xffffu) >= thr16 ? v.x * sc : 0.f, (w.x >> 16) >= thr16 ? v.y * sc : 0.f,
                     (w.y & 0xffffu) >= thr16 ? v.z * sc : 0.f, (w.y >> 16) >= thr16 ? v.w * sc : 0.f);
}
// ... for elements e .. e + 3 of the stream `key` (e % 4 == 0)
__device__ __forceinline__ float4 dropout_apply4(float4 v, uint2 key, uint64_t e, uint32_t thr16, float sc) {
  return dropout_select4(v, dropout_word4(key, (uint32_t)e, (uint32_t)(e >> 32)), thr16, sc);
}

// Gradient w.r.t. the BatchNorm output of y = dropout(act(BN(x))) for one element: the dropout keep mask is regenerated
// from (seed, element index) by the caller, the tanh output is recomputed from x (gamma/beta/mean/rstd) unless `yv` supplies it.
// `keep_sc`: the element's dropout factor (0 or 1 / (1 - p); 1 without dropout), from dropout_scale4 of its group.
__device__ __forceinline__ float bn_dz_elem(float g, float xh, float ga, float be, int act, bool has_y, float yv, float keep_sc) {
  g *= keep_sc;
  if (act == STYLER_ACT_TANH) {
    const float o = has_y ? yv : fast_tanh(xh * ga + be);
    g *= 1.f - o * o;
  }
  return g;
}

