// Argument block shared by the two MFMA GEMM / implicit-conv engines (gemm_conv.hip: 128x128 / 64x64 tiles, register-staged
// operands; gemm256.hip: 256x256 tile, eight waves, LDS-DMA operand loads).
#pragma once
#include "common.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

struct GemmArgs {
  const void* x; int64_t ldx;                      // fp32, or bf16 with the A16 kernels (ldx in elements either way)
  const void* w;
  const float* scale; const float* shift;
  const float* res; int64_t ldres;
  void* y; int64_t ldy;                            // fp32, or bf16 with the Y16 kernels
  int B, L, cin, n, kw, act, pad;
  const int64_t* len;
  int mt, nt;                                      // tile counts
  const int2* rowinfo;                             // packed rows (styler_pack_plan): (t, len - 1 - t) per row, or null
  const void* mask; int64_t ldmask;                // epilogue: v = mask[row, col] > 0 ? v : 0 (ReLU backward), or null
  int mask16;                                      // the mask tensor is bf16
  uint64_t* trace;                                 // styler_gemm_set_trace: 8 words per block (phase timestamps), or null
  int ksplit = 1;                                  // gemm256.hip: split-K factor of the launch (1 or 2)
  float* part = nullptr;                           // ... and its fp32 partial tiles [ksplit][B*L][n] (styler_gemm_set_workspace)
  int* cnt = nullptr;                              // round 6, split-K = 2 finished IN the kernel: 2 ints per tile (ticket, flag), zero on entry and
                                                   // on exit; `part` then holds ONE raw accumulator image per tile (styler_gemm_set_counters)
  int res16 = 0;                                   // the residual tensor is bf16 (STYLER_IO_RES_BF16; ldres in elements)
  int x3n1 = 0;                                    // STYLER_IO_X3A: x rows hold [hi | lo] of cin / 3 channels each; x3n1 = (cin / 3) / 64 chunks per part,
                                                   // channel chunk cc >= 2 * x3n1 (the third product) reads chunk cc - 2 * x3n1 (hi again)
  uint16_t* y3 = nullptr;                          // round 5 (styler_set_x3_out): ALSO store the bf16x3 split of the fp32 output,
  int y3parts = 0;                                 // rows of y3parts * n bf16: [hi | lo (| hi)] (fp32 outputs only)
};

// gemm256.hip: returns 1 when the 256x256 LDS-DMA engine takes the launch (and has enqueued it), 0 when the shape is not
// eligible, < 0 on a launch error.
// `ws` / `ws_bytes`: the workspace registered for this call (styler_gemm_take_workspace), needed by its split-K path.
int styler_gemm256_try(const GemmArgs& a, int x16, int y16, hipStream_t st, void* ws, int64_t ws_bytes);
// hands over (and clears) the workspace the host thread registered with styler_gemm_set_workspace
void styler_gemm_take_workspace(void** ws, int64_t* bytes);
// gemm_conv.hip: split-K factor of a small-M, long-K launch on the 64 x 64 tile (1: none)
int styler_gemm_small_ksplit(int B, int L, int cin, int n, int kw, int act, bool has_mask);
// gemm256.hip: the combine pass of a split-K launch (a.part holds ks partial tiles [B*L][n]; epilogue from a)
int styler_gemm_combine(const GemmArgs& a, int ks, int y16, hipStream_t st);
