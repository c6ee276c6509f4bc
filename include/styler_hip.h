/* styler_hip.h -- C ABI of libstyler_hip.so: the MI355X (gfx950) kernels behind the
 * STYLER forward/backward hot path.
 *
 * The reference (keonlee9420/STYLER) is pure Python on stock PyTorch ops; it has no FFI.
 * Each entry point below replaces the PyTorch op sequence at the cited reference
 * file:line (paths relative to the reference root).  The host side
 * (the styler_amd Python package) binds these with ctypes and mirrors the reference's nn.Module API.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless named host_*.  No torch types cross this ABI.
 *  - Activations are fp32, channels-last [B, L, C], row stride `ld*` in ELEMENTS (so an
 *    op can read or write a channel slice of a wider buffer: no torch.cat / split copies).
 *  - Lengths are int64 [B] as in the reference (src_len / mel_len); masks are derived in
 *    the kernels (position t of item b is padding iff t >= len[b]).
 *  - `stream` is a hipStream_t (passed as void*); every call only enqueues work.
 *  - Return value: 0 on success, a negative STYLER_E* code on argument errors, or a
 *    positive hipError_t from the launch.
 */
#ifndef STYLER_HIP_H
#define STYLER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STYLER_EINVAL (-1)
#define STYLER_EALIGN (-2)

/* activation codes for fused epilogues */
#define STYLER_ACT_NONE 0
#define STYLER_ACT_RELU 1
#define STYLER_ACT_TANH 2
#define STYLER_ACT_LOGCLAMP 3 /* log(max(v, 1e-5)): dynamic_range_compression, audio_processing.py:80-86 */
#define STYLER_ACT_LEAKY 4    /* v > 0 ? v : 0.1 v: LRELU_SLOPE of the vocoder, hifigan/models.py:7,97 */
#define STYLER_ACT_CRELU 5    /* min(max(v, 0), 20): DeepSpeaker's clipped ReLU, deepspeaker/conv_models.py:83-86 */
/* OR-ed into `act`: the residual input joins BEFORE scale / shift / activation, y = act(scale * (acc + res) + shift)
 * -- the last call of a convolution that is summed over several GEMM calls (2-D taps of the ResCNN, conv_models.py) */
#define STYLER_ACT_RES_FIRST 0x100

/* io_flags of styler_conv_gemm[_packed] / styler_wgrad[_packed] (throughput mode only): the tensor behind the float
 * pointer is bf16 (row strides stay in ELEMENTS).  Used for the FFN hidden activation and its gradient, which are only
 * ever consumed as bf16 MFMA operands or as a sign mask. */
#define STYLER_IO_X_BF16 1    /* conv_gemm: x;   wgrad: x  */
#define STYLER_IO_Y_BF16 2    /* conv_gemm: y (no residual);   wgrad: dz */
#define STYLER_IO_MASK_BF16 4 /* conv_gemm: mask */
#define STYLER_IO_RES_BF16 8  /* conv_gemm: res (then allowed together with STYLER_IO_Y_BF16);  pack / unpack: see there */
#define STYLER_IO_Z_BF16 16   /* GroupNorm / BatchNorm forward and backward: the normalised tensor x (a convolution's output,
                                 kept for the backward) is stored as bf16 -- throughput mode; the statistics are those of the
                                 rounded values, forward and backward agree on them */
/* styler_wgrad / styler_wgrad_packed / styler_wgrad_group_desc only (the bf16x3 arithmetic on fp32-typed operands): the
 * kernel stages the LOW part of the operand, bf16(v - float(bf16(v))), instead of its rounding to bf16 (the high part);
 * ignored for an operand stored as bf16.  dw += dz_hi^T x_lo and dw += dz_lo^T x_hi without materialising the low parts. */
#define STYLER_IO_X_LO 32
#define STYLER_IO_DZ_LO 64
/* the same entry points, both operands bf16 split tensors [hi | lo (| hi)] (styler_split3_bf16: row strides >= 2 n / 2 cin,
 * STYLER_IO_X_BF16 | STYLER_IO_Y_BF16 set): dw += dz_hi^T x_hi + dz_hi^T x_lo + dz_lo^T x_hi as ONE launch whose contraction axis
 * runs over the three parts (one set of split-K partial tiles instead of three); db += colsum(dz_hi) + colsum(dz_lo).
 * LDS-DMA kernels only: styler_wgrad_x3cat_ok(n, cin, kw, pad_left) says whether the shape has one. */
#define STYLER_IO_X3CAT 128
/* the same entry points with a deferred reduce (defer_reduce = 1 / grouped members): `db` is not the bias gradient but a
 * [splits][n] fp32 slot array (splits = styler_wgrad_splits_io of the call) into which split s STORES its column sums of dz;
 * the caller folds the slots into the bias gradient(s) in split order with styler_wgrad_reduce_multi (a descriptor with
 * cin = kw = 1).  No atomics: the bias gradients of nn.Linear / nn.Conv1d are bit-reproducible.  db2 must be NULL. */
#define STYLER_IO_DB_SLOTS 256
/* styler_groupnorm_relu_bwd / styler_aug_classifier_tail_bwd_io: the parameter-gradient pointers are SLOT arrays that the
 * launch stores its per-block sums into (GroupNorm: [B][C] for gamma and for beta, single-pass kernel only); the caller folds
 * them in slot order with styler_wgrad_reduce_multi.  No fp32 atomics: the gradients are bit-reproducible. */
#define STYLER_IO_PARAM_SLOTS 512
/* styler_conv_gemm / styler_conv_gemm_packed (bf16 MFMA mode, STYLER_IO_X_BF16 set, cin = 3 C with C % 64 == 0): x is the COMPACT
 * bf16x3 split [hi | lo] of C channels each (styler_split3_bf16 with parts = 2, row stride >= 2 C); the channel chunks of
 * the third product (cin / 3 * 2 .. cin) read the hi block a second time.  The weight row stays [w_hi | w_hi | w_lo]. */
#define STYLER_IO_X3A 1024
/* styler_add_layernorm io_flags (round 3: the decoder's residual stream is stored as bf16 in throughput mode) */
#define STYLER_LN_RES_BF16 1  /* res is bf16 */
#define STYLER_LN_Y_BF16 2    /* y is written as bf16 (ldy in elements) */
#define STYLER_LN_SUM_BF16 4  /* sum_out is written as bf16 -- and the row is normalised from that rounded sum, which is what
                               * styler_layernorm_bwd recomputes the statistics from */

/* arithmetic of the MFMA GEMM core */
#define STYLER_PREC_F32  0  /* v_mfma_f32_32x32x2_f32: exact fp32 (parity mode)            */
#define STYLER_PREC_BF16 1  /* v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate    */

int styler_abi_version(void);

/* ---- GEMM / Conv1d-as-implicit-GEMM -------------------------------------------------
 * y[b,t,n] = act( scale[n] * (sum_{j<kw} sum_{c<cin} x[b, t+j-kw/2, c] * w[n, j*cin+c])
 *                 + shift[n] ) (+ res[b,t,n])          x = 0 outside 0 <= t+j-kw/2 < L
 * Replaces nn.Linear (kw=1: SubLayers.py:41-43,58; styler.py:31; modules.py:211-216,
 * 250-271) and nn.Conv1d with 'same' padding (SubLayers.py:72-76; modules.py:103-161,
 * 444-453; Layers.py:78-118).  `w` is [n, kw*cin] row-major, i.e. nn.Linear layout, or
 * nn.Conv1d weight [n, cin, kw] permuted to [n, kw, cin].  With prec = BF16, `w` points
 * to bf16 (uint16) data of the same layout.  scale may be NULL (=1); shift is the bias
 * (or the folded BatchNorm shift, Layers.py:91); res may be NULL.  cin % 4 == 0,
 * ldx % 4 == 0 and 16-byte aligned x / w are required.
 * If len != NULL, output rows with t >= len[b] are written as 0 (masked_fill).
 * mask (optional, [B,L,n] with row stride ldmask): the value is zeroed where mask <= 0, BEFORE the residual
 * add -- the dX GEMM of a layer that feeds a ReLU takes the ReLU's forward output here, so
 * dx = (dy W^T) * relu'(h) + res is one launch (autograd of SubLayers.py:86-89). */
int styler_conv_gemm(const float* x, int64_t ldx, const void* w, const float* scale,
                     const float* shift, const float* res, int64_t ldres, float* y,
                     int64_t ldy, int B, int L, int cin, int n, int kw, int act, int prec,
                     const int64_t* len, const float* mask, int64_t ldmask, int io_flags, void* stream);

/* ---- packed rows (the decoder runs on the valid frames only) --------------------------
 * Every FFT block of the decoder (transformer/Models.py:111-135) zeroes its padded rows
 * (Layers.py:29,32), masks padded keys and convolves over zeros there; on the padded rectangle
 * [B, T] that is arithmetic on zeros for every padded frame.  The packed layout stores the valid
 * rows of all items back to back (capacity B*T rows; the valid count stays on the device).
 * styler_pack_plan derives the index tables from the int64 lengths (clamped to [0, T]):
 *   cu [B+1] int32: first packed row of each item;  rowinfo [B*T] x (int32 t, int32 len-1-t),
 *   (0,-1) behind the data;  chunktab [B*T/64 + B] x int32[4] = (first row, t0, len, b) of the
 *   64-row K chunks of the weight-gradient GEMM;  counts [2] int64 = (valid rows, chunks).
 * counts doubles as the `len` of the single packed "item" for the row-wise entry points
 * (styler_add_layernorm, styler_layernorm_bwd, styler_act_bwd, styler_conv_gemm with B = 1). */
int styler_pack_plan(const int64_t* len, int B, int T, int32_t* cu, int32_t* rowinfo,
                     int32_t* chunktab, int64_t* counts, void* stream);
/* packed[cu[b]+t] = padded[b,t] (+ add[t,:], e.g. the positional table of Models.py:120-125), t < len[b].
 * io_flags (both directions): STYLER_IO_X_BF16 = the PADDED tensor is bf16, STYLER_IO_Y_BF16 = the PACKED tensor is bf16. */
int styler_pack_rows(const float* padded, int64_t ldp, float* packed, int64_t ldk, const float* add,
                     const int32_t* cu, int B, int T, int C, int io_flags, void* stream);
/* padded[b,t] = t < len[b] ? packed[cu[b]+t] : 0   (also the backward of styler_pack_rows) */
int styler_unpack_rows(const float* packed, int64_t ldk, float* padded, int64_t ldp,
                       const int32_t* cu, int B, int T, int C, int io_flags, void* stream);
/* styler_conv_gemm on packed rows: taps never cross an item (rowinfo), tiles behind the data are
 * skipped, rows >= nrows[0] are written as 0. */
int styler_conv_gemm_packed(const float* x, int64_t ldx, const void* w, const float* scale,
                            const float* shift, const float* res, int64_t ldres, float* y,
                            int64_t ldy, int rows, int cin, int n, int kw, int act, int prec,
                            const int64_t* nrows, const int32_t* rowinfo, const float* mask,
                            int64_t ldmask, int io_flags, void* stream);

/* styler_conv_gemm with an explicit left padding and any kw <= 9 (even allowed):
 *   y[b,t,:] = act(scale * sum_j x[b, t + j - pad, :] w[:, j, :] + shift) (+ res),  rows outside [0, L) read as 0.
 * With row strides ldx = d * C (and B = 1) on the d phase views of a sequence this is a dilated Conv1d
 * (hifigan/models.py:27-55); an 11-tap conv is two 6-tap calls (pad 5, then pad 0 accumulating through `res`); a
 * ConvTranspose1d(k = 2u, stride u, padding u/2) is the 3-tap conv (pad 1) whose n axis is (phase, c_out)
 * (hifigan/models.py:128-137). */
int styler_conv_gemm_pad(const float* x, int64_t ldx, const void* w, const float* scale,
                         const float* shift, const float* res, int64_t ldres, float* y, int64_t ldy,
                         int B, int L, int cin, int n, int kw, int pad, int act, int prec,
                         void* stream);

/* Which tile engine styler_conv_gemm dispatches for a shape: bit0 = 128x128 block tile (else
 * 64x64), bit1 = bf16 MFMA (else fp32 MFMA).  Used by bench.py to attribute launches. */
int styler_conv_gemm_variant(int B, int L, int cin, int n, int kw, int prec);
/* The engine a styler_conv_gemm call with these arguments runs on: 0..3 = styler_conv_gemm_variant, 4 = the
 * 256 x 256 eight-wave LDS-DMA engine (csrc/gemm256.hip: bf16 MFMA mode, x stored as bf16 -- io_flags &
 * STYLER_IO_X_BF16 --, cin % 64 == 0, at least 8 K steps and 1.5 tiles of 256 x 256 per CU; `packed` != 0: the call is
 * styler_conv_gemm_packed, whose row count is a capacity).  Same arithmetic either way: both
 * engines accumulate the same v_mfma_f32_32x32x16_bf16 sequence, results are bit-equal. */
/* Up to 8 INDEPENDENT small GEMMs in one launch (bf16 MFMA, 64 x 64 tile, k = 1, fp32 activations in / out; bf16 weight
 * shadows [n, cin], cin % 8 == 0): y_p = act(scale_p * x_p w_p^T + shift_p) (+ res_p), as styler_conv_gemm computes each.
 * The nn.Linear launches of the S-domain that do not depend on one another (modules.py:179-182 -- the four BiLSTMs' input
 * projections; 250-271, 335-348 -- the style MLPs; 23-45 -- the classifiers' first Linear).
 * `flags` (the same for every member of a launch): bit 0 = x is bf16 (ldx in elements, ldx % 8 == 0); bit 1 = x is the
 * compact bf16x3 split [hi | lo] (cin = 3 C, C % 64 == 0: STYLER_IO_X3A) -- the grouped launches of the bf16x3 arithmetic. */
typedef struct StylerGemmProblem {
  const void* x; const void* w; const void* scale; const void* shift; const void* res; void* y; const void* len;
  int64_t ldx, ldres, ldy;
  int32_t B, L, cin, n, act, flags;
} StylerGemmProblem;
int styler_conv_gemm_group(const StylerGemmProblem* probs, int count, void* stream);
int styler_conv_gemm_engine(int B, int L, int cin, int n, int kw, int prec, int io_flags, int64_t ldx, int packed);
/* ... with the epilogue inputs that decide whether the 256 x 256 engine runs the launch as split-K = 2 (plain epilogue, no
 * ReLU mask): 4 for those launches as well. */
int styler_conv_gemm_engine2(int B, int L, int cin, int n, int kw, int prec, int io_flags, int64_t ldx, int packed,
                             int act, int has_mask);
/* Test-only policy overrides of that engine: split_mode 0 = policy, 1 = never split-K, 2 = split-K wherever the epilogue
 * allows it; take_all 1 = no tile bound and no short-K guard (-1 keeps a value).  Returns the previous pair as
 * split_mode | take_all << 2.  (The tile bound of styler_gemm256_config is only a bound.) */
int styler_gemm256_policy(int split_mode, int take_all);
/* Test / tuning hook of that engine: enabled (0 / 1) and the smallest tile count it takes; -1 keeps a value (defaults:
 * STYLER_GEMM256, STYLER_GEMM256_MIN_TILES or 1, 384).  Returns the previous state as enabled | min_tiles << 1. */
int styler_gemm256_config(int enabled, int min_tiles);
/* Round 6: the engine's tile HEIGHT.  ht = 4: 256-row tiles, ht = 3: 192-row tiles (same LDS image and phase schedule,
 * three MFMA row tiles per wave row), ht = 0: the dispatch policy picks per launch (rounds of 256 CUs x tile height);
 * min_tiles3 = the smallest count of 192 x 256 tiles the policy gives to the 192-row tile (0: never; -1 keeps; defaults
 * STYLER_GEMM256_HT or 0, STYLER_GEMM256_MIN_TILES3 or 300).  Any other ht keeps the value.  Returns the previous state
 * as ht | min_tiles3 << 3.  Same results bit for bit at either height (tests/test_10_hip_parity.py::test_gemm256_*).
 * Replaces: nothing in the reference (a scheduling knob of nn.Conv1d / nn.Linear forward + dX, Layers.py:78-118). */
int styler_gemm256_height(int ht, int min_tiles3);
/* The narrow-output tile of styler_conv_gemm (bf16 MFMA mode, 64 < n <= 96 -- the 80 mel channels of PostNet's last
 * convolution, of the dX of its first one and of mel_linear, Layers.py:78-118, styler.py:22): one 128 x 96 block tile per
 * row block instead of two 64 x 64 tiles.  enabled (0 / 1) and the smallest row count B * L it takes; -1 keeps a value
 * (defaults: STYLER_GEMM_N96 or 1, 32768).  Returns the previous state as enabled | min_rows << 1.  Same MFMA sequence per
 * output element as the other tiles: results are bit-equal. */
int styler_gemm_n96_config(int enabled, int min_rows);
/* Split-K scratch of the 256 x 256 engine.  styler_conv_gemm_workspace_bytes: bytes of fp32 partial tiles a
 * styler_conv_gemm / styler_conv_gemm_packed call with these arguments wants (0: none; today only launches with 96..160
 * data-carrying tiles and >= 64 K steps and a plain epilogue: the dX of the FFN's k = 9 convolution,
 * transformer/SubLayers.py:72-76).  styler_gemm_set_workspace registers [ptr, ptr + bytes) for the NEXT conv_gemm call of
 * the calling host thread (consumed by it); without a registration the call runs unsplit. */
int64_t styler_conv_gemm_workspace_bytes(int B, int L, int cin, int n, int kw, int act, int prec, int io_flags,
                                         int64_t ldx, int packed, int has_mask);
int styler_gemm_set_workspace(void* ptr, int64_t bytes);
/* Round 6: `count` int32 counters, ZERO on entry (2 per 256 x 256 tile of the call: 2 * ceil(B L / 256) * (n / 256)), registered
 * for the next styler_conv_gemm / styler_conv_gemm_packed call of this host thread next to its workspace.  A split-K = 2 launch
 * of the 256 x 256 engine is then finished INSIDE the kernel -- the half that arrives second adds the first one's accumulators
 * (one raw image per tile in the workspace) and runs the epilogue: no partial rows in HBM, no combine launch.  fp32 addition is
 * commutative, so the result does not depend on the arrival order: same bits launch after launch, and the same bits as the
 * combine pass when there is no `scale`.  The kernel leaves the counters zero (a buffer can be re-used by later calls in stream
 * order, never by two calls that may run concurrently).  Without counters (or with STYLER_GEMM256_FIXUP=0 /
 * styler_gemm256_fixup(0)) the launch uses the combine pass. */
int styler_gemm_set_counters(void* ptr, int64_t count);
int styler_gemm256_fixup(int enabled);
/* Round 5 (bf16x3 arithmetic): producers write the operand split themselves.  The next PRODUCER call of this host thread --
 * styler_conv_gemm / styler_conv_gemm_packed (fp32 output in bf16 MFMA mode; whatever engine takes it, including its split-K
 * combine pass), styler_add_layernorm, styler_layernorm_bwd (its dx), styler_groupnorm_relu / _bwd, styler_batchnorm_train /
 * _bwd, styler_attention_fwd_x3 (its output) / styler_attention_bwd_x3 (dqkv) (fp32 outputs, contiguous rows) -- ALSO stores the [hi | lo (| hi)] bf16 split of its fp32 output rows into y3: rows of parts * C bf16, parts = 2 | 3,
 * the layout and the values of styler_split3_bf16 bit for bit, so that the GEMM consuming the output as a bf16x3 operand
 * needs no split pass (transformer/SubLayers.py:41-61,83-89, transformer/Layers.py:91-128, modules.py:103-172 and their
 * autograd).  Consumed by that call; a producer that cannot honour it returns STYLER_EINVAL.  y3 = NULL clears it. */
int styler_set_x3_out(void* y3, int parts);
/* Split-K of the 64 x 64 tile (bf16 MFMA mode): a k >= 3 convolution over few rows and a long contraction axis -- at most
 * 256 tiles, >= 48 (chunk, tap) steps, plain epilogue: the dX of the text encoder's FFN convolution, M = B * S rows,
 * K = 9 * 1024 (transformer/SubLayers.py:72-76, Models.py:60-84) -- deals its 64-channel chunks to about four blocks per CU;
 * the partial tiles (same workspace protocol as above) are folded in a fixed order by the combine pass.  enabled 0 / 1
 * (-1 keeps; default STYLER_GEMM_SMALL_SPLITK or 1); returns the previous value. */
int styler_gemm_small_split_config(int enabled);
/* Measurement hook (tools/gemm_trace.py): while `buf` is non-null every styler_conv_gemm block writes 8 uint64 words at
 * buf[8 * blockIdx]: block, then 100 MHz timestamps at entry / first tile staged / main loop done / stores issued /
 * stores acknowledged, the hardware id register and the tile index.  Pass NULL to switch it off (the default). */
int styler_gemm_set_trace(void* buf);

/* fp32 -> bf16 (round-to-nearest-even) weight shadow for STYLER_PREC_BF16 */
int styler_cast_bf16(const float* src, uint16_t* dst, int64_t count, void* stream);
/* bf16 -> fp32 (exact): the way back of the optional bf16 gradient all-reduce (the data-parallel replacement of
 * train.py:33's nn.DataParallel reduce; STYLER_ALLREDUCE_BF16=1). */
int styler_cast_from_bf16(const uint16_t* src, float* dst, int64_t count, void* stream);

/* Many strided 3-D copies (fp32 source -> fp32 | bf16 destination) in one launch: the refresh of every derived
 * weight layout after an optimiser step.  Descriptor i owns blocks [block_start[i], block_start[i+1]) of 1024
 * elements; element (a0,a1,a2) of dims (d0,d1,d2): dst[a0*ds0+a1*ds1+a2*ds2] = src[a0*ss0+a1*ss1+a2*ss2]
 * (+ src2[same index] when src2 != 0); strides in elements, may be negative with src pre-offset.
 * flags: bit0 = bf16 destination; bit1 = tiled transpose: the caller asserts ds2 == 1, ss0 == d1, ss1 == +1 or -1 (src
 * pre-offset to the element (0, 0, 0) as always) and ss2 = the row stride of the source: the descriptor then owns
 * ceil(d2 / 64) * ceil(d0*d1 / 64) blocks, each moving a [64 a2] x [64 merged (a0, a1)] tile through LDS (the transposed
 * dX weight layouts: coalesced on both sides); bit2 = tap interleave: ds2 == 1, ss1 == 1, ss2 == d1 <= 9 (the [n, kw, cin]
 * conv layout from [n, cin, kw]): ceil(d0 / 2) * ceil(d2 / 128) blocks, each moving 128 a2 values x d1 taps of two a0. */
typedef struct {
  uint64_t src, src2, dst;
  int64_t ss0, ss1, ss2, ds0, ds1, ds2, block_start;
  int32_t d0, d1, d2, flags;
} StylerCopyDesc;
int styler_strided_copy_multi(const StylerCopyDesc* desc_dev, int count, int64_t total_blocks, void* stream);
/* Round 6: the same launch with the owner of every block given (blockmap[b] = index of the descriptor block b belongs to, device
 * int32 [total_blocks]) instead of searched for by every block. */
int styler_strided_copy_multi_map(const StylerCopyDesc* desc_dev, int count, int64_t total_blocks, const int32_t* blockmap,
                                  void* stream);

/* Conv1d weight repack [n, cin, kw] <-> [n, kw, cin] (state-dict layout <-> kernel layout);
 * out_bf16 != 0 writes the bf16 shadow directly (dst is uint16). */
int styler_repack_conv_weight(const float* src, void* dst, int n, int cin, int kw,
                              int to_kernel_layout, int out_bf16, void* stream);

/* ---- attention ----------------------------------------------------------------------
 * Multi-head self-attention over the fused QKV projection (ScaledDotProductAttention,
 * transformer/Modules.py:14-25 inside MultiHeadAttention, SubLayers.py:44-56): 4 heads,
 * d_k = 64, scores / sqrt(64), keys t >= len[b] masked with -inf, softmax, PV.  Online
 * softmax in LDS/registers: the [4B, L, L] score tensor is never materialised.
 * qkv: [B, L, 768] = q | k | v (each 4 heads x 64, head-major within the 256 block);
 * out: [B, L, 256] (b x lq x (n*dv), SubLayers.py:54-56).  lse (optional, [B,4,L]) gets
 * the log-sum-exp per query row for the backward pass.
 * cu (optional, int32 [B+1], requires len): packed rows -- item b occupies rows cu[b] .. cu[b]+len[b]-1
 * of qkv / out (styler_pack_plan); L is then only the grid / lse stride (the longest item allowed). */
int styler_attention_fwd(const float* qkv, float* out, float* lse, int B, int L,
                         const int64_t* len, const int32_t* cu, void* stream);

/* Throughput-mode variants: bf16 MFMA operands (Q, K, V, P, dS, dO rounded to bf16 while staged),
 * fp32 softmax / accumulation; same arguments as the exact-fp32 entry points.  Query rows t >= len[b]
 * are don't-care in the reference's use (zeroed after the following LayerNorm, Layers.py:29): the
 * forward may leave zeros in `out` / `lse` there, and the backward treats their dout as zero. */
int styler_attention_fwd_bf16(const float* qkv, float* out, float* lse, int B, int L,
                              const int64_t* len, const int32_t* cu, void* stream);
/* The same forward with io_flags.  STYLER_IO_X_BF16: qkv is STORED as bf16 ([rows][768] elements; pass the bf16 pointer) --
 * throughput mode writes it that way from the QKV GEMM's epilogue, its only readers are these kernels.  The 1/sqrt(d_k)
 * scale is then applied to the raw scores inside the exponent (one fma) instead of to q before its rounding: every
 * operand is still rounded once (SubLayers.py:44-52, Modules.py:14-25). */
int styler_attention_fwd_bf16_io(const void* qkv, void* out, float* lse, int B, int L,
                                 const int64_t* len, const int32_t* cu, int io_flags, void* stream);
/* ... STYLER_IO_Y_BF16 on the forward: `out` is written as bf16 (its readers -- the output projection, that projection's
 * weight gradient, the backward's delta -- round it to bf16 or accept it rounded).
 * Backward: io_flags & STYLER_IO_Y_BF16: dqkv is written as bf16 [B,L,768] (what its consumers -- the QKV dX GEMM and the three
 * weight gradients -- round it to anyway); STYLER_IO_X_BF16: qkv is stored as bf16 (as above); STYLER_IO_MASK_BF16: `out` is
 * stored as bf16; STYLER_IO_RES_BF16: `dout` is stored as bf16 (pass the bf16 pointers in their places). */
int styler_attention_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse,
                              void* dqkv, float* delta_ws, int B, int L, const int64_t* len,
                              const int32_t* cu, int io_flags, void* stream);
/* The same two operations in the bf16x3 arithmetic (STYLER_PREC_BF16X3; Modules.py:14-25, SubLayers.py:41-56 and their
 * autograd): fp32 tensors on both sides, every MFMA operand carried as hi + lo bf16 and every product as hi hi + hi lo +
 * lo hi on the bf16 matrix cores, fp32 softmax / lse / delta.  Arguments as styler_attention_fwd / styler_attention_bwd. */
int styler_attention_fwd_x3(const float* qkv, float* out, float* lse, int B, int L, const int64_t* len, const int32_t* cu,
                            void* stream);
int styler_attention_bwd_x3(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                            float* delta_ws, int B, int L, const int64_t* len, const int32_t* cu, void* stream);

/* ---- normalisation / epilogues ------------------------------------------------------
 * y = LayerNorm_256(x + res) * gamma + beta, then rows t >= len[b] set to 0
 * (SubLayers.py:58-59,86-87 + Layers.py:29,32).  res, len may be NULL.  C must be 256.
 * If dot_w != NULL the kernel instead writes the scalar out[b,t] = <y, dot_w> + dot_b[0]
 * (masked) -- the StylePredictor's LayerNorm -> dropout -> Linear(256,1) -> masked_fill
 * tail, modules.py:449-465 -- and y may be NULL; drop_p > 0 applies the train-mode dropout behind the
 * LayerNorm (to y, and to what the tail sees) with the counter-based stream of styler_dropout (seed drop_seed).
 * in_drop_p > 0 (train mode) applies dropout to x BEFORE the residual add -- the nn.Dropout of
 * SubLayers.py:58,86 -- with the stream styler_dropout(x [B*L, 256], seed in_drop_seed) would draw;
 * sum_out (optional) receives the pre-norm sum dropout(x) + res that styler_layernorm_bwd consumes
 * (defined on unmasked rows only).  y16 (optional, with y): a second copy of y as bf16 (round to nearest even; zeros on
 * masked rows) for the GEMMs that consume it as their activation operand (styler_conv_gemm with STYLER_IO_X_BF16). */
int styler_add_layernorm(const float* x, int64_t ldx, const float* res, int64_t ldres,
                         const float* gamma, const float* beta, float* y, int64_t ldy,
                         const float* dot_w, const float* dot_b, float* dot_out, int B, int L,
                         int C, const int64_t* len, float drop_p, uint64_t drop_seed, float in_drop_p,
                         uint64_t in_drop_seed, float* sum_out, int64_t ldsum, uint16_t* y16, int64_t ldy16,
                         int io_flags, void* stream);

/* Round 6: the Linear in front of that LayerNorm and the LayerNorm as ONE launch (throughput mode, bf16 operands):
 *   s = dropout(a W^T + bias, drop_p) + res;   y = LayerNorm_256(s) * gamma + beta;   rows t >= len[b]: y = 0
 * a: bf16 rows [B*L, K] (row stride lda elements, K % 64 == 0), w: bf16 [256, K] (nn.Linear layout / the k = 1 Conv1d
 * weight), bias fp32 [256] or NULL.  Replaces `fc` + dropout + residual + layer_norm of MultiHeadAttention
 * (transformer/SubLayers.py:55-61) and `w_2` + dropout + residual + layer_norm of PositionwiseFeedForward
 * (SubLayers.py:86-89) together with the masked_fill of Layers.py:29,32: a 128 x 256 tile owns whole rows, the fp32
 * projection never reaches HBM.  res / y / sum_out / y16 / len / drop_seed / io_flags (STYLER_LN_*) exactly as
 * styler_add_layernorm (in_drop_p = drop_p there): the same dropout stream, the same statistics, so
 * styler_layernorm_bwd consumes sum_out unchanged.  styler_linear_ln_ok says whether a shape is taken (1) or not (0). */
int styler_linear_ln(const void* a, int64_t lda, int K, const void* w, const float* bias, const void* res,
                     int64_t ldres, const float* gamma, const float* beta, void* y, int64_t ldy, void* sum_out,
                     int64_t ldsum, uint16_t* y16, int64_t ldy16, int B, int L, const int64_t* len, float drop_p,
                     uint64_t drop_seed, int io_flags, void* stream);
int styler_linear_ln_ok(int64_t rows, int K, int n, int64_t lda);
/* Measurement hook (tools/linear_ln_trace.py): the launches that follow write 8 x uint64 per block into buf -- block, then
 * 100 MHz timestamps of entry, masks known, first K step landed, K loop done, tile staged, rows issued, stores acknowledged;
 * NULL switches it off.  Replaces nothing in the reference. */
int styler_linear_ln_set_trace(void* buf);

/* y = relu(GroupNorm(x)) with groups of 16 channels and statistics over 16 ch x the whole
 * padded L (modules.py:103-113,171-175; eps 1e-5).  In place allowed (y == x).
 * workspace: 2*B*C/16 doubles (scratch; ws_zeroed != 0 = the caller hands it over already zeroed, e.g. a
 * slice of one slab cleared once per step, and the entry point skips its memset -- same for the three
 * other norm entry points with a workspace); stats (optional): [B][C/16][2] floats = mean, rstd of every
 * group, the input styler_groupnorm_relu_bwd needs. */
int styler_groupnorm_relu(const float* x, int64_t ldx, const float* gamma, const float* beta,
                          void* y, int64_t ldy, float* stats, double* workspace, int ws_zeroed, int B,
                          int L, int C, int io_flags, void* stream);
/* io_flags & STYLER_IO_Y_BF16 (here and in styler_batchnorm_train; STYLER_IO_Y_BF16 on the two backward entry points means
 * dx, and STYLER_IO_X_BF16 there means the incoming dy is bf16): the output tensor is bf16 (ldy in elements).  Throughput mode stores the activations between the convolutions of
 * the AudioEncoder / PostNet stacks -- and the gradients w.r.t. those convolutions' outputs -- that way: their only
 * consumers (the next convolution, the dX GEMM, the weight gradient) round to bf16 first, so no result changes and
 * every one of them moves half the bytes. */

/* io_flags & STYLER_IO_Z_BF16 (the same four entry points): x itself -- the convolution output the norm reads, kept for
 * the backward -- is bf16 (ldx in elements; pass the bf16 pointer as `x`).  GroupNorm accepts it in its single-pass variants
 * only: items of up to styler_groupnorm_fused_rows(bwd) rows (1024 forward, 512 backward; 0 = variants switched off). */
int styler_groupnorm_fused_rows(int bwd);

/* BatchNorm1d folding for eval mode (Layers.py:91,105,118): scale = g * rsqrt(var + eps),
 * shift = (conv_bias - mean) * scale + b.  All [C]. */
int styler_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, const float* conv_bias, float* scale,
                   float* shift, int C, void* stream);

/* Train-mode BatchNorm1d over (B*L) rows incl. padded frames: computes batch mean / biased
 * var per channel of x (= conv output incl. bias), writes y = act((x-mean)*rstd*g + b),
 * saves mean / rstd [C] for backward and updates running stats with momentum 0.1
 * (unbiased var), as torch.nn.BatchNorm1d does.  workspace: styler_bn_workspace_doubles(rows, C, segs) doubles -- one [2C]
 * slot per (segment, 128-row chunk) that the statistics kernels STORE and a fixed-order fold adds up (round 5: no fp64 atomics,
 * bit-reproducible, nothing to zero: ws_zeroed is ignored).
 * drop_p > 0: y = dropout(act(BN(x))) -- the F.dropout of Layers.py:126-128 in the same pass, with the stream
 * styler_dropout(seed drop_seed) would draw on the [rows, C] tensor. */
int styler_batchnorm_train(const float* x, const float* gamma, const float* beta, void* y,
                           float* save_mean, float* save_rstd, float* running_mean,
                           float* running_var, double* workspace, int ws_zeroed, int64_t rows, int C,
                           int act, float drop_p, uint64_t drop_seed, int segs, int io_flags, void* stream);
/* `segs` >= 1 (rows % segs == 0): rows [s * rows/segs, (s+1) * rows/segs) are normalised with THEIR OWN batch statistics,
 * exactly as `segs` separate calls (the clean and the noisy decode of styler.py:52,55 run through the PostNet as one batch;
 * Layers.py:126 uses per-call statistics); save_mean / save_rstd are [segs, C], the running statistics receive the segments'
 * momentum updates in order. */
int64_t styler_bn_workspace_doubles(int64_t rows, int C, int segs);

/* ---- embeddings / positions ---------------------------------------------------------
 * out[b,t,:] = emb[text[b,t],:] + pe[t,:]            (Models.py:73-74; emb [152,256])
 * x may be NULL-free variant: out[b,t,:] = x[b,t,:] + pe[t,:]   (Models.py:124-125) */
int styler_embed_pos(const int64_t* text, const float* emb, const float* pe, float* out,
                     int B, int L, int C, void* stream);
int styler_add_pos(const float* x, int64_t ldx, const float* pe, float* out, int B, int L,
                   int C, void* stream);
/* out[r,:] = table[ids[r],:] -- nn.Embedding lookup with the int32 bucket ids the bucketise kernel emits: the un-summed
 * pitch_embedding / energy_embedding tensors StyleModeling.predict_inference returns (modules.py:300-303). */
int styler_gather_rows(const int32_t* ids, const float* table, float* out, int64_t rows, int C, void* stream);
/* Sinusoid table rows [0, L) x 256, angles in float64 then cast (Models.py:11-30); used
 * when L > 1000 in eval mode (Models.py:69-71,120-122). */
int styler_sinusoid_table(float* pe, int L, int C, void* stream);

/* ---- style encoders -----------------------------------------------------------------
 * quantize_1D_torch (utils.py:417-429) fused with the first Conv1d(257 -> C, k=5) of the
 * f0 / energy streams (modules.py:120-146): idx = 0 if v <= 0 else rint(v*255)+1; the
 * one-hot [B,L,257] is never built: y[b,t,:] = bias + sum_j wt[j, idx[b,t+j-2], :].
 * wt is the conv weight permuted to [5, 257, C].  err_flag (int32[1]) is set to 1 if any
 * v > 1 (the reference's assert, utils.py:423).  idx_out (optional) int32 [B,L]. */
int styler_onehot_conv5(const float* v, const float* wt, const float* bias, float* y,
                        int64_t ldy, int32_t* idx_out, int32_t* err_flag, int B, int L, int C,
                        void* stream);

/* mel_calibrator (utils.py:351-384): per item resample frames mel_len[b] -> src_len[b]
 * by near-equal segment means or repeats; rows s >= src_len[b] are zero.
 * x [B, T, C] -> y [B, S, C]. */
int styler_mel_calibrate(const float* x, int64_t ldx, float* y, int64_t ldy,
                         const int64_t* mel_len, const int64_t* src_len, int B, int T, int S,
                         int C, void* stream);
/* The same with io_flags.  STYLER_IO_X_BF16: x is stored as bf16 (ldx in elements) -- throughput-mode training keeps the
 * concatenated output of the AudioEncoder's conv stacks (modules.py:176-177) that way; the means are taken in fp32. */
int styler_mel_calibrate_io(const void* x, int64_t ldx, float* y, int64_t ldy,
                            const int64_t* mel_len, const int64_t* src_len, int B, int T, int S,
                            int C, int io_flags, void* stream);

/* One direction-pair of one nn.LSTM layer (modules.py:117,179-182; gate order i,f,g,o; zero
 * initial state; run over the whole padded S).  gx [B, S, 2*4H] holds the input
 * projections x W_ih^T + b_ih + b_hh for (forward | reverse) (computed by
 * styler_conv_gemm); w_hh [2, 4H, H].  out [B, S, 2H] = forward | reverse hidden states.
 * H must be 64 or 80.  cell_out (optional, [B,S,2H]) / gates_out (optional [B,S,2*4H],
 * post-activation) are saved for backward. */
int styler_lstm_bidir(const float* gx, const float* w_hh, float* out, float* cell_out,
                      float* gates_out, int B, int S, int H, void* stream);

/* Up to 4 independent BiLSTM layers in ONE launch (same semantics as styler_lstm_bidir per
 * descriptor; `descs` is a HOST array of `count` <= 4 entries, H in {64, 80}). */
typedef struct StylerLstmDesc {
  const float* gx;
  const float* w_hh;
  float* out;
  float* cell_out;   /* optional */
  float* gates_out;  /* optional */
  int32_t H;
  int32_t _pad;
} StylerLstmDesc;
int styler_lstm_bidir_multi(const StylerLstmDesc* descs, int count, int B, int S, void* stream);

/* AugmentationClassifier tail (modules.py:29-45): h [B,S,256] (= d_fc1 output) ->
 * LayerNorm -> ReLU -> Linear(256,2) -> LogSoftmax -> mean over all S rows (pads
 * included) -> out [B,2]. */
int styler_aug_classifier_tail(const float* h, const float* ln_g, const float* ln_b,
                               const float* w2, const float* b2, float* out, int B, int S,
                               void* stream);

/* ---- LengthRegulator ----------------------------------------------------------------
 * modules.py:396-423 + utils.pad (utils.py:332-348).  Step 1 (scan): durations [B,S]
 * (int64 if dur_is_float == 0, else fp32 truncated like int()) -> inclusive prefix sums
 * csum [B,S] int32 and mel_len [B] int64 (= un-cropped sum).  If log_d != NULL the
 * free-running rounding is fused: d = max(rint(exp(log_d) - 1) * d_control, 0)
 * (modules.py:357-358) and also written to dur_out (fp32, optional).
 * Step 2 (expand): out[b,t,:] = x[b, i, :] with csum[b,i-1] <= t < csum[b,i], zero for
 * t >= min(mel_len[b], T); frame_idx [B,T] int32 (optional) gets i or -1. */
int styler_duration_scan(const void* dur, int dur_is_float, const float* log_d, float d_control,
                         float* dur_out, int32_t* csum, int64_t* mel_len, int B, int S,
                         void* stream);
int styler_length_regulate(const float* x, int64_t ldx, const int32_t* csum, float* out,
                           int64_t ldo, int32_t* frame_idx, int B, int S, int T, int C,
                           void* stream);

/* bucketize + 2 embeddings + 4-way add (modules.py:365-385):
 * out[b,t,:] = text[b,t,:] + pitch_emb[bucketize(p[b,t]*p_scale, pitch_bins)] +
 *              speaker[b,t,:] + energy_emb[bucketize(e[b,t]*e_scale, energy_bins)]
 * bins are 255 ascending fp32 bounds, bucketize(right=False).  noise (optional) is added
 * into out2 = out + noise (the noisy-branch input, styler.py:55).  ids (optional) int32. */
int styler_bucket_embed_add(const float* text, int64_t ldt, const float* speaker, int64_t lds,
                            const float* p, float p_scale, const float* e, float e_scale,
                            const float* pitch_bins, const float* energy_bins,
                            const float* pitch_emb, const float* energy_emb, float* out,
                            const float* noise, int64_t ldn, float* out2, int32_t* p_ids,
                            int32_t* e_ids, int B, int T, void* stream);

/* Round 6: the decoder-input concatenation of StyleModeling.forward (modules.py:335-350) as one launch.  All operands fp32,
 * contiguous [B, S, 256] (spk: [B, 256]); enc [B, S, 1280] = [text | pitch_up + neck_up | speaker | neck_up + energy_up | residual_up],
 * dp [B, S, 256] = neck_up + duration_up (the duration predictor's input).  styler_add3: y = (a + b) + c over [rows, C] views. */
int styler_style_cat(const float* te, const float* pu, const float* tnu, const float* spk, const float* eu, const float* ru,
                     const float* du, float* enc, float* dp, int B, int S, void* stream);
int styler_add3(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c, int64_t ldc, float* y, int64_t ldy,
                int64_t rows, int C, void* stream);
/* elementwise helpers used to stitch channel slices: y = a (+ b) with row strides, and a
 * per-item row broadcast y[b,t,:] = (a[b,t,:] if a) + v[b,:] (speaker repeat,
 * modules.py:324-325,333). */
int styler_add2(const float* a, int64_t lda, const float* b, int64_t ldb, float* y,
                int64_t ldy, int64_t rows, int C, void* stream);
int styler_add_rowvec(const float* a, int64_t lda, const float* v, int64_t ldv, float* y,
                      int64_t ldy, int B, int L, int C, void* stream);
/* bf16x3 arithmetic (precision STYLER_PREC_BF16X3 of the host layer): fp32-class products on the bf16 matrix cores.  An
 * operand is carried as hi + lo (hi = bf16(v), lo = bf16(v - hi)); a x w = a_hi w_hi + a_hi w_lo + a_lo w_hi is ONE bf16
 * GEMM over a three times longer contraction axis: the activation blocks (hi, lo, hi) against the weight row
 * [w_hi | w_hi | w_lo] (StylerCopyDesc.flags bit 3 = low part); styler_conv_gemm runs it with prec = STYLER_PREC_BF16,
 * cin = 3C.  styler_split3_bf16 writes the activation row: parts = 3 -> [hi | lo | hi] (bf16 [rows, 3C]); parts = 2 ->
 * the compact [hi | lo] ([rows, 2C], C % 64 == 0: the GEMM takes it with STYLER_IO_X3A).  Replaces the fp32 arithmetic of
 * every nn.Linear / nn.Conv1d of the path (transformer/SubLayers.py:41-61,72-89; modules.py; transformer/Layers.py:78-118)
 * at ~2.4x the bf16 cost instead of 16x.  `count` (optional, device int64): only rows < count[0] are written (packed
 * rows).  styler_lo_part: y = bf16(x - hi) stored as fp32 (tests; the weight-gradient kernels stage low parts themselves). */
int styler_split3_bf16(const float* x, int64_t ldx, void* y, int64_t rows, int C, const int64_t* count, int parts,
                       void* stream);
int styler_lo_part(const float* x, int64_t ldx, float* y, int64_t rows, int C, const int64_t* count, void* stream);
/* Up to 8 styler_split3_bf16 passes in one launch: seg.src = fp32 rows (ld_src), seg.dst = the bf16 split (contiguous rows of
 * parts * C), seg.rows, seg.C, seg._pad = parts (2 or 3); ld_dst is ignored. */
struct StylerCopySeg;
int styler_split3_multi(const struct StylerCopySeg* segs, int count, void* stream);
/* Up to 8 strided row copies in one launch (the descriptors travel in the kernel arguments): segment k copies `rows` rows of
 * `C` floats (C % 4 == 0) from src (row stride ld_src; NULL = zero fill) to dst (row stride ld_dst).  The torch.cat /
 * torch.split plumbing of modules.py:218-223,350,362 and the gathered slice gradients of their backward: one launch per
 * concatenation instead of one copy per part. */
typedef struct StylerCopySeg {
  const void* src;
  void* dst;
  int64_t ld_src, ld_dst, rows;
  int32_t C, _pad;
} StylerCopySeg;
int styler_copy_rows_multi(const StylerCopySeg* segs, int count, void* stream);
/* y = leaky_relu(scale * (a (+ b) (+ c)), slope) over `count` contiguous floats: the pre-activation of every vocoder
 * conv (hifigan/models.py:96,98,157) fused with the resblock average `xs / num_kernels` (164). */
int styler_leaky_sum(const float* a, const float* b, const float* c, float* y, int64_t count,
                     float scale, float slope, void* stream);

/* get_mask_from_lengths (utils.py:223-232): mask[b,t] = (t >= len[b]), 1 byte per element
 * (torch.bool storage), True = padding. */
int styler_length_mask(const int64_t* len, uint8_t* mask, int B, int L, void* stream);
/* Round 6: the two masks of one forward (src_mask [B0, L0], mel_mask [B1, L1]; styler.py:42-43) in one launch. */
int styler_length_mask2(const int64_t* len0, uint8_t* mask0, int B0, int L0, const int64_t* len1, uint8_t* mask1, int B1, int L1,
                        void* stream);

/* ---- losses (loss.py:16-50) -----------------------------------------------------------
 * acc[0] += sum over valid rows (t < len[b]) and C columns of (a-b)^2  (kind 0)
 *           or |a-b| (kind 1); acc[1] += number of valid elements.  acc is double[2],
 * caller zeroes it.  Mean-reduced loss = acc[0]/acc[1] (masked_select + MSELoss/L1Loss). */
int styler_masked_err_sum(const float* a, int64_t lda, const float* b, int64_t ldb,
                          double* acc, int kind, int B, int L, int C, const int64_t* len,
                          void* stream);

/* ---- STFT -> mel (audio/stft.py:51-79,141-160) ----------------------------------------
 * wav [B, N] in [-1,1] -> reflect padding 512, hop 256, windowed DFT, magnitude,
 * mel = log(clamp(mel_basis @ mag, 1e-5)), energy = ||mag||_2, F = 1 + N/256 frames.
 * The framing conv (stft.py:65-69) is the MFMA implicit GEMM over the padded signal viewed
 * as [B, F+3, 256] with 4 taps.  basis: [1028, 1024] (rows 0..512 Re, 513..1025 Im of the
 * windowed DFT, last 2 rows zero), fp32 or bf16 per `prec`; mel_basis: [80, 516] fp32 (cols
 * 513..515 zero).  Outputs channels-last: mel [B, F, 80], energy [B, F], mag (optional)
 * [B, F, 516].  workspace: styler_stft_mel_workspace_bytes(B, N) bytes.
 * err_flag (int32[1], optional) is set to 1 if any |wav| > 1 (the asserts at stft.py:151-152). */
int64_t styler_stft_mel_workspace_bytes(int B, int N);
int styler_stft_mel(const float* wav, int64_t ldw, const void* basis, const float* mel_basis,
                    float* mag, float* mel, float* energy, void* workspace, int32_t* err_flag,
                    int B, int N, int prec, void* stream);
/* The same for a RAGGED batch (BASELINE config 5: utterances of 3-4 s in one batch): item b holds wav_len[b] <= N
 * samples (int64 [B], device; NULL = all N).  Each item is reflected at ITS end and transformed as if alone
 * (audio/tools.py:37-55 handles one utterance per call); frame_len[b] = 1 + wav_len[b] / 256 is written back (int64
 * [B]) and frames at or past it are zeros in mel / energy / mag / e_scaled (the collate's padding, utils.py:296-329).
 * e_scaled (optional, [B, F]) = clip((energy - e_min) / (e_max - e_min), 0, 1): utils.energy_rescaling
 * (utils.py:410-414), the model's `e_input`. */
int styler_stft_mel_varlen(const float* wav, int64_t ldw, const int64_t* wav_len, const void* basis,
                           const float* mel_basis, float* mag, float* mel, float* energy, float* e_scaled,
                           float e_min, float e_max, int64_t* frame_len, void* workspace, int32_t* err_flag,
                           int B, int N, int prec, void* stream);

/* ---- DeepSpeaker ResCNN speaker embedding (deepspeaker/audio_ds.py, batcher.py, conv_models.py) --------
 * The reference computes it in TensorFlow from `python_speech_features.fbank` features; neither dependency nor the
 * pretrained weights exist here (parity unpinned).  Every Conv2D except the first, the framing DFT, the mel
 * projection and the Dense layer are calls of styler_conv_gemm_pad; these are the kernels in between.
 *
 * styler_ds_vad_bounds: read_mfcc's silence trim (audio_ds.py:36-41): bounds[b] = (offsets[0], offsets[-1]) of the samples
 *   with |x| > np.percentile(|x|, 95) (exact order statistics, linear interpolation); wav_len (int64 [B], optional) gives
 *   each utterance's own length; thr_out (optional) receives the thresholds.
 * styler_ds_fbank: pre-emphasis 0.97, 551-sample rectangular frames every 221 samples, |rfft_1024|^2 / 1024, 64 HTK-mel
 *   filters, zero -> eps, per-frame (v - mean) / max(std, 1e-12) (mfcc_fbank / normalize_frames, audio_ds.py:128-139), for
 *   the 160-frame window starting at frame0[b] (crop_mode 0; batcher.py:23-29 draws it at random) or centred (crop_mode 1);
 *   frames past the utterance's own count are zero (pad_mfcc).  out: [B, 160, 64].  basis [1028, 672], fb [64, 516].
 * styler_ds_conv1: Conv2D(1 -> 64, 5x5, stride 2, TF 'same': pad 1 before / 2 after) + folded BatchNorm + clipped ReLU:
 *   x [B, H, W] -> y [B, H/2 + 3, W/2, 64] with rows 1..H/2 live (the H-padded layout of the ResCNN activations).
 * styler_ds_rows: dst [B, Hd + 3, W, C] <- src [B, Hsp, W, C]: dst row 1 + h = src row src_row0 + h * step, padding rows
 *   (0, Hd + 1, Hd + 2) zeroed; src == dst (step 1, src_row0 1): only the padding rows are rewritten.
 * styler_ds_crelu_add: out = min(max(a + b, 0), 20) (identity_block, conv_models.py:88-118).
 * styler_l2_normalize_rows: K.l2_normalize(y, axis = 1) (conv_models.py:64). */
int styler_ds_vad_bounds(const float* wav, int64_t ldw, const int64_t* wav_len, int B, int N, int64_t* bounds,
                         float* thr_out, void* stream);
int64_t styler_ds_fbank_workspace_bytes(int B);
int styler_ds_fbank(const float* wav, int64_t ldw, const int64_t* bounds, const int64_t* frame0, int crop_mode,
                    const void* basis, const float* fb, float* out, void* workspace, int B, int prec, void* stream);
int styler_ds_conv1(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int H,
                    int W, void* stream);
int styler_ds_rows(const float* src, float* dst, int B, int Hd, int Hsp, int W, int C, int src_row0, int step,
                   void* stream);
int styler_ds_crelu_add(const float* a, const float* b, float* out, int64_t n, void* stream);
int styler_l2_normalize_rows(const float* x, float* y, int rows, int C, void* stream);

/* ==== backward / training entry points ==================================================
 * Parameter gradients are ACCUMULATED (atomicAdd) into fp32 buffers in the PARAMETER layout
 * of the reference state dict; the caller zeroes them once per step. */

/* dz = mask(dy) * act'(y): backward of the fused GEMM epilogue (ReLU / tanh use the saved
 * post-activation y); rows t >= len[b] get zero (masked_fill backward). */
int styler_act_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dz,
                   int64_t lddz, int B, int L, int C, int act, const int64_t* len, void* stream);
/* Up to 8 of those in one launch (ReLU / tanh, no length mask; dz contiguous [rows, C]): the activation backward of the
 * members of a grouped Linear node. */
typedef struct StylerActSeg {
  const void* dy; const void* y; void* dz;
  int64_t lddy, ldy, rows;
  int32_t C, act;
} StylerActSeg;
int styler_act_bwd_multi(const StylerActSeg* segs, int count, void* stream);

/* Weight (+ bias) gradient of Linear / Conv1d, all kw taps in one launch (autograd of
 * SubLayers.py:41-43,72-76 etc.):
 *   dw[nn*stride_n + c*stride_c + j*stride_j] += sum_{b,t} dz[b,t,nn] * x[b, t+j-pad_left, c]
 *   db[nn] += sum_{b,t} dz[b,t,nn]     (db may be NULL; db2, optional, receives the same sums --
 *                                       nn.LSTM's bias_ih / bias_hh pair)
 * x = 0 outside the item.  prec F32: exact-fp32 MFMA; BF16: operands rounded to bf16 while
 * staged, fp32 accumulate.  Split over rows; partial tiles go to `workspace` and a reduce
 * kernel adds them into dw.  kw in {1,3,5,9}.
 * Linear: kw 1, pad 0, strides (cin, 1, 0); Conv1d [n, cin, kw]: pad kw/2, strides (cin*kw, kw, 1);
 * LSTM W_hh: kw 1, pad_left = +1 / -1 selects h_{t-1} / h_{t+1}. */
int styler_wgrad(const float* dz, int64_t lddz, const float* x, int64_t ldx, float* dw, float* db,
                 float* db2, int64_t stride_n, int64_t stride_c, int64_t stride_j, int B, int L, int n, int cin,
                 int kw, int pad_left, int prec, void* workspace, int defer_reduce, int io_flags,
                 void* stream);
/* styler_wgrad on packed rows (`rows` = capacity, counts[0] valid).  Workspace / split count: those of
 * styler_wgrad_workspace_bytes / styler_wgrad_splits for (B = 1, L = rows, pad_left = kw/2). */
int styler_wgrad_packed(const float* dz, int64_t lddz, const float* x, int64_t ldx, float* dw, float* db,
                        int64_t stride_n, int64_t stride_c, int64_t stride_j, int rows, int n, int cin,
                        int kw, int prec, void* workspace, int defer_reduce, const int32_t* rowinfo,
                        const int32_t* chunktab, const int64_t* counts, int io_flags, void* stream);
/* bf16 mode, BOTH operands resident as bf16 (io_flags X | Y), n % 8 == cin % 8 == 0: the operands are fetched by LDS-DMA
 * into a ring of stages (wgrad_dma_kernel, gemm_bwd.hip) instead of through registers.
 * styler_wgrad_dma_config(mode, stages128): mode 2 (default) = 512-thread blocks of two K groups that add their
 * accumulators through LDS, i.e. HALF the split-K partial tiles of the other kernels; 1 = 256-thread blocks with the
 * register-staged kernel's split plan and its partial tiles bit for bit; 0 = the register-staged kernel; -1 keeps the
 * value.  stages128 in {2, 3}: ring depth of the 128 x 128 Linear tile in mode 1 (0 keeps it).  Returns the previous
 * setting as mode | (stages128 << 2).  Env: STYLER_WGRAD_DMA=0|1|2.  The split count / workspace of a launch depend on its
 * operand formats: use the _io forms below with the io_flags the launch will be given. */
int styler_wgrad_dma_config(int mode, int stages128);
/* Round 5 tuning knobs of the bf16 weight-gradient engine: styler_wgrad_tune(knob, value) sets knob to value (0 | 1; any
 * other value only queries) and returns the previous value (STYLER_EINVAL for an unknown knob).
 *   knob 0 (default 0 -- measured: no gain, the operands are Infinity-Cache resident; env STYLER_WGRAD_XCDMAP): block -> (split, tile) map of launches with fewer than 8 splits.  1: the 8 XCDs
 *     form an (s8 x n8 x c8) grid over (splits x n-tiles x c-tiles) and every XCD owns one sub-box, so a dz / x column block is
 *     fetched into as few L2s as possible (the decoder FFN's k = 9 gradient: 250 -> ~85 MB fetched per launch); 0: the former
 *     tile-major map.  Results are bit-identical (the same blocks compute the same partial tiles).
 *   knob 1 (default 1, env STYLER_WGRAD_K5_TALL): k = 5 gradients with both operands bf16-resident, n % 128 == 0 and at least 64
 *     tiles of 64 x 64 (the PostNet's 512 -> 512 convolutions: 128.6 -> 115.2 us per launch) on a
 *     128 (n) x 64 (c) x 5 taps block tile instead of 64 x 64 x 5 (fewer operand bytes per MFMA; the split plan changes with it,
 *     so query the workspace / split count with the _io forms AFTER setting the knob).
 *     Value 2 (test-only): the tall tile for EVERY eligible k = 5 gradient, whatever its tile count.
 *   knob 2 (default 0 -- measured: k = 9 122 -> 128 us, no gain; env STYLER_WGRAD_RING4): EXPERIMENT, a ring of four
 *     stages instead of three in the LDS-DMA kernels.  Same partial tiles.
 * A caller that flips knob 1 between steps must re-plan: the split counts and workspace bytes of the affected launches
 * change, so sizes cached from the _io queries (styler_amd.ops.wgrad_tune resets WgradArena's sizing and descriptor tables; a
 * captured hipGraph must be re-captured).
 * (autograd of transformer/SubLayers.py:72-76, transformer/Layers.py:78-118) */
int styler_wgrad_tune(int knob, int value);
int styler_wgrad_x3cat_ok(int n, int cin, int kw, int pad_left);
int styler_wgrad_splits_io(int B, int L, int n, int cin, int kw, int pad_left, int prec, int io_flags);
int64_t styler_wgrad_workspace_bytes_io(int B, int L, int n, int cin, int kw, int pad_left, int prec, int io_flags);
/* Split count styler_wgrad uses for a shape (workspace = splits * n * kw * cin floats). */
int styler_wgrad_splits(int B, int L, int n, int cin, int kw, int pad_left, int prec);

/* Grouped launch of many weight gradients (bf16 mode): the caller fills one descriptor per problem on the host with
 * styler_wgrad_group_desc (the return value is the member's block count, 0 = does not qualify, launch it with
 * styler_wgrad), sorts the members by `variant`, copies the array to the device and launches once per variant.
 * Partial tiles land in each member's `workspace` exactly as styler_wgrad(defer_reduce = 1) leaves them: reduce with
 * styler_wgrad_reduce_multi. */
typedef struct {
  uint64_t dz, x, db, db2, ws;         /* device pointers */
  uint64_t counts;                     /* packed rows (styler_pack_plan counts) or 0 */
  uint64_t chunktab;                   /* packed rows, kw > 1: the item-aligned K-chunk table, else 0 */
  int64_t lddz, ldx;
  int32_t B, L, n, cin, pad_left, ct, cpi, cps, tiles, splits, block_start, nblocks, variant, kw;
} StylerWgradGroupDesc;
/* Members may be any bf16-mode weight gradient styler_wgrad / styler_wgrad_packed accept (conv taps, packed rows,
 * io_flags); `want_splits` > 0 caps the member's split-K count (a group fills the chip together, so its members need far
 * fewer partial tiles than a stand-alone launch); the caller assigns block_start (cumulative nblocks, members with
 * splits >= 8 aligned to a multiple of 8) and launches each variant's members with one styler_wgrad_group call. */
int styler_wgrad_group_desc(StylerWgradGroupDesc* out, const float* dz, int64_t lddz, const float* x,
                            int64_t ldx, float* db, float* db2, int B, int L, int n, int cin, int kw, int pad_left,
                            int prec, void* workspace, const int64_t* packed_counts, const int32_t* packed_chunktab,
                            int io_flags, int want_splits);
int styler_wgrad_group(const StylerWgradGroupDesc* desc_dev, int count, int total_blocks, int variant, void* stream);

/* Deferred reduction: with defer_reduce != 0 styler_wgrad leaves its partial tiles in `workspace`;
 * one styler_wgrad_reduce_multi launch then folds the partials of MANY gradients into their
 * parameter-layout buffers (dw[nn*stride_n + c*stride_c + j*stride_j] += sum_split ws[...]).
 * `desc_dev` is a device array; descriptor i owns blocks [block_start, next.block_start), one
 * block per 1024 outputs: block_start[0] = 0, total_blocks = sum ceil(n*kw*cin / 1024). */
typedef struct StylerWgradDesc {
  const void* ws;
  void* dw;
  int64_t stride_n, stride_c, stride_j;
  int64_t block_start;
  int32_t n, cin, kw, splits;
} StylerWgradDesc;
int styler_wgrad_reduce_multi(const StylerWgradDesc* desc_dev, int count, int64_t total_blocks,
                              void* stream);
/* Round 6: the same fold with the descriptor of every block given (blockmap[b], device int32 [total_blocks]) instead of searched. */
int styler_wgrad_reduce_multi_map(const StylerWgradDesc* desc_dev, int count, int64_t total_blocks, const int32_t* blockmap,
                                  void* stream);
/* Blocks descriptor i owns (block_start[i+1] - block_start[i]): ceil(n*kw*cin / 1024), or -- conv taps reduced into the
 * parameter layout (stride_j == 1, stride_c == kw) -- n * ceil(cin / 128): one block per (n, 128 channels, all taps). */
int64_t styler_wgrad_reduce_blocks(int n, int cin, int kw, int64_t stride_c, int64_t stride_j);
/* bytes of `workspace` styler_wgrad needs for a shape (split-K partial tiles, reduced without atomics) */
int64_t styler_wgrad_workspace_bytes(int B, int L, int n, int cin, int kw, int pad_left, int prec);

/* out[c] += sum over rows of dz[row, c] (bias gradient); out2 (optional) gets the same sum
 * (nn.LSTM's b_ih and b_hh). */
int styler_colsum(const float* dz, int64_t lddz, float* out, float* out2, int64_t rows, int C,
                  void* stream);

/* Weight of the dX convolution: dst[c, j, nn] = src[nn, c, kw-1-j] (src in parameter layout
 * [n, cin, kw]); dx = styler_conv_gemm(dz, dst, cin := n, n := cin, kw). */
int styler_repack_weight_bwd(const float* src, void* dst, int n, int cin, int kw, int out_bf16,
                             void* stream);

/* Attention backward (recomputes P from lse): dqkv [B,L,768]; delta_ws: B*4*L floats. */
int styler_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                         float* dqkv, float* delta_ws, int B, int L, const int64_t* len,
                         const int32_t* cu, void* stream);

/* LayerNorm(256) backward from the saved INPUT x (= pre-norm sum).  dx may be NULL.  With
 * dot_w (predictor tail) the incoming gradient is dout [B,L] and ddot_w/ddot_b accumulate.
 * dx_drop (optional) = dx * the dropout mask of the forward's in_drop (the gradient of the branch that
 * went through dropout; dx itself is the gradient of the residual).
 * replicas > 1: dgamma / dbeta / ddot_w point to ZEROED scratch [replicas][256] instead of the gradients (block i
 * adds into replica i % replicas; the caller folds them, e.g. with styler_wgrad_reduce_multi descriptors
 * n = 256, cin = kw = 1, splits = replicas): hundreds of blocks adding into one 256-float vector serialise in L2.
 * drop_p > 0 without dot_w: dy is the gradient w.r.t. dropout(LayerNorm(x)) (styler_add_layernorm's drop_p written to y).
 * flags & STYLER_LNB_RELU_INPUT: x is the output of a ReLU (StylePredictor: Conv1d -> ReLU -> LayerNorm,
 * modules.py:430-447) and dx is returned as the gradient w.r.t. the ReLU's INPUT (dx where x > 0, else 0). */
#define STYLER_LNB_RELU_INPUT 1
#define STYLER_LNB_X_BF16 2     /* x (the saved pre-norm sum) is bf16 */
#define STYLER_LNB_DY_BF16 4    /* dy is bf16 */
#define STYLER_LNB_DX_BF16 8    /* dx is written as bf16 */
#define STYLER_LNB_DXD_BF16 16  /* dx_drop is written as bf16 */
#define STYLER_LNB_DOTB_SLOTS 64 /* with replicas >= the launch's blocks: ddot_b is a [replicas] slot array (stored, not added) */
int styler_layernorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy,
                         const float* gamma, const float* beta, float* dx, int64_t lddx,
                         float* dgamma, float* dbeta, const float* dot_w, const float* dout,
                         float* ddot_w, float* ddot_b, int B, int L, int C, const int64_t* len,
                         float drop_p, uint64_t drop_seed, float in_drop_p, uint64_t in_drop_seed,
                         float* dx_drop, int64_t lddxd, int replicas, int flags, void* stream);

/* dst_k[c] += sum_{r < replicas} src_k[r * n + c] in replica order, k = 0..2 (src_1/dst_1, src_2/dst_2 optional): the
 * deterministic fold of the per-block parameter-gradient slots styler_layernorm_bwd writes with replicas >= 256 (autograd
 * of nn.LayerNorm's weight / bias, SubLayers.py:29,61; modules.py:441-447).  Stand-alone calls use it; inside a training
 * step the same fold is a member of styler_wgrad_reduce_multi. */
int styler_fold_replicas(const float* src0, const float* src1, const float* src2, float* dst0, float* dst1,
                         float* dst2, int replicas, int n, void* stream);

/* stats = the forward's [B][C/16][2] (mean, rstd); workspace 2*B*C/16 doubles (scratch). */
int styler_groupnorm_relu_bwd(const float* x, int64_t ldx, const void* dy, int64_t lddy,
                              const float* gamma, const float* beta, const float* stats, void* dx,
                              int64_t lddx, float* dgamma, float* dbeta, double* workspace, int ws_zeroed,
                              int B, int L, int C, int io_flags, void* stream);

/* BatchNorm1d(train)+act(+dropout) backward; x, y, dy, dx contiguous [rows, C]; workspace styler_bn_workspace_doubles(rows, C, segs) doubles.
 * y may be NULL when beta is given: the tanh output is then recomputed from x (one read less); drop_p / drop_seed
 * must repeat the forward's (the mask is regenerated, dy is the gradient of the dropped output). */
int styler_batchnorm_bwd(const float* x, const float* y, const void* dy, const float* gamma,
                         const float* save_mean, const float* save_rstd, void* dx, float* dgamma,
                         float* dbeta, double* workspace, int ws_zeroed, int64_t rows, int C, int act,
                         const float* beta, float drop_p, uint64_t drop_seed, int segs, int io_flags, void* stream);

int styler_embed_bwd(const int64_t* text, const float* dy, int64_t lddy, float* demb, int B, int L,
                     int C, void* stream);
/* The same without atomics: one block per table row (V rows, row 0 = padding_idx) adds the dy rows of its tokens in token
 * order -- bit-reproducible (nn.Embedding backward, Models.py:52-53). */
int styler_embed_bwd_det(const int64_t* text, const float* dy, int64_t lddy, float* demb, int B, int L, int C, int V,
                         void* stream);
/* Backward of styler_onehot_conv5: materialise the one-hot rows [rows, 260] (257 zero-padded)
 * so that the weight gradient becomes styler_wgrad(dy, onehot, dw [C,257,5], db, strides
 * (257*5, 5, 1), n = C, cin = 257, kw = 5) on the MFMA engine. */
int styler_onehot_expand(const float* v, float* onehot, int64_t rows, void* stream);
int styler_mel_calibrate_bwd(const float* dy, int64_t lddy, float* dx, int64_t lddx,
                             const int64_t* mel_len, const int64_t* src_len, int B, int T, int S,
                             int C, void* stream);
/* io_flags & STYLER_IO_Y_BF16: dx is written as bf16 (lddx in elements). */
int styler_mel_calibrate_bwd_io(const float* dy, int64_t lddy, void* dx, int64_t lddx,
                                const int64_t* mel_len, const int64_t* src_len, int B, int T, int S,
                                int C, int io_flags, void* stream);
int styler_lstm_bidir_bwd(const float* dout, const float* gates, const float* cell,
                          const float* w_hh, float* dgp, int B, int S, int H, void* stream);
/* Backward of up to 4 BiLSTM layers in one launch (HOST descriptor array). */
typedef struct StylerLstmBwdDesc {
  const float* dout;
  const float* gates;
  const float* cell;
  const float* w_hh;
  float* dgp;
  int32_t H;
  int32_t _pad;
} StylerLstmBwdDesc;
int styler_lstm_bidir_bwd_multi(const StylerLstmBwdDesc* descs, int count, int B, int S, void* stream);
/* Round 6: the same two recurrences on the matrix cores (csrc/lstm_mfma.hip) -- a block takes 16 items of one (layer,
 * direction) and runs every step as v_mfma_f32_16x16x32_bf16 against W_hh fragments held in registers; h_t / the gate
 * gradients cross the block as bf16 through LDS.  parts = 1: bf16 products, fp32 accumulation (the bf16 mode's GEMM
 * arithmetic); parts = 3: bf16x3 products (operands split hi + lo, three products per product).  Same descriptors, same
 * saved tensors (post-activation gates, cell, out) and outputs as the two entry points above, which stay the fp32 path.
 * Replaces: nn.LSTM(bidirectional=True) forward + autograd, modules.py:100-101,117,132,147,162,179-182. */
int styler_lstm_bidir_multi_mfma(const StylerLstmDesc* descs, int count, int B, int S, int parts, void* stream);
int styler_lstm_bidir_bwd_multi_mfma(const StylerLstmBwdDesc* descs, int count, int B, int S, int parts, void* stream);
int styler_aug_classifier_tail_bwd(const float* h, const float* ln_g, const float* ln_b,
                                   const float* w2, const float* b2, const float* dout, float* dh,
                                   float* dln_g, float* dln_b, float* dw2, float* db2, int B, int S,
                                   void* stream);
/* ... with the four parameter gradients as slot arrays (io_flags & STYLER_IO_PARAM_SLOTS): [slots][256], [slots][256],
 * [slots][512], [slots][2], slots = styler_aug_classifier_tail_slots(B, S) (modules.py:38-45 autograd). */
int styler_aug_classifier_tail_slots(int B, int S);
int styler_aug_classifier_tail_bwd_io(const float* h, const float* ln_g, const float* ln_b, const float* w2, const float* b2,
                                      const float* dout, float* dh, float* dln_g, float* dln_b, float* dw2, float* db2, int B,
                                      int S, int io_flags, void* stream);
int styler_length_regulate_bwd(const float* dy, int64_t lddy, const int32_t* csum, float* dx,
                               int64_t lddx, int B, int S, int T, int C, void* stream);
int styler_bucket_embed_bwd(const float* dy, const int32_t* p_ids, const int32_t* e_ids,
                            float* dpitch_emb, float* denergy_emb, int B, int T, void* stream);
/* ... into slot arrays [styler_bucket_embed_slices()][256 * 256] per table (stored; folded by styler_wgrad_reduce_multi). */
int styler_bucket_embed_slices(void);
int styler_bucket_embed_bwd_slots(const float* dy, const int32_t* p_ids, const int32_t* e_ids, float* dpitch_slots,
                                  float* denergy_slots, int B, int T, void* stream);
/* out[b,:] (+)= sum_t x[b,t,:] */
int styler_rowsum(const float* x, int64_t ldx, float* out, int64_t ldo, int B, int L, int C,
                  int accumulate, void* stream);
/* da = gscale[0] * d(err)/da / acc[1] on valid rows (acc from styler_masked_err_sum). */
int styler_masked_err_bwd(const float* a, int64_t lda, const float* b, int64_t ldb,
                          const double* acc, const float* gscale, float* da, int kind, int B, int L,
                          int C, const int64_t* len, void* stream);
/* NLLLoss(mean) on [B,2] log-probs (loss.py:46-48): loss[0] (optional) and/or dlogp. */
int styler_nll(const float* logp, const int64_t* label, float* loss, const float* gscale,
               float* dlogp, int B, void* stream);
/* ---- loss head without glue kernels (csrc/losses.hip) -------------------------------------------------------------
 * Masked MSE (kind 0) / L1 (kind 1) term of loss.py:16-44 with the mean taken in the kernel: `acc` = 4 doubles, ZERO on
 * entry (sum, count, arrival ticket, pad); mean_out[0] = sum / count, written by the last block (may be NULL: sums only). */
int styler_masked_err_mean(const float* a, int64_t lda, const float* b, int64_t ldb, double* acc, float* mean_out,
                           int kind, int B, int L, int C, const int64_t* len, void* stream);
/* Round 6: size of a masked-error accumulator in doubles (zero on entry): totals, ticket and one slot pair per block -- the blocks
 * STORE their sums and the last one adds them in a fixed order (no atomics on the values: the means are deterministic). */
#define STYLER_MASKED_ACC_DOUBLES 2052
/* Up to 8 masked-error terms per launch (forward: mean + accumulator per term; backward: da = gscale * d mean / d a):
 * loss.py:16-50 calls MSELoss / L1Loss on masked_select copies five times per STYLERLoss.forward and twice per
 * cal_mel_loss -- here one launch each way per call.  Fields as the arguments of styler_masked_err_mean / _bwd
 * (a, b [B, L, C] with row strides lda / ldb; acc 4 doubles, zero on entry of the forward; len int64 [B] or NULL). */
typedef struct StylerMaskedTerm {
  const void* a; const void* b; void* acc; void* mean; const void* len; const void* gscale; void* da;
  int64_t lda, ldb;
  int32_t B, L, C, kind;
} StylerMaskedTerm;
int styler_masked_err_mean_multi(const StylerMaskedTerm* terms, int count, void* stream);
int styler_masked_err_bwd_multi(const StylerMaskedTerm* terms, int count, void* stream);
/* The three NLLLoss(mean) terms of one classifier triple, summed (loss.py:46-48, 60-68): loss[0] = sum_k -mean_b
 * logp_k[b, label[b]] (label NULL: every label = label_const, the zeros / ones of train.py:139,152); with dlogp3
 * ([3, B, 2]) also the gradient -gscale[0] / B at the label entry, 0 elsewhere. */
int styler_nll3(const float* lp0, const float* lp1, const float* lp2, const int64_t* label, int label_const, float* loss,
                const float* gscale, float* dlogp3, int B, void* stream);
/* out[0] = sum_i weights[i] * terms[i][0] over n <= 16 scalar device tensors (`terms` and `weights` are HOST arrays): the
 * total loss of train.py:156-160 in one launch; styler_scale_weights is its backward, out[i] = g[0] * weights[i]. */
int styler_weighted_sum(const float* const* terms, const float* weights, int n, float* out, void* stream);
int styler_scale_weights(const float* g, const float* weights, int n, float* out, void* stream);
/* Round 6: the tail of the train step's loss head as ONE launch each way (train.py:139-160: the classifier NLL of the main
 * pass, that of the DAT pass, and the weighted total).  means: n <= 8 HOST-array pointers to scalar device tensors (the
 * masked-error means, in the order of the total's sum); weights: n + 2 HOST floats (the last two weigh the two NLL3 terms);
 * lp6: HOST array of the six [B, 2] log-probability tensors (main d, p, e, then DAT d, p, e); label0 / label1 int64 [B] or
 * NULL (every label = label_const0 / 1).  out3 = {total, nll3(main), nll3(DAT)}.  Backward: gw[i] = g[0] * weights[i] for
 * i < n, d6 [6, B, 2] = -(g[0] * weights[n + k / 3]) / B at the label entry, 0 elsewhere.  Bit-identical to
 * styler_nll3 x 2 + styler_weighted_sum (forward) and styler_scale_weights + styler_nll3 x 2 (backward). */
int styler_loss_tail(const float* const* means, const float* weights, int n, const float* const* lp6,
                     const int64_t* label0, int label_const0, const int64_t* label1, int label_const1, int B,
                     float* out3, void* stream);
int styler_loss_tail_bwd(const float* g, const float* weights, int n, const float* const* lp6,
                         const int64_t* label0, int label_const0, const int64_t* label1, int label_const1, int B,
                         float* gw, float* d6, void* stream);
/* Registers the device address of a uint64 step counter (or NULL to unregister).  Every dropout-drawing entry
 * point (styler_dropout, styler_add_layernorm / styler_layernorm_bwd with drop_p > 0) then uses
 * seed + counter * odd-constant as its stream key, read on the device at execution time: a training step
 * captured in a hipGraph draws fresh masks on every replay although its host seeds are baked in.  The caller
 * increments the counter once per step, before the forward. */
int styler_set_dropout_counter(const uint64_t* counter_dev);
/* Round 6: the head of a training step as ONE launch (was: two fills + a counter increment): zeros into [a, a + a_bytes) (the flat
 * gradient, train.py:185 `zero_grad`) and [b, b + b_bytes) (the norm kernels' statistics slab), counter[0] += 1 (the dropout step
 * counter registered with styler_set_dropout_counter).  Any of the three may be NULL / 0; sizes and addresses 16-byte multiples. */
int styler_step_begin(void* a, int64_t a_bytes, void* b, int64_t b_bytes, uint64_t* counter, void* stream);

/* y = x * keep / (1-p); keep is a counter-based hash of (seed, element index): the same call on
 * dy is the backward (no mask tensor). */
int styler_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, float p,
                   uint64_t seed, void* stream);
/* out[0] += sum g^2 (fp64) -- the global gradient norm of clip_grad_norm_ (train.py:181-182). */
int styler_sumsq(const float* g, int64_t n, double* out, void* stream);
/* clip (coef = min(1, max_norm / (norm + 1e-6)), sumsq may be NULL) fused with Adam
 * (hparams.py:99-101; bias-corrected, no weight decay) over flat buffers; `step` >= 1 is Adam's own update count.
 * `grad_scale` (> 0): g holds gradient / grad_scale -- the rank SUM of the data-parallel all-reduce with
 * grad_scale = 1 / world (the mean over replicas that replaces nn.DataParallel's gather, train.py:33); the kernel
 * uses grad_scale * g and norm = grad_scale * sqrt(sumsq), so no separate division pass exists.  1.0 on one rank. */
int styler_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq,
                     float max_norm, float lr, float beta1, float beta2, float eps, int step,
                     float grad_scale, void* stream);

/* Self-test: out_swap / out_shfl [groups * 64] receive, for every 64-value group of `in`, the wave-wide sum computed by the
 * library's reduction (v_permlane32/16_swap + DPP) and by the six-step __shfl_xor butterfly; the two must be bit-identical. */
int styler_wave_sum_selftest(const float* in, float* out_swap, float* out_shfl, int groups, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STYLER_HIP_H */
