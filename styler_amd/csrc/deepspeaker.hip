// DeepSpeaker front end and glue kernels (deepspeaker/audio_ds.py:35-46,128-139, batcher.py:23-29,
// conv_models.py:28-135).  The heavy arithmetic (the framing DFT, the mel projection, every Conv2D of the ResCNN except the
// single-channel first one, the Dense layer) runs on the implicit-GEMM engine (gemm_conv.hip); this file holds what sits
// between those calls:
//
//   vad_bounds        : read_mfcc's silence trim -- the 95th percentile of |audio| (exact order statistics by radix select,
//                       numpy's linear interpolation) and the first / last sample above it, one block per utterance
//   fbank_rows        : pre-emphasis (0.97), crop to the chosen 160-frame window, lay the signal out as hop rows
//                       [B, 162, 224] (frame f = rows f..f+2: 551-sample rectangular window, hop 221, zero padded)
//   powspec           : |X|^2 / NFFT of the framing DFT -> [B*160, 516]
//   fbank_normalize   : feat == 0 -> eps, per-frame (v - mean) / max(std, 1e-12) over the 64 filters (normalize_frames),
//                       frames past the utterance's own count are zeros (pad_mfcc)
//   conv5x5s2_c1      : the first Conv2D (1 -> 64 channels, 5x5, stride 2, TF 'same') + folded BatchNorm + clipped ReLU,
//                       direct (cin = 1 is not a GEMM)
//   rows_gather_zero  : [B, Hp, W, C] row subsampling (stride-2 convs are computed at stride 1 along H) and re-zeroing of
//                       the H padding rows every conv leaves dirty
//   crelu_add         : out = min(max(a + b, 0), 20) (identity_block tail)
//   l2_normalize_rows : K.l2_normalize(y, axis=1)
#include "common.h"

#define DS_FRAME 551                 // round_half_up(0.025 * 22050)
#define DS_HOP 221                   // round_half_up(0.010 * 22050)
#define DS_ROW 224                   // hop row padded to a multiple of 8 floats (GEMM alignment)
#define DS_FRAMES 160                // NUM_FRAMES, deepspeaker/constants.py
#define DS_ROWS (DS_FRAMES + 2)
#define DS_NBIN 513
#define DS_SPEC_LD 1028
#define DS_P_LD 516
#define DS_NFILT 64

// ---- silence trim: np.percentile(|audio|, 95) + first / last index above it (audio_ds.py:36-41) ----------------------
// One block of 1024 threads per utterance.  |x| as uint32 bits is monotonic, so the k-th smallest is found by an MSB-first
// radix select (4 passes of 8 bits, histogram in LDS); numpy interpolates linearly between ranks floor(q) and floor(q) + 1
// with q = 0.95 * (n - 1).  bounds[b] = (start, end): the kept samples are audio[start:end] = audio[offsets[0]:offsets[-1]]
// (the last sample above the threshold is EXCLUDED, as in the reference slice); (0, 0) when nothing exceeds the threshold.
__global__ __launch_bounds__(1024) void vad_bounds_kernel(const float* __restrict__ wav, int64_t ldw,
                                                          const int64_t* __restrict__ wav_len, int N,
                                                          int64_t* __restrict__ bounds, float* __restrict__ thr_out) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_k, s_cnt_le, s_next;
  __shared__ int s_first, s_last;
  const int b = blockIdx.x, tid = threadIdx.x;
  int n = N;
  if (wav_len) { const int64_t l = wav_len[b]; n = l < N ? (int)l : N; }
  const float* x = wav + (int64_t)b * ldw;
  if (n <= 0) { if (tid == 0) { bounds[2 * b] = 0; bounds[2 * b + 1] = 0; if (thr_out) thr_out[b] = 0.f; } return; }
  const double q = 0.95 * (double)(n - 1);
  const unsigned int k_lo = (unsigned int)q;
  const double frac = q - (double)k_lo;
  if (tid == 0) { s_prefix = 0u; s_k = k_lo; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = tid; i < 256; i += 1024) hist[i] = 0u;
    __syncthreads();
    const unsigned int prefix = s_prefix;
    const unsigned int himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < n; i += 1024) {
      const unsigned int key = __float_as_uint(fabsf(x[i]));
      if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int k = s_k, bin = 0;
      for (; bin < 256; ++bin) { if (k < hist[bin]) break; k -= hist[bin]; }
      s_k = k;
      s_prefix = prefix | (bin << shift);
    }
    __syncthreads();
  }
  const unsigned int key_lo = s_prefix;                    // bits of the k_lo-th smallest |x|
  if (tid == 0) { s_cnt_le = 0u; s_next = 0xffffffffu; }
  __syncthreads();
  unsigned int cnt = 0, nxt = 0xffffffffu;
  for (int i = tid; i < n; i += 1024) {
    const unsigned int key = __float_as_uint(fabsf(x[i]));
    if (key <= key_lo) ++cnt; else if (key < nxt) nxt = key;
  }
  atomicAdd(&s_cnt_le, cnt);
  atomicMin(&s_next, nxt);
  __syncthreads();
  const float v_lo = __uint_as_float(key_lo);
  const float v_hi = (s_cnt_le > k_lo + 1u || s_next == 0xffffffffu) ? v_lo : __uint_as_float(s_next);
  const float thr = (float)((double)v_lo + ((double)v_hi - (double)v_lo) * frac);
  if (tid == 0) { s_first = n; s_last = -1; }
  __syncthreads();
  int first = n, last = -1;
  for (int i = tid; i < n; i += 1024)
    if (fabsf(x[i]) > thr) { if (i < first) first = i; if (i > last) last = i; }
  atomicMin(&s_first, first);
  atomicMax(&s_last, last);
  __syncthreads();
  if (tid == 0) {
    const bool any = s_last >= 0;
    bounds[2 * b] = any ? s_first : 0;
    bounds[2 * b + 1] = any ? s_last : 0;
    if (thr_out) thr_out[b] = thr;
  }
}

extern "C" int styler_ds_vad_bounds(const float* wav, int64_t ldw, const int64_t* wav_len, int B, int N, int64_t* bounds,
                                    float* thr_out, void* stream) {
  if (!wav || !bounds || B <= 0 || N <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(vad_bounds_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, wav, ldw, wav_len, N, bounds, thr_out);
  return launch_status();
}

// ---- frames of the chosen window as hop rows ------------------------------------------------------------------------
// Utterance b keeps audio[start:end] (bounds); python_speech_features.fbank pre-emphasises it (s[0] = a[0], s[k] = a[k] -
// 0.97 a[k-1]), cuts frames of 551 samples every 221 (zero padded at the end: numframes = 1 + ceil((len - 551) / 221), or 1)
// and sample_from_mfcc takes 160 consecutive frames starting at frame0[b] (batcher.py:23-29: random; here the caller's
// choice).  frame_info[b] = (numframes, first frame): written for the normalisation kernel.  crop_mode: 0 = frame0 from
// `frame0_in`, 1 = centre of the utterance.
__global__ __launch_bounds__(256) void fbank_rows_kernel(const float* __restrict__ wav, int64_t ldw,
                                                         const int64_t* __restrict__ bounds,
                                                         const int64_t* __restrict__ frame0_in, int crop_mode,
                                                         float* __restrict__ xr, int32_t* __restrict__ frame_info) {
  const int r = blockIdx.x, b = blockIdx.y, c = threadIdx.x;       // grid (DS_ROWS, B), 224 live threads
  const int64_t start = bounds[2 * b], end = bounds[2 * b + 1];
  const int len = (int)(end - start);
  const int nframes = len > DS_FRAME ? 1 + (len - DS_FRAME + DS_HOP - 1) / DS_HOP : 1;
  int f0 = 0;
  if (nframes > DS_FRAMES) {
    f0 = crop_mode == 1 ? (nframes - DS_FRAMES) / 2 : (int)frame0_in[b];
    if (f0 < 0) f0 = 0;
    if (f0 > nframes - DS_FRAMES) f0 = nframes - DS_FRAMES;
  }
  if (r == 0 && c == 0) { frame_info[2 * b] = nframes; frame_info[2 * b + 1] = f0; }
  if (c >= DS_ROW) return;
  float v = 0.f;
  if (c < DS_HOP) {
    const int k = (f0 + r) * DS_HOP + c;                           // sample index inside the kept range
    if (k < len) {
      const float* a = wav + (int64_t)b * ldw + start;
      v = k == 0 ? a[0] : a[k] - 0.97f * a[k - 1];
    }
  }
  xr[((int64_t)b * DS_ROWS + r) * DS_ROW + c] = v;
}

// ---- power spectrum: |X|^2 / 1024 per bin, one wave per frame ---------------------------------------------------------
__global__ __launch_bounds__(256) void powspec_kernel(const float* __restrict__ spec, float* __restrict__ p, int64_t frames) {
  const int lane = threadIdx.x & 63;
  const int64_t fr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (fr >= frames) return;
  const int64_t b = fr / DS_FRAMES, f = fr - b * DS_FRAMES;
  const float* sp = spec + (b * DS_ROWS + f) * DS_SPEC_LD;
  for (int k = lane; k < DS_P_LD; k += 64) {
    float v = 0.f;
    if (k < DS_NBIN) { const float re = sp[k], im = sp[DS_NBIN + k]; v = (re * re + im * im) * (1.0f / 1024.0f); }
    p[fr * DS_P_LD + k] = v;
  }
}

// ---- normalize_frames + pad_mfcc: one wave per frame, lane = filter -----------------------------------------------------
__global__ __launch_bounds__(256) void fbank_normalize_kernel(const float* __restrict__ feat, const int32_t* __restrict__ frame_info,
                                                              float* __restrict__ out, int64_t frames) {
  const int lane = threadIdx.x & 63;
  const int64_t fr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (fr >= frames) return;
  const int64_t b = fr / DS_FRAMES;
  const int f = (int)(fr - b * DS_FRAMES);
  const int nframes = frame_info[2 * b], f0 = frame_info[2 * b + 1];
  float v = feat[fr * DS_NFILT + lane];
  if (v == 0.f) v = 2.220446049250313e-16f;                       // numpy.finfo(float).eps
  const float mean = wave_sum(v) * (1.f / DS_NFILT);
  const float d = v - mean;
  const float sd = sqrtf(wave_sum(d * d) * (1.f / DS_NFILT));      // np.std: population standard deviation
  out[fr * DS_NFILT + lane] = (f0 + f < nframes) ? d / fmaxf(sd, 1e-12f) : 0.f;
}

extern "C" int64_t styler_ds_fbank_workspace_bytes(int B) {
  if (B <= 0) return 0;
  return 4 * ((int64_t)B * DS_ROWS * DS_ROW + (int64_t)B * DS_ROWS * DS_SPEC_LD + (int64_t)B * DS_FRAMES * DS_P_LD +
              (int64_t)B * DS_FRAMES * DS_NFILT) + 8 * (int64_t)B + 512;
}

int styler_conv_gemm_impl(const float* x, int64_t ldx, const void* w, const float* scale, const float* shift,
                          const float* res, int64_t ldres, float* y, int64_t ldy, int B, int L, int cin, int n,
                          int kw, int pad, int act, int prec, const int64_t* len, void* stream);

// basis: [1028, 3 * 224] (rows 0..512 Re, 513..1025 Im of the 1024-point DFT restricted to the first 551 samples, laid
// out per hop row: column j * 224 + c <-> sample j * 221 + c, pad columns zero), fp32 or bf16 per `prec`;
// fb: [64, 516] fp32 (python_speech_features.get_filterbanks(64, 1024, 22050, 0, 11025), columns 513..515 zero).
extern "C" int styler_ds_fbank(const float* wav, int64_t ldw, const int64_t* bounds, const int64_t* frame0, int crop_mode,
                               const void* basis, const float* fb, float* out, void* workspace, int B, int prec, void* stream) {
  if (!wav || !bounds || !basis || !fb || !out || !workspace || B <= 0 || (crop_mode != 0 && crop_mode != 1)) return STYLER_EINVAL;
  if (crop_mode == 0 && !frame0) return STYLER_EINVAL;
  if ((uintptr_t)workspace & 15) return STYLER_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* xr = reinterpret_cast<float*>(workspace);
  float* spec = xr + (int64_t)B * DS_ROWS * DS_ROW;
  float* pw = spec + (int64_t)B * DS_ROWS * DS_SPEC_LD;
  float* feat = pw + (int64_t)B * DS_FRAMES * DS_P_LD;
  int32_t* info = reinterpret_cast<int32_t*>(feat + (int64_t)B * DS_FRAMES * DS_NFILT);
  hipLaunchKernelGGL(fbank_rows_kernel, dim3(DS_ROWS, B), dim3(256), 0, st, wav, ldw, bounds, frame0, crop_mode, xr, info);
  int rc = styler_conv_gemm_impl(xr, DS_ROW, basis, nullptr, nullptr, nullptr, 0, spec, DS_SPEC_LD, B, DS_ROWS, DS_ROW,
                                 DS_SPEC_LD, 3, 0, STYLER_ACT_NONE, prec, nullptr, stream);
  if (rc) return rc;
  const int64_t frames = (int64_t)B * DS_FRAMES;
  hipLaunchKernelGGL(powspec_kernel, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, st, spec, pw, frames);
  rc = styler_conv_gemm_impl(pw, DS_P_LD, fb, nullptr, nullptr, nullptr, 0, feat, DS_NFILT, 1, (int)frames, DS_P_LD, DS_NFILT,
                             1, 0, STYLER_ACT_NONE, STYLER_PREC_F32, nullptr, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(fbank_normalize_kernel, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, st, feat, info, out, frames);
  return launch_status();
}

// ---- first Conv2D: [B, 160, 64] (one channel) -> [B, Hp = 83, 32, 64] (rows 1..80 live), 5x5, stride 2, TF 'same' ---------
// out[b, h, w, co] = crelu(scale[co] * sum_{i,j} x[b, 2h + i - 1, 2w + j - 1] * w[(i * 5 + j) * 64 + co] + shift[co]);
// TensorFlow's 'same' for an even extent, kernel 5, stride 2 pads 1 before and 2 after.  One thread per (position, 4 co).
__global__ __launch_bounds__(256) void conv5x5s2_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ y, int B, int H, int W) {
  const int Ho = H / 2, Wo = W / 2, Hp = Ho + 3;
  const int64_t total = (int64_t)B * Ho * Wo * 16;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(t & 15);
    int64_t p = t >> 4;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho); const int b = (int)(p / Ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int h = 2 * ho + i - 1;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int ww = 2 * wo + j - 1;
        if (ww < 0 || ww >= W) continue;
        const float v = x[((int64_t)b * H + h) * W + ww];
        const float4 k = *reinterpret_cast<const float4*>(w + (i * 5 + j) * 64 + q * 4);
        acc.x += v * k.x; acc.y += v * k.y; acc.z += v * k.z; acc.w += v * k.w;
      }
    }
    const float4 sc = *reinterpret_cast<const float4*>(scale + q * 4), sf = *reinterpret_cast<const float4*>(shift + q * 4);
    float4 o;
    o.x = fminf(fmaxf(acc.x * sc.x + sf.x, 0.f), 20.f); o.y = fminf(fmaxf(acc.y * sc.y + sf.y, 0.f), 20.f);
    o.z = fminf(fmaxf(acc.z * sc.z + sf.z, 0.f), 20.f); o.w = fminf(fmaxf(acc.w * sc.w + sf.w, 0.f), 20.f);
    *reinterpret_cast<float4*>(y + (((int64_t)b * Hp + 1 + ho) * Wo + wo) * 64 + q * 4) = o;
  }
}

extern "C" int styler_ds_conv1(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int H,
                               int W, void* stream) {
  if (!x || !w || !scale || !shift || !y || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return STYLER_EINVAL;
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * 16;
  int64_t blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(conv5x5s2_c1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, B, H, W);
  return launch_status();
}

// ---- row gather + padding rows: dst [B, Hd + 3, W, C] <- src [B, Hs_p, W, C] ---------------------------------------------
// dst row 1 + h (h < Hd) = src row src_row0 + h * step; dst rows 0, Hd + 1, Hd + 2 = 0.  With src == dst, step == 1 and
// src_row0 == 1 only the padding rows are rewritten (the live rows are left alone): the re-zeroing after a conv.
__global__ __launch_bounds__(256) void rows_gather_zero_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                               int Hd, int Hsp, int rowlen4, int src_row0, int step, int inplace) {
  const int Hp = Hd + 3;
  const int64_t total = (int64_t)B * Hp * rowlen4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % rowlen4);
    const int64_t rr = t / rowlen4;
    const int r = (int)(rr % Hp), b = (int)(rr / Hp);
    const bool live = r >= 1 && r <= Hd;
    if (live && inplace) continue;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) v = reinterpret_cast<const float4*>(src)[((int64_t)b * Hsp + src_row0 + (int64_t)(r - 1) * step) * rowlen4 + c];
    reinterpret_cast<float4*>(dst)[t] = v;
  }
}

extern "C" int styler_ds_rows(const float* src, float* dst, int B, int Hd, int Hsp, int W, int C, int src_row0, int step,
                              void* stream) {
  if (!src || !dst || B <= 0 || Hd <= 0 || Hsp <= 0 || W <= 0 || C <= 0 || ((W * C) & 3) || step <= 0 || src_row0 < 0) return STYLER_EINVAL;
  if (src_row0 + (int64_t)(Hd - 1) * step >= Hsp) return STYLER_EINVAL;
  const int inplace = src == dst;
  if (inplace && (step != 1 || src_row0 != 1 || Hsp != Hd + 3)) return STYLER_EINVAL;
  const int rowlen4 = W * C / 4;
  const int64_t total = (int64_t)B * (Hd + 3) * rowlen4;
  int64_t blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rows_gather_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, B, Hd, Hsp,
                     rowlen4, src_row0, step, inplace);
  return launch_status();
}

// ---- out = min(max(a + b, 0), 20) -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crelu_add_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                        float4* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i];
    out[i] = make_float4(fminf(fmaxf(x.x + y.x, 0.f), 20.f), fminf(fmaxf(x.y + y.y, 0.f), 20.f),
                         fminf(fmaxf(x.z + y.z, 0.f), 20.f), fminf(fmaxf(x.w + y.w, 0.f), 20.f));
  }
}

extern "C" int styler_ds_crelu_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0 || (n & 3)) return STYLER_EINVAL;
  if (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)out & 15)) return STYLER_EALIGN;
  int64_t blocks = (n / 4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(crelu_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n / 4);
  return launch_status();
}

// ---- y[r, :] = x[r, :] / max(||x[r, :]||_2, 1e-6) (K.l2_normalize: x / sqrt(max(sum x^2, 1e-12))) ---------------------------
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[(int64_t)r * C + c]; s += v * v; }
  s = wave_sum(s);
  const float inv = rsqrtf(fmaxf(s, 1e-12f));
  for (int c = lane; c < C; c += 64) y[(int64_t)r * C + c] = x[(int64_t)r * C + c] * inv;
}

extern "C" int styler_l2_normalize_rows(const float* x, float* y, int rows, int C, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0) return STYLER_EINVAL;
  hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, rows, C);
  return launch_status();
}
