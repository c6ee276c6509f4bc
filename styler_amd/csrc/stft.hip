// STFT -> mel front end (audio/stft.py:51-79,141-160; audio/tools.py:37-55).
//
//  1. frame_pad_kernel : reflect-pad 512 samples each side and lay the signal out as hop-sized rows
//                        [B, F+3, 256] (frame f = rows f..f+3), HBM-bound copy; flags |wav| > 1.
//  2. the framing conv of stft.py:65-69 = styler_conv_gemm_impl with 4 taps, pad 0, cin 256, n 1028
//     (1026 DFT rows + 2 zero rows): 2.1 MFLOP per frame on the MFMA engine, no [B,1,N+1024] unfold.
//  3. magnitude_kernel : sqrt(re^2 + im^2) -> mag [B, F, 516], energy = ||mag||_2 (one wave per frame).
//  4. mel projection + log-clamp = styler_conv_gemm_impl (kw 1, cin 516, n 80, ACT_LOGCLAMP).
#include "common.h"

int styler_conv_gemm_impl(const float* x, int64_t ldx, const void* w, const float* scale, const float* shift,
                          const float* res, int64_t ldres, float* y, int64_t ldy, int B, int L, int cin, int n,
                          int kw, int pad, int act, int prec, const int64_t* len, void* stream);

#define NFFT 1024
#define HOP 256
#define NBIN 513
#define SPEC_LD 1028
#define MAG_LD 516

// Ragged batches (wav_len != null): item b holds wav_len[b] <= N samples; the reflection happens at ITS end (each
// utterance is transformed on its own in the reference, tools.py:37-55) and hop rows past its padded signal are zeros.
__global__ __launch_bounds__(256) void frame_pad_kernel(const float* __restrict__ wav, int64_t ldw,
                                                        const int64_t* __restrict__ wav_len,
                                                        int64_t* __restrict__ frame_len, float* __restrict__ xr,
                                                        int32_t* __restrict__ err, int N, int rows) {
  // grid (rows, B); thread = sample within the hop row
  const int r = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
  int Nb = N;
  if (wav_len) { const int64_t l = wav_len[b]; Nb = l < N ? (int)l : N; }
  if (frame_len && r == 0 && c == 0) frame_len[b] = Nb > NFFT / 2 ? 1 + Nb / HOP : 0;
  int i = r * HOP + c - NFFT / 2;                 // index into the unpadded signal
  if (i < 0) i = -i;
  if (i >= Nb) i = 2 * (Nb - 1) - i;
  float v = 0.f;
  if (i >= 0 && i < Nb && r * HOP + c < Nb + NFFT) v = wav[(int64_t)b * ldw + i];
  if (err && fabsf(v) > 1.f) atomicOr(err, 1);
  xr[((int64_t)b * rows + r) * HOP + c] = v;
}

// one wave per frame: lanes stride the 513 bins
// frames at or past frame_len[b] (ragged batches) are padding: magnitude, energy and the rescaled energy are written as
// zeros there, like the zero padding of the reference's collate (utils.py:296-329).  e_scaled (optional) =
// clip((energy - e_min) / (e_max - e_min), 0, 1), utils.energy_rescaling (utils.py:410-414): the model's e_input.
__global__ __launch_bounds__(256) void magnitude_kernel(const float* __restrict__ spec, float* __restrict__ mag,
                                                        float* __restrict__ mag_out, float* __restrict__ energy,
                                                        float* __restrict__ e_scaled, float e_min, float e_max,
                                                        const int64_t* __restrict__ frame_len, int F, int rows, int B) {
  const int lane = threadIdx.x & 63;
  const int64_t fr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (fr >= (int64_t)B * F) return;
  const int b = (int)(fr / F), f = (int)(fr % F);
  const bool live = !frame_len || f < frame_len[b];
  const float* sp = spec + ((int64_t)b * rows + f) * SPEC_LD;
  float* mp = mag + fr * MAG_LD;
  float e = 0.f;
  for (int k = lane; k < MAG_LD; k += 64) {
    float m = 0.f;
    if (k < NBIN && live) {
      const float re = sp[k], im = sp[NBIN + k];
      m = sqrtf(re * re + im * im);
      e += m * m;
    }
    mp[k] = m;
    if (mag_out) mag_out[fr * MAG_LD + k] = m;
  }
  e = wave_sum(e);
  if (lane == 0) {
    const float en = sqrtf(e);
    energy[fr] = en;
    if (e_scaled) {
      const float v = (en - e_min) / (e_max - e_min);
      e_scaled[fr] = live ? fminf(fmaxf(v, 0.f), 1.f) : 0.f;
    }
  }
}

extern "C" int64_t styler_stft_mel_workspace_bytes(int B, int N) {
  if (B <= 0 || N < NFFT / 2 + 1) return 0;
  const int64_t F = 1 + N / HOP, rows = F + 3;
  return 4 * ((int64_t)B * rows * HOP + (int64_t)B * rows * SPEC_LD + (int64_t)B * F * MAG_LD) + 256;
}

extern "C" int styler_stft_mel_varlen(const float* wav, int64_t ldw, const int64_t* wav_len, const void* basis,
                                      const float* mel_basis, float* mag, float* mel, float* energy, float* e_scaled,
                                      float e_min, float e_max, int64_t* frame_len, void* workspace, int32_t* err_flag, int B,
                                      int N, int prec, void* stream) {
  if (!wav || !basis || !mel_basis || !mel || !energy || !workspace || B <= 0 || N < NFFT / 2 + 1) return STYLER_EINVAL;
  if (wav_len && !frame_len) return STYLER_EINVAL;            // ragged batches need the per-item frame counts back
  if (e_scaled && !(e_max > e_min)) return STYLER_EINVAL;
  if ((uintptr_t)workspace & 15) return STYLER_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int F = 1 + N / HOP, rows = F + 3;
  float* xr = reinterpret_cast<float*>(workspace);
  float* spec = xr + (((int64_t)B * rows * HOP + 3) & ~(int64_t)3);
  float* magp = spec + (int64_t)B * rows * SPEC_LD;
  hipLaunchKernelGGL(frame_pad_kernel, dim3(rows, B), dim3(HOP), 0, st, wav, ldw, wav_len, frame_len, xr, err_flag, N, rows);
  int rc = styler_conv_gemm_impl(xr, HOP, basis, nullptr, nullptr, nullptr, 0, spec, SPEC_LD, B, rows, HOP, SPEC_LD, 4, 0,
                                 STYLER_ACT_NONE, prec, nullptr, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(magnitude_kernel, dim3((unsigned)(((int64_t)B * F + 3) / 4)), dim3(256), 0, st, spec, magp, mag,
                     energy, e_scaled, e_min, e_max, wav_len ? frame_len : nullptr, F, rows, B);
  // rows at or past frame_len[b] are written as zeros by the GEMM epilogue (the collate's padding value, not log(1e-5))
  rc = styler_conv_gemm_impl(magp, MAG_LD, mel_basis, nullptr, nullptr, nullptr, 0, mel, 80, B, F, MAG_LD, 80, 1, 0,
                             STYLER_ACT_LOGCLAMP, STYLER_PREC_F32, wav_len ? frame_len : nullptr, stream);
  if (rc) return rc;
  return launch_status();
}

extern "C" int styler_stft_mel(const float* wav, int64_t ldw, const void* basis, const float* mel_basis, float* mag,
                               float* mel, float* energy, void* workspace, int32_t* err_flag, int B, int N, int prec,
                               void* stream) {
  return styler_stft_mel_varlen(wav, ldw, nullptr, basis, mel_basis, mag, mel, energy, nullptr, 0.f, 1.f, nullptr, workspace,
                                err_flag, B, N, prec, stream);
}
