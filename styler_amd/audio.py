"""On-device STFT -> mel front end (reference audio/stft.py:15-160, audio/tools.py:37-55).

`TacotronSTFT.mel_spectrogram(y)` keeps the reference's call shape (y [B, N] in [-1, 1] ->
(mel [B, 80, T], energy [B, T])) but runs wholly on the GPU: no `.cuda()`/`.cpu()` ping-pong
(stft.py:65-69).  `mel_spectrogram_cl` returns the channels-last [B, T, 80] layout the model consumes.

The mel filterbank restates librosa==0.7.2 `filters.mel(sr, n_fft, n_mels, fmin, fmax)` (Slaney scale,
area-normalised) -- a third-party function absent from the reference tree; parity at that boundary is
unpinned."""
import numpy as np
import torch
import torch.nn as nn

from . import hparams as hp
from . import ops
from ._lib import lib
from .runtime import rt


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    hz = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(hz)
    ramps = hz[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, fft_f.size))
    for i in range(n_mels):
        w[i] = np.maximum(0.0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[:, None]
    return w.astype(np.float32)


class STFT(nn.Module):
    """stft.py:15-49 (forward transform only; the inverse / Griffin-Lim is only referenced from
    commented-out code in the reference)."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, window="hann"):
        super().__init__()
        assert (filter_length, hop_length, win_length, window) == (1024, 256, 1024, "hann"), \
            "kernels are specialised for n_fft 1024 / hop 256 / periodic Hann"
        self.filter_length, self.hop_length, self.win_length = filter_length, hop_length, win_length
        four = np.fft.fft(np.eye(filter_length))
        cutoff = filter_length // 2 + 1
        basis = torch.FloatTensor(np.vstack([np.real(four[:cutoff]), np.imag(four[:cutoff])])[:, None, :])
        n = np.arange(win_length)
        basis = basis * torch.from_numpy(0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)).float()
        self.register_buffer("forward_basis", basis.float())        # [1026, 1, 1024] as in the reference


class TacotronSTFT(nn.Module):
    """stft.py:120-160."""

    def __init__(self, filter_length=hp.filter_length, hop_length=hp.hop_length, win_length=hp.win_length,
                 n_mel_channels=hp.n_mel_channels, sampling_rate=hp.sampling_rate, mel_fmin=hp.mel_fmin,
                 mel_fmax=hp.mel_fmax):
        super().__init__()
        assert n_mel_channels == 80
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        self.register_buffer("mel_basis", torch.from_numpy(
            slaney_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)))
        self._packed = {}

    def _pack(self, device, prec):
        key = (str(device), prec)
        if key not in self._packed:
            basis = torch.zeros(1028, 1024, device=device)
            basis[:1026] = self.stft_fn.forward_basis[:, 0, :].to(device)
            melb = torch.zeros(80, 516, device=device)
            melb[:, :513] = self.mel_basis.to(device)
            self._packed[key] = (ops.cast_bf16(basis) if prec == ops.PREC_BF16 else basis, melb)
        return self._packed[key]

    def mel_spectrogram_cl(self, y, want_mag=False):
        """y [B, N] fp32 on the GPU -> (mel [B, T, 80], energy [B, T][, mag [B, T, 513] view])."""
        out = self._run(y, None, want_mag, False)
        return (out["mel"], out["energy"], out["mag"]) if want_mag else (out["mel"], out["energy"])

    def _run(self, y, wav_len, want_mag, want_e_input):
        if not y.is_cuda:
            raise RuntimeError("styler_amd.audio runs on the MI355X HIP path only (no CPU fallback)")
        B, N = y.shape
        F = 1 + N // 256
        dev = y.device
        basis, melb = self._pack(dev, rt.kernel_prec())
        ws = torch.empty(int(lib.styler_stft_mel_workspace_bytes(B, N)), device=dev, dtype=torch.uint8)
        mel = torch.empty(B, F, 80, device=dev, dtype=torch.float32)
        energy = torch.empty(B, F, device=dev, dtype=torch.float32)
        mag = torch.empty(B, F, 516, device=dev, dtype=torch.float32) if want_mag else None
        e_in = torch.empty(B, F, device=dev, dtype=torch.float32) if want_e_input else None
        flen = torch.empty(B, device=dev, dtype=torch.int64) if wav_len is not None else None
        err = torch.zeros(1, device=dev, dtype=torch.int32) if rt.strict_inputs else None
        y = y if y.stride(1) == 1 else y.contiguous()
        if wav_len is not None:
            assert wav_len.dtype == torch.int64 and wav_len.is_cuda and wav_len.is_contiguous() and wav_len.numel() == B
        ops._chk(lib.styler_stft_mel_varlen(y.data_ptr(), y.stride(0), ops._ptr(wav_len), basis.data_ptr(), melb.data_ptr(),
                                            ops._ptr(mag), mel.data_ptr(), energy.data_ptr(), ops._ptr(e_in),
                                            float(hp.energy_min), float(hp.energy_max), ops._ptr(flen), ws.data_ptr(),
                                            ops._ptr(err), B, N, rt.kernel_prec(), ops._stream()), "styler_stft_mel_varlen")
        if err is not None and int(err.item()) != 0:
            raise AssertionError("mel_spectrogram: wav outside [-1, 1] (stft.py:151-152)")
        return {"mel": mel, "energy": energy, "mag": mag[..., :513] if want_mag else None, "e_input": e_in, "mel_len": flen}

    def features(self, wavs, wav_len=None):
        """A (ragged) batch of utterances -> what `STYLER.forward` consumes from the audio side (BASELINE config 5):
        wavs [B, N_max] in [-1, 1], wav_len int64 [B] on the device (None: all N_max) ->
        dict(mel [B, T, 80], energy [B, T], e_input [B, T] = utils.energy_rescaling(energy), mel_len int64 [B]).
        Every item is transformed as if alone (tools.py:37-55), frames past its end are zero padding."""
        out = self._run(wavs, wav_len, False, True)
        if out["mel_len"] is None:
            out["mel_len"] = torch.full((wavs.shape[0],), 1 + wavs.shape[1] // 256, device=wavs.device, dtype=torch.int64)
        out.pop("mag")
        return out

    def mel_spectrogram(self, y):
        """Reference layout: (mel [B, 80, T], energy [B, T])."""
        mel, energy = self.mel_spectrogram_cl(y)
        return mel.transpose(1, 2), energy


def get_mel_from_wav(audio, norm=True, stft=None):
    """audio/tools.py:37-55: one utterance [N] (int16-range floats when norm=True) ->
    (mel [80, T], energy [T], clipt)."""
    stft = stft or _default_stft(audio.device)
    clipt = False
    audio_norm = (audio / hp.max_wav_value if norm else audio).unsqueeze(0)
    if not norm:
        pre_min = torch.min(audio_norm)
        audio_norm = torch.clamp(audio_norm, -1, 1)
        clipt = bool(pre_min != torch.min(audio_norm))
    mel, energy = stft.mel_spectrogram(audio_norm.float().contiguous())
    return mel.squeeze(0), energy.squeeze(0), clipt


_STFT = {}


def _default_stft(device):
    if str(device) not in _STFT:
        _STFT[str(device)] = TacotronSTFT().to(device)
    return _STFT[str(device)]
