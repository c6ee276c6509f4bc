"""DeepSpeaker ResCNN speaker embedding on the device (BASELINE config 5).

Mirrors what the reference computes with TensorFlow + python_speech_features on the host
(`deepspeaker/embedding.predict_embedding`, embedding.py:13-24 -> `read_mfcc` audio_ds.py:35-46 -> `sample_from_mfcc`
batcher.py:23-29 -> `DeepSpeakerModel` conv_models.py:28-135): waveform -> silence trim -> 64 log-free mel filterbank
energies, standardised per frame -> 160-frame window -> ResCNN (4 x [Conv 5x5 / 2 + 3 identity blocks of two 3x3 convs],
clipped ReLU, BatchNorm) -> temporal mean -> Dense(512) -> L2 normalisation.

PARITY UNPINNED: TensorFlow, python_speech_features and the pretrained weights are absent from the reference tree and from
this image; the path is checked against the self-consistent CPU restatement in oracle/deepspeaker_oracle.py only.  Weights
use the Keras names and layouts (`<layer>/kernel` [kh, kw, cin, cout], `<layer>_bn/gamma` ...), so a `get_weights()` dump of
the reference's model loads with `load_keras_weights`.

No new GEMM kernel: the framing DFT, the mel projection, every Conv2D but the first and the Dense layer are calls of the
implicit-GEMM engine.  A 2-D convolution over [B, H, W, C] (H = time, W = filters, channels last) is `kh` 1-D convolutions
along W whose source is shifted by whole rows; activations live in an H-padded layout [B, 1 + H + 2, W, C] with zero
padding rows, so the row shift is a pointer offset that is uniform over the flattened (b, h) items, and the partial sums
meet in the engine's epilogue (`ACT_RES_FIRST`: y = act(scale * (acc + res) + shift)).  Stride 2 along W reads the even /
odd column phases as strided row views; stride 2 along H is computed at stride 1 and subsampled by the row-gather kernel
(4 of the 28 convs, +13 % MACs)."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import lib
from .runtime import rt

NUM_FRAMES, NUM_FBANKS = 160, 64
_FRAME, _HOP, _ROW, _NFFT = 551, 221, 224, 1024
BN_EPS = 1e-3
_STAGES = (64, 128, 256, 512)


def _hz2mel(hz):
    return 2595 * np.log10(1 + hz / 700.0)


def _mel2hz(mel):
    return 700 * (10 ** (mel / 2595.0) - 1)


def htk_filterbanks(nfilt=NUM_FBANKS, nfft=_NFFT, samplerate=22050):
    """python_speech_features.get_filterbanks(nfilt, nfft, samplerate, 0, samplerate / 2)."""
    melpoints = np.linspace(_hz2mel(0), _hz2mel(samplerate / 2), nfilt + 2)
    bins = np.floor((nfft + 1) * _mel2hz(melpoints) / samplerate)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fb.astype(np.float32)


def layer_shapes():
    shapes, cin = {}, 1
    for stage, filters in enumerate(_STAGES, start=1):
        names = [(f"conv{filters}-s", 5, cin)] + [(f"res{stage}_{b}_branch_2{ab}", 3, filters) for b in range(3) for ab in "ab"]
        for name, k, ci in names:
            shapes[name + "/kernel"] = (k, k, ci, filters)
            shapes[name + "/bias"] = (filters,)
            for t in ("gamma", "beta", "moving_mean", "moving_variance"):
                shapes[f"{name}_bn/{t}"] = (filters,)
        cin = filters
    shapes["affine/kernel"] = (2048, 512)
    shapes["affine/bias"] = (512,)
    return shapes


class DeepSpeaker(nn.Module):
    """`embed_utterances(wavs, wav_len)` -> [B, 512] L2-normalised speaker embeddings (the `speaker_embed` input of
    `STYLER.forward`).  Inference only (the reference never trains this model)."""

    def __init__(self):
        super().__init__()
        self.w = nn.ParameterDict()
        for name, shape in layer_shapes().items():
            init = torch.ones(shape) if name.endswith(("gamma", "moving_variance")) else torch.zeros(shape)
            if name.endswith("kernel"):
                fan_in = int(np.prod(shape[:-1]))
                init = (torch.rand(shape) * 2 - 1) * (3.0 / fan_in) ** 0.5
            self.w[name.replace("/", ":")] = nn.Parameter(init, requires_grad=False)
        self._packed = {}

    # -- weights --------------------------------------------------------------------------------------------------------
    def weight(self, name):
        return self.w[name.replace("/", ":")]

    def load_keras_weights(self, weights):
        """`weights`: dict name -> array in the Keras layout (`layer_shapes()`), e.g. built from `model.get_weights()`."""
        with torch.no_grad():
            for name, shape in layer_shapes().items():
                v = torch.as_tensor(np.asarray(weights[name]), dtype=torch.float32)
                if tuple(v.shape) != tuple(shape):
                    raise ValueError(f"{name}: expected {shape}, got {tuple(v.shape)}")
                self.weight(name).copy_(v)
        self._packed = {}

    def _fold(self, name, dev):
        """BatchNorm (inference) + conv bias as the GEMM epilogue's scale / shift."""
        g, b = self.weight(name + "_bn/gamma"), self.weight(name + "_bn/beta")
        m, v = self.weight(name + "_bn/moving_mean"), self.weight(name + "_bn/moving_variance")
        scale = g / torch.sqrt(v + BN_EPS)
        shift = b + (self.weight(name + "/bias") - m) * scale
        return scale.to(dev).contiguous(), shift.to(dev).contiguous()

    def _mat(self, k, dev, bf16):
        """[taps, cin, cout] (Keras order) -> kernel layout [cout, taps * cin]."""
        taps, cin, cout = k.shape
        m = k.permute(2, 0, 1).reshape(cout, taps * cin).contiguous().to(dev)
        return ops.cast_bf16(m) if bf16 else m

    def _pack(self, dev, prec):
        key = (str(dev), prec)
        if key in self._packed:
            return self._packed[key]
        bf16 = prec == ops.PREC_BF16
        P = {}
        # framing DFT restricted to the 551 samples of a frame, laid out per hop row (3 rows of 224 columns)
        n = np.arange(_FRAME)
        kk = np.arange(_NFFT // 2 + 1)
        ang = 2.0 * np.pi * np.outer(kk, n) / _NFFT
        basis = np.zeros((1028, 3 * _ROW), dtype=np.float64)
        cols = (n // _HOP) * _ROW + (n % _HOP)
        basis[:513, cols] = np.cos(ang)
        basis[513:1026, cols] = -np.sin(ang)
        bt = torch.from_numpy(basis.astype(np.float32)).to(dev)
        P["basis"] = ops.cast_bf16(bt) if bf16 else bt
        fb = torch.zeros(NUM_FBANKS, 516)
        fb[:, :513] = torch.from_numpy(htk_filterbanks())
        P["fb"] = fb.to(dev)
        # first conv: direct kernel, weight [25, 64]
        P["conv1_w"] = self.weight("conv64-s/kernel").reshape(25, 64).contiguous().to(dev)
        P["conv1_ss"] = self._fold("conv64-s", dev)
        for stage, filters in enumerate(_STAGES, start=1):
            if stage > 1:
                k = self.weight(f"conv{filters}-s/kernel")                 # [5, 5, cin, cout]
                P[f"s{stage}"] = [(self._mat(k[i, 0::2], dev, bf16), self._mat(k[i, 1::2], dev, bf16)) for i in range(5)]
                P[f"s{stage}_ss"] = self._fold(f"conv{filters}-s", dev)
            for b in range(3):
                for ab in "ab":
                    name = f"res{stage}_{b}_branch_2{ab}"
                    k = self.weight(name + "/kernel")                      # [3, 3, c, c]
                    P[name] = [self._mat(k[i], dev, bf16) for i in range(3)]
                    P[name + "_ss"] = self._fold(name, dev)
        # temporal mean over the 10 live rows folded into the Dense weight
        wd = (self.weight("affine/kernel").t() * 0.1).contiguous().to(dev)
        P["affine"] = ops.cast_bf16(wd) if bf16 else wd
        P["affine_b"] = self.weight("affine/bias").to(dev).contiguous()
        self._packed[key] = P
        return P

    # -- front end --------------------------------------------------------------------------------------------------------
    def vad_bounds(self, wavs, wav_len=None, want_threshold=False):
        B, N = wavs.shape
        bounds = torch.empty(B, 2, device=wavs.device, dtype=torch.int64)
        thr = torch.empty(B, device=wavs.device, dtype=torch.float32) if want_threshold else None
        ops._chk(lib.styler_ds_vad_bounds(ops._f32(wavs).data_ptr(), wavs.stride(0), ops._ptr(wav_len), B, N,
                                          bounds.data_ptr(), ops._ptr(thr), ops._stream()), "styler_ds_vad_bounds")
        return (bounds, thr) if want_threshold else bounds

    def fbank_window(self, wavs, bounds, frame0=None):
        """-> [B, 160, 64]: the standardised filterbank features of the chosen 160-frame window (frame0 int64 [B] on the
        device; None = the centre window -- the reference draws it at random, batcher.py:25)."""
        B = wavs.shape[0]
        P = self._pack(wavs.device, rt.kernel_prec())
        ws = torch.empty(int(lib.styler_ds_fbank_workspace_bytes(B)), device=wavs.device, dtype=torch.uint8)
        out = torch.empty(B, NUM_FRAMES, NUM_FBANKS, device=wavs.device, dtype=torch.float32)
        ops._chk(lib.styler_ds_fbank(wavs.data_ptr(), wavs.stride(0), bounds.data_ptr(), ops._ptr(frame0),
                                     0 if frame0 is not None else 1, P["basis"].data_ptr(), P["fb"].data_ptr(),
                                     out.data_ptr(), ws.data_ptr(), B, rt.kernel_prec(), ops._stream()), "styler_ds_fbank")
        return out

    # -- ResCNN -----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _rows(src, dst, B, Hd, Hsp, W, C, src_row0, step):
        ops._chk(lib.styler_ds_rows(src.data_ptr(), dst.data_ptr(), B, Hd, Hsp, W, C, src_row0, step, ops._stream()),
                 "styler_ds_rows")

    def _conv3(self, x, mats, ss, B, H, W, C, prec):
        """3x3, stride 1, 'same', + BatchNorm + clipped ReLU on the H-padded layout [B, H + 3, W, C]."""
        Hp = H + 3
        M = B * Hp - 2
        out = torch.empty_like(x)
        xi, oi = x.view(B * Hp, W, C), out.view(B * Hp, W, C)
        o = oi[1:1 + M]
        for i in range(3):
            last = i == 2
            ops.conv_gemm_pad(xi[i:i + M], mats[i], ss[1] if last else None, kw=3, pad=1, prec=prec, out=o,
                              res=o if i else None, scale=ss[0] if last else None,
                              act=(ops.ACT_CRELU | ops.ACT_RES_FIRST) if last else ops.ACT_NONE)
        self._rows(out, out, B, H, Hp, W, C, 1, 1)                           # the padding rows caught partial garbage
        return out

    def _conv5s2(self, x, mats, ss, B, H, W, C, Co, prec):
        """5x5, stride 2, TensorFlow 'same' (pad 1 before / 2 after) + BatchNorm + clipped ReLU:
        [B, H + 3, W, C] -> [B, H/2 + 3, W/2, Co].  Computed at stride 1 along H, subsampled by the row gather."""
        Hp, Wo = H + 3, W // 2
        M = B * Hp - 4
        tmp = torch.empty(B * Hp, Wo, Co, device=x.device, dtype=torch.float32)
        xi = x.view(B * Hp, W, C)
        o = tmp[0:M]
        first = True
        for i in range(5):
            src = xi[i:i + M]
            for phase, (kw, pad) in ((1, (3, 1)), (0, (2, 0))):              # odd columns: taps j = 0, 2, 4; even: j = 1, 3
                last = i == 4 and phase == 0
                ops.conv_gemm_pad(src[:, phase::2, :], mats[i][0 if phase == 1 else 1], ss[1] if last else None, kw=kw,
                                  pad=pad, prec=prec, out=o, res=None if first else o, scale=ss[0] if last else None,
                                  act=(ops.ACT_CRELU | ops.ACT_RES_FIRST) if last else ops.ACT_NONE)
                first = False
        out = torch.empty(B, H // 2 + 3, Wo, Co, device=x.device, dtype=torch.float32)
        self._rows(tmp, out, B, H // 2, Hp, Wo, Co, 0, 2)
        return out

    def rescnn(self, feats):
        """feats [B, 160, 64] -> [B, 512] L2-normalised."""
        if not feats.is_cuda:
            raise RuntimeError("styler_amd.deepspeaker runs on the MI355X HIP path only (no CPU fallback)")
        B, H, W = feats.shape
        assert (H, W) == (NUM_FRAMES, NUM_FBANKS)
        dev, prec = feats.device, rt.kernel_prec()
        P = self._pack(dev, prec)
        H, W = H // 2, W // 2
        x = torch.empty(B, H + 3, W, 64, device=dev, dtype=torch.float32)
        ops._chk(lib.styler_ds_conv1(ops._f32(feats.contiguous()).data_ptr(), P["conv1_w"].data_ptr(), P["conv1_ss"][0].data_ptr(),
                                     P["conv1_ss"][1].data_ptr(), x.data_ptr(), B, 2 * H, 2 * W, ops._stream()), "styler_ds_conv1")
        self._rows(x, x, B, H, H + 3, W, 64, 1, 1)
        C = 64
        for stage, filters in enumerate(_STAGES, start=1):
            if stage > 1:
                x = self._conv5s2(x, P[f"s{stage}"], P[f"s{stage}_ss"], B, H, W, C, filters, prec)
                H, W, C = H // 2, W // 2, filters
            for b in range(3):
                na, nb = f"res{stage}_{b}_branch_2a", f"res{stage}_{b}_branch_2b"
                y = self._conv3(x, P[na], P[na + "_ss"], B, H, W, C, prec)
                y = self._conv3(y, P[nb], P[nb + "_ss"], B, H, W, C, prec)
                z = torch.empty_like(x)
                ops._chk(lib.styler_ds_crelu_add(y.data_ptr(), x.data_ptr(), z.data_ptr(), x.numel(), ops._stream()),
                         "styler_ds_crelu_add")
                x = z
        # [B, 10 + 3, 4, 512]: rows sum (the padding rows are zero) -> mean folded into the Dense weight -> L2 normalise
        pooled = ops.rowsum(x.view(B, H + 3, W * C))
        emb = ops.conv_gemm(pooled.view(1, B, W * C), P["affine"], P["affine_b"], n=512, prec=prec).view(B, 512)
        out = torch.empty_like(emb)
        ops._chk(lib.styler_l2_normalize_rows(emb.data_ptr(), out.data_ptr(), B, 512, ops._stream()), "styler_l2_normalize_rows")
        return out

    def embed_utterances(self, wavs, wav_len=None, frame0=None):
        """embedding.predict_embedding (embedding.py:13-24) for a (ragged) batch of waveforms."""
        wavs = wavs if wavs.stride(1) == 1 else wavs.contiguous()
        bounds = self.vad_bounds(wavs, wav_len)
        return self.rescnn(self.fbank_window(wavs, bounds, frame0))

    forward = embed_utterances
