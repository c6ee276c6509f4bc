"""CPU restatement of the DeepSpeaker speaker-embedding path.  TEST INFRASTRUCTURE ONLY: imported by
tests/, never by styler_amd/.

PARITY UNPINNED.  The reference computes the embedding with a TensorFlow/Keras model (deepspeaker/conv_models.py:28-135,
embedding.py:13-24) on features from `python_speech_features.fbank` (audio_ds.py:128-139); TensorFlow,
python_speech_features (requirements.txt: python_speech_features==0.6) and the pretrained ResCNN weights are all absent from
/root/reference and from this image, and the reference holds no test or fixture for this path.  What follows restates

  * python_speech_features 0.6 -- `sigproc.preemphasis / framesig / powspec`, `base.fbank / get_filterbanks / hz2mel /
    mel2hz` -- from its published source (the functions are ~60 lines of numpy);
  * the reference's own glue: `read_mfcc` (audio_ds.py:35-46), `normalize_frames` (138-139), `pad_mfcc` (122-125),
    `sample_from_mfcc` (batcher.py:23-29), `DeepSpeakerModel` (conv_models.py:28-135) with Keras' documented semantics
    (Conv2D 'same' = TensorFlow SAME padding, BatchNormalization epsilon 1e-3, inference mode);

and is self-consistent only: the HIP path is tested against THIS file, nothing checks this file against the reference."""
import math

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE, NUM_FRAMES, NUM_FBANKS, NFFT = 22050, 160, 64, 1024          # constants.py; calculate_nfft(22050, 1024/22050)
BN_EPS = 1e-3                                                                # keras BatchNormalization default


# ---- python_speech_features 0.6 ---------------------------------------------------------------------------------------
def round_half_up(number):
    return int(math.floor(number + 0.5))                                     # decimal ROUND_HALF_UP for positive numbers


def hz2mel(hz):
    return 2595 * np.log10(1 + hz / 700.0)


def mel2hz(mel):
    return 700 * (10 ** (mel / 2595.0) - 1)


def get_filterbanks(nfilt=NUM_FBANKS, nfft=NFFT, samplerate=SAMPLE_RATE, lowfreq=0, highfreq=None):
    highfreq = highfreq or samplerate / 2
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fbank = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fbank[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fbank[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fbank


def fbank(signal, samplerate=SAMPLE_RATE, winlen=0.025, winstep=0.01, nfilt=NUM_FBANKS, nfft=NFFT, preemph=0.97):
    """base.fbank with its defaults as mfcc_fbank calls it (rectangular window): returns (features, energies)."""
    signal = np.asarray(signal, dtype=np.float64)
    signal = np.append(signal[0], signal[1:] - preemph * signal[:-1])
    frame_len, frame_step = round_half_up(winlen * samplerate), round_half_up(winstep * samplerate)
    slen = len(signal)
    numframes = 1 if slen <= frame_len else 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((numframes - 1) * frame_step + frame_len)
    pad = np.concatenate((signal, np.zeros((padlen - slen,))))
    idx = np.arange(frame_len)[None, :] + frame_step * np.arange(numframes)[:, None]
    frames = pad[idx]
    pspec = 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))
    energy = np.sum(pspec, 1)
    energy = np.where(energy == 0, np.finfo(float).eps, energy)
    feat = np.dot(pspec, get_filterbanks(nfilt, nfft, samplerate).T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    return feat, energy


# ---- the reference's glue -----------------------------------------------------------------------------------------------
def vad_bounds(audio):
    """read_mfcc, audio_ds.py:36-41: (offsets[0], offsets[-1]) of the samples above the 95th percentile of |audio|."""
    energy = np.abs(np.asarray(audio, dtype=np.float32))
    thr = np.percentile(energy, 95)
    offsets = np.where(energy > thr)[0]
    if offsets.size == 0:
        return 0, 0
    return int(offsets[0]), int(offsets[-1])


def mfcc_fbank(signal):
    """audio_ds.py:128-139: filterbank energies, each frame standardised over its 64 values."""
    feat, _ = fbank(signal)
    return np.array([(v - np.mean(v)) / max(np.std(v), 1e-12) for v in feat], dtype=np.float32)


def read_mfcc(audio):
    s, e = vad_bounds(audio)
    return mfcc_fbank(np.asarray(audio, dtype=np.float32)[s:e])


def sample_from_mfcc(mfcc, frame0, max_length=NUM_FRAMES):
    """batcher.py:23-29 with the random start made an argument (`choice(range(0, len - max_length + 1))` there)."""
    if mfcc.shape[0] >= max_length:
        return mfcc[frame0:frame0 + max_length]
    return np.vstack((mfcc, np.zeros((max_length - len(mfcc), mfcc.shape[1]), dtype=mfcc.dtype)))


# ---- DeepSpeakerModel (conv_models.py:28-135), inference ------------------------------------------------------------------
def clipped_relu(x):
    return torch.clamp(x, 0.0, 20.0)


def _conv_bn(P, name, x, stride):
    """Keras Conv2D(padding='same') + BatchNormalization on NCHW tensors; weights in Keras layout [kh, kw, cin, cout]."""
    k = P[name + "/kernel"].permute(3, 2, 0, 1)
    kh = k.shape[2]
    if stride == 1:
        x = F.conv2d(x, k, P[name + "/bias"], padding=kh // 2)
    else:                                              # TensorFlow SAME, even extent: total = k - stride, the extra at the end
        total = kh - stride
        lo, hi = total // 2, total - total // 2
        x = F.conv2d(F.pad(x, (lo, hi, lo, hi)), k, P[name + "/bias"], stride=stride)
    g, b = P[name + "_bn/gamma"], P[name + "_bn/beta"]
    m, v = P[name + "_bn/moving_mean"], P[name + "_bn/moving_variance"]
    return (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + BN_EPS) * g[None, :, None, None] + b[None, :, None, None]


def rescnn(P, feats):
    """feats [B, 160, 64] -> L2-normalised embedding [B, 512]."""
    x = feats[:, None]                                 # [B, 1, H = frames, W = filters]
    for stage, filters in enumerate((64, 128, 256, 512), start=1):
        x = clipped_relu(_conv_bn(P, f"conv{filters}-s", x, 2))
        for block in range(3):
            base = f"res{stage}_{block}_branch"
            y = clipped_relu(_conv_bn(P, base + "_2a", x, 1))
            y = clipped_relu(_conv_bn(P, base + "_2b", y, 1))
            x = clipped_relu(y + x)
    B = x.shape[0]
    x = x.permute(0, 2, 3, 1).reshape(B, -1, 2048)     # channels-last, Reshape((-1, 2048)): [B, time, 4 * 512]
    x = x.mean(dim=1)
    x = x @ P["affine/kernel"] + P["affine/bias"]
    return x / torch.sqrt(torch.clamp((x * x).sum(dim=1, keepdim=True), min=1e-12))


def embed_utterance(P, audio, frame0=None):
    """embedding.predict_embedding (embedding.py:13-24) for one waveform; frame0 None = centre window."""
    mfcc = read_mfcc(audio)
    if frame0 is None:
        frame0 = max(0, (mfcc.shape[0] - NUM_FRAMES) // 2)
    win = sample_from_mfcc(mfcc, frame0)
    return rescnn(P, torch.from_numpy(np.ascontiguousarray(win))[None])[0]


def layer_shapes():
    """name -> shape of every weight of the inference model, Keras naming / layout."""
    shapes = {}
    cin = 1
    for stage, filters in enumerate((64, 128, 256, 512), start=1):
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
