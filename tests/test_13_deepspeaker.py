"""DeepSpeaker speaker embedding on the device against the CPU restatement in oracle/deepspeaker_oracle.py.
PARITY UNPINNED (TensorFlow, python_speech_features and the weights are absent: see that file's header): these tests show
that the HIP path computes what the restatement defines, nothing more."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _weights(seed=11):
    from oracle import deepspeaker_oracle as D
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shape in D.layer_shapes().items():
        u = torch.rand(shape, generator=g) * 2 - 1
        if name.endswith("kernel"):
            P[name] = u * (3.0 / int(np.prod(shape[:-1]))) ** 0.5 * 1.6
        elif name.endswith("gamma"):
            P[name] = 1.0 + 0.2 * u
        elif name.endswith("moving_variance"):
            P[name] = 1.0 + 0.4 * u
        else:
            P[name] = 0.1 * u
    return P


def _speechlike(B, N, seed, lens=None):
    """Noise bursts between silences, so that the percentile trim has something to cut."""
    g = torch.Generator().manual_seed(seed)
    wav = torch.zeros(B, N)
    for b in range(B):
        n = int(lens[b]) if lens is not None else N
        lo = int(torch.randint(500, 4000, (1,), generator=g))
        hi = n - int(torch.randint(500, 4000, (1,), generator=g))
        env = 0.05 + 0.45 * torch.rand(1, generator=g)
        wav[b, lo:hi] = (torch.rand(hi - lo, generator=g) - 0.5) * 2 * env
        wav[b, :n] += 1e-3 * (torch.rand(n, generator=g) - 0.5)
    return wav


def test_vad_bounds_and_fbank_window(dev):
    from oracle import deepspeaker_oracle as D
    from styler_amd.deepspeaker import DeepSpeaker
    lens = torch.tensor([66150, 88200, 40000, 30000, 77175])        # 30000 samples -> fewer than 160 frames after the trim
    wav = _speechlike(5, 88200, 3, lens)
    ds = DeepSpeaker().to(dev)
    bounds, thr = ds.vad_bounds(wav.to(dev), lens.to(dev), want_threshold=True)
    bounds, thr = bounds.cpu(), thr.cpu()
    for b in range(5):
        a = wav[b, :int(lens[b])].numpy()
        ref_thr = float(np.percentile(np.abs(a), 95))
        assert abs(float(thr[b]) - ref_thr) <= 1e-6 * max(1.0, ref_thr)
        off = np.where(np.abs(a) > float(thr[b]))[0]                 # same threshold bits -> identical comparison
        assert (int(bounds[b, 0]), int(bounds[b, 1])) == (int(off[0]), int(off[-1]))
    frame0 = torch.tensor([7, 0, 3, 0, 100])
    for f0 in (None, frame0):
        got = ds.fbank_window(wav.to(dev), bounds.to(dev), None if f0 is None else f0.to(dev)).cpu()
        for b in range(5):
            mf = D.mfcc_fbank(wav[b, int(bounds[b, 0]):int(bounds[b, 1])].numpy())
            start = max(0, (mf.shape[0] - 160) // 2) if f0 is None else min(int(f0[b]), max(0, mf.shape[0] - 160))
            ref = torch.from_numpy(D.sample_from_mfcc(mf, start))
            if b == 3:
                assert mf.shape[0] < 160 and float(got[b, mf.shape[0]:].abs().max()) == 0.0        # pad_mfcc: zero frames
            # standardised features are O(1); the fp32 1024-point DFT of white noise keeps ~1e-4 relative in the power
            assert float((got[b] - ref).abs().max()) <= 2e-3, (b, float((got[b] - ref).abs().max()))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_rescnn_matches_restatement(dev, prec):
    from oracle import deepspeaker_oracle as D
    from styler_amd import rt
    from styler_amd.deepspeaker import DeepSpeaker
    P = _weights()
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(6, 160, 64, generator=g)
    feats[4, 100:] = 0.0                                            # a zero-padded (short) utterance
    ref = D.rescnn(P, feats)
    ds = DeepSpeaker()
    ds.load_keras_weights(P)
    ds = ds.to(dev)
    rt.set_precision(prec)
    try:
        got = ds.rescnn(feats.to(dev)).cpu()
    finally:
        rt.set_precision("fp32")
    assert got.shape == (6, 512)
    assert float((got.norm(dim=1) - 1).abs().max()) <= 1e-5
    err = float((got - ref).abs().max())
    assert err <= (2e-5 if prec == "fp32" else 5e-3), err          # unit-norm vectors: entries ~ 0.04
    cos = (got * ref).sum(dim=1)
    assert float(cos.min()) >= (1 - 1e-6 if prec == "fp32" else 0.999)


def test_embed_utterances_and_config5_front_end(dev, ref_state_dict):
    """wav batch -> embeddings == restatement per utterance; and the full config-5 front end (mel / energy / e_input +
    speaker embedding from the same wavs) drives STYLER.forward."""
    from oracle import deepspeaker_oracle as D
    from styler_amd import STYLER, rt
    from styler_amd.deepspeaker import DeepSpeaker
    from styler_amd.pipeline import WavFrontEnd, forward_from_wavs
    P = _weights(12)
    lens = torch.tensor([66150, 88200, 77175, 66150])
    wav = _speechlike(4, 88200, 9, lens)
    ds = DeepSpeaker()
    ds.load_keras_weights(P)
    ds = ds.to(dev)
    emb = ds.embed_utterances(wav.to(dev), lens.to(dev)).cpu()
    for b in range(4):
        ref = D.embed_utterance(P, wav[b, :int(lens[b])].numpy())
        assert float((emb[b] - ref).abs().max()) <= 2e-4, (b, float((emb[b] - ref).abs().max()))

    B, S = 4, 24
    g = torch.Generator().manual_seed(2)
    fe = WavFrontEnd(speaker_encoder=ds).to(dev)
    m = STYLER()
    m.load_state_dict(ref_state_dict)
    m = m.to(dev).eval()
    src_len = torch.tensor([24, 20, 22, 18])
    text = torch.randint(1, 152, (B, S), generator=g) * (torch.arange(S)[None] < src_len[:, None])
    T = 1 + 88200 // 256
    mel_len = 1 + lens // 256
    valid = (torch.arange(T)[None] < mel_len[:, None])
    p_norm = torch.rand(B, T, generator=g) * valid
    f0 = (80.0 + 300.0 * torch.rand(B, T, generator=g)) * valid
    D = torch.zeros(B, S, dtype=torch.long)                          # teacher-forced: durations summing to the frame counts
    for b in range(B):
        s, t = int(src_len[b]), int(mel_len[b])
        D[b, :s] = t // s
        D[b, :t % s] += 1
    strict, rt.strict_inputs = rt.strict_inputs, False
    try:
        with torch.no_grad():
            out, feats = forward_from_wavs(m, fe, wav.to(dev), lens.to(dev), text.to(dev), src_len.to(dev), p_norm.to(dev),
                                           d_target=D.to(dev), p_target=f0.to(dev))
    finally:
        rt.strict_inputs = strict
    assert torch.equal(feats["mel_len"].cpu(), mel_len) and feats["speaker_embed"].shape == (B, 512)
    assert float((feats["speaker_embed"].cpu() - emb).abs().max()) <= 1e-6
    mel = out[0][0]
    assert mel.shape == (B, T, 80) and torch.isfinite(mel).all() and torch.equal(out[7].cpu(), mel_len)
    pad = ~valid
    assert float(out[2].abs().max()) > 0 and float(mel[pad.to(dev)].abs().max()) < 1e3      # finite everywhere, padding included
