"""CPU: pin the oracle (oracle/styler_oracle.py) against fixtures generated from the
reference itself (tests/golden/make_golden.py).  Tolerance 1e-5 abs unless stated;
integer outputs bit-exact."""
import numpy as np
import pytest
import torch

from oracle import styler_oracle as O

T = torch.from_numpy


def close(a, b, tol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
    assert err <= tol, f"max abs err {err:.3e} > {tol}"


def test_fft_block(golden, ref_state_dict):
    g, P = golden("fft_block"), ref_state_dict
    x, lens = T(g["x"]), T(g["lens"])
    pad = O.length_mask(lens, x.shape[1])
    pre = "decoder.layer_stack.0"
    close(O.attention(P, pre + ".slf_attn", x, pad), g["attn_out"])
    close(O.pos_ffn(P, pre + ".pos_ffn", x), g["ffn_out"])
    close(O.fft_block(P, pre, x, pad), g["y"])


def test_encoder_decoder(golden, ref_state_dict):
    g, P = golden("enc_dec"), ref_state_dict
    pad = O.length_mask(T(g["lens"]), g["x"].shape[1])
    close(O.text_encoder(P, "style_modeling.style_encoder.text_encoder", T(g["text"]), pad), g["enc"])
    close(O.decoder(P, "decoder", T(g["x"]), pad), g["dec"], 2e-5)


def test_decoder_long_position_table(golden, ref_state_dict):
    from closed_form import hash_uniform
    g, P = golden("decoder_long"), ref_state_dict
    L = int(g["lens"][0])
    x = torch.from_numpy(0.1 * hash_uniform(99, L * 256).reshape(1, L, 256)).float()
    pad = O.length_mask(T(g["lens"]), L)
    close(O.sinusoid_table(L, 256)[-4:], g["pe_tail"], 0)
    close(O.decoder(P, "decoder", x, pad)[:, ::50], g["y"], 2e-5)
    with pytest.raises(RuntimeError):
        O.decoder(P, "decoder", x, pad, training=True)     # Models.py:124-125 broadcast failure


def test_position_table_matches_stored(golden):
    close(O.sinusoid_table(1001, 256)[[0, 1, 2, 499, 1000]], golden("decoder_long")["pe_stored_rows"], 0)


def test_style_predictor(golden, ref_state_dict):
    g = golden("style_predictor")
    pad = O.length_mask(T(g["lens"]), g["x"].shape[1])
    close(O.style_predictor(ref_state_dict, "style_modeling.pitch_predictor", T(g["x"]), pad), g["y"])


def test_length_regulator(golden):
    g = golden("length_regulator")
    x = T(g["x"])
    for d, ml, ko, kl in ((g["d_int"], None, "o1", "l1"), (g["d_int"], 14, "o2", "l2"),
                          (g["d_int"], 9, "o3", "l3"), (g["d_flt"], None, "o4", "l4")):
        out, mel_len = O.length_regulate(x, T(d), ml)
        close(out, g[ko], 0)
        assert mel_len.dtype == torch.int64 and np.array_equal(mel_len.numpy(), g[kl])


def test_duration_round(golden):
    g = golden("duration_round")
    for c in (1.0, 0.7, 1.3):
        close(O.rounded_duration(T(g["log_d"]), c), g[f"c{c}"], 0)


def test_mel_calibrator(golden):
    g = golden("mel_calibrator")
    close(O.mel_calibrate(T(g["x"]), T(g["mel_len"]), T(g["src_len"])), g["y"], 1e-6)


def test_quantize(golden):
    g = golden("quantize")
    idx = O.quantize_index(T(g["x"]))
    assert np.array_equal(idx.numpy(), g["idx"]) and np.array_equal(idx.numpy(), g["onehot_argmax"])
    with pytest.raises(AssertionError):
        O.quantize_index(torch.tensor([[1.5]]))


def test_bucketize_and_bins(golden, ref_state_dict):
    g, P = golden("bucketize"), ref_state_dict
    close(P["style_modeling.pitch_bins"], g["pitch_bins"], 0)
    close(P["style_modeling.energy_bins"], g["energy_bins"], 0)
    v = T(g["v"])
    assert np.array_equal(torch.bucketize(v, P["style_modeling.pitch_bins"]).numpy(), g["p_idx"])
    assert np.array_equal(torch.bucketize(v, P["style_modeling.energy_bins"]).numpy(), g["e_idx"])


def test_audio_encoder(golden, ref_state_dict):
    g, P = golden("audio_encoder"), ref_state_dict
    cat = O.encoder_input_cat(T(g["mel"]), T(g["f0_norm"]), T(g["energy_input"]), T(g["mel_aug"]))
    d, p, e, r = O.audio_encoder(P, "style_modeling.style_encoder.audio_encoder", cat,
                                 T(g["mel_len"]), T(g["src_len"]))
    for a, k in ((d, "d"), (p, "p"), (e, "e"), (r, "r")):
        close(a, g[k], 2e-5)


def test_bilstm(golden, ref_state_dict):
    g = golden("bilstm")
    close(O.bilstm2(ref_state_dict, "style_modeling.style_encoder.audio_encoder.lstm_2", T(g["x"])), g["y"])


def test_aug_classifier(golden, ref_state_dict):
    g = golden("aug_classifier")
    close(O.aug_classifier(ref_state_dict, "style_modeling.augmentation_classifier_d", T(g["x"])), g["y"])


def test_postnet(golden, ref_state_dict):
    g = golden("postnet_eval")
    close(O.postnet(ref_state_dict, "postnet", T(g["x"])), g["y"], 2e-5)
    g = golden("postnet_train")
    close(O.postnet(ref_state_dict, "postnet", T(g["x"]), training="bn_only"), g["y"], 5e-5)


def _batch(g):
    return {k[3:]: T(g[k]) for k in g.files if k.startswith("in_")}


def test_full_teacher_forced(golden, ref_state_dict):
    g = golden("full_teacher")
    b = _batch(g)
    S, Tm = b["text"].shape[1], b["mel_target"].shape[1]
    out = O.styler_forward(ref_state_dict, b["text"], b["mel_target"], b["mel_aug"], b["f0_norm"],
                           b["energy_input"], b["src_len"], b["mel_len"], b["D"], b["f0"],
                           b["energy"], S, Tm, speaker_embed=b["speaker_embed"])
    (mel, mel_n), (post, post_n), log_d, p_pred, e_pred, src_pad, mel_pad, mel_len, aug = out
    for a, k in ((mel, "mel"), (mel_n, "mel_n"), (post, "post"), (post_n, "post_n"),
                 (log_d, "log_d"), (p_pred, "p_pred"), (e_pred, "e_pred"), (aug[0], "aug_d"),
                 (aug[1], "aug_p"), (aug[2], "aug_e")):
        close(a, g[k], 5e-5)
    assert np.array_equal(src_pad.numpy(), g["src_mask"]) and np.array_equal(mel_pad.numpy(), g["mel_mask"])
    assert np.array_equal(mel_len.numpy(), g["mel_len"])


def test_full_free_running(golden, ref_state_dict):
    g, b = golden("full_free"), _batch(golden("full_teacher"))
    S = b["text"].shape[1]
    out = O.styler_forward(ref_state_dict, b["text"], b["mel_target"], b["mel_target"], b["f0_norm"],
                           b["energy_input"], b["src_len"], b["mel_len"], None, None, None, S, None,
                           speaker_embed=b["speaker_embed"], d_control=1.2, p_control=0.9,
                           e_control=1.1)
    (mel, mel_n), (post, post_n), log_d, p_pred, e_pred, _, mel_pad, mel_len, _ = out
    assert np.array_equal(mel_len.numpy(), g["mel_len"])
    assert np.array_equal(mel_pad.numpy(), g["mel_mask"])
    close(O.rounded_duration(log_d, 1.2), g["dur"], 0)
    for a, k in ((mel, "mel"), (mel_n, "mel_n"), (post, "post"), (post_n, "post_n"),
                 (log_d, "log_d"), (p_pred, "p_pred"), (e_pred, "e_pred")):
        close(a, g[k], 5e-5)


def test_train_step_losses_and_grads(golden, ref_state_dict):
    from golden.make_golden import grad_sample
    g, b = golden("train_step"), _batch(golden("full_teacher"))
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "position_enc" not in k
             and "_bins" not in k and "running_" not in k else v) for k, v in ref_state_dict.items()}
    losses = O.train_losses(P, b, training="bn_only")
    close(torch.stack(losses), g["losses"], 2e-4)
    losses[0].backward()
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in P.values() if p.grad is not None))
    assert abs(float(gn) - float(g["grad_norm"])) <= 1e-3 * float(g["grad_norm"])
    no_grad = sorted(k for k, p in P.items() if p.requires_grad and p.grad is None)
    assert no_grad == sorted(str(k) for k in g["no_grad_keys"])
    for k in g.files:
        if k.startswith("g:"):
            got = grad_sample(P[k[2:]].grad)
            scale = max(1.0, float(g["n:" + k[2:]]))
            close(got / scale, g[k] / scale, 2e-4)


def test_noam_lr(golden):
    g = golden("noam_lr")
    for n, lr in zip(g["steps"], g["lr"]):
        assert abs(O.noam_lr(int(n)) - float(lr)) <= 1e-12 * max(1.0, abs(lr)) + 1e-15


def test_stft_mel(golden):
    g = golden("stft")
    wav = T(g["wav"])
    close(O.stft_basis()[[0, 1, 7, 512, 513, 514, 700, 1025]], g["basis_rows"], 1e-6)
    close(O.stft_magnitude(wav), g["mag"], 2e-4)
    mel, energy = O.mel_spectrogram(wav)
    close(mel, g["mel"], 2e-4)
    close(energy, g["energy"], 1e-3)
    fb = O.mel_filterbank().numpy()
    # self-checks of the (unpinned) librosa-0.7.2 Slaney filterbank restatement
    assert fb.shape == (80, 513) and (fb >= 0).all()
    peaks = fb.argmax(1)
    assert (np.diff(peaks) > 0).all() and fb[:, 372:].sum() == 0     # fmax 8000 Hz -> bin 371.5
    with pytest.raises(AssertionError):
        O.mel_spectrogram(torch.full((1, 2048), 1.5))


def test_mel_filterbank_independent_derivation(golden):
    """The Slaney filterbank restatement (audio/stft.py:141-143 -> librosa_mel_fn(22050, 1024, 80, 0, 8000)) against a second,
    independent implementation: HF transformers' librosa-compatible `mel_filter_bank` (fixture + script:
    tests/golden/make_golden_melbasis.py).  A cross-check of the restated formula -- librosa 0.7.2 itself is still absent."""
    g = golden("mel_filterbank_hf")
    fb = O.mel_filterbank().double().numpy()
    ref = g["mel_basis"]
    assert ref.shape == fb.shape == (80, 513)
    assert np.abs(fb - ref).max() <= 1e-7 * ref.max()                # measured 3.5e-8 (the oracle builds it in fp32)
    # the device-side basis is this matrix: styler_amd/audio.py takes it from the same restated formula
    from styler_amd.audio import slaney_mel_filterbank
    dev_fb = np.asarray(slaney_mel_filterbank(22050, 1024, 80, 0.0, 8000.0), dtype=np.float64)
    assert np.abs(dev_fb - ref).max() <= 1e-7 * ref.max()


def test_hifigan_generator(golden, hifigan_state_dict):
    """hifigan/models.py:155-169 on the weight-normed checkpoint format; 7 mel frames -> 1792 samples."""
    g = golden("hifigan")
    wav = O.hifigan_generator(hifigan_state_dict, T(g["mel"]))
    close(wav, g["wav"], 2e-6)
    assert float(np.abs(g["wav"]).max()) < 0.5          # not saturated: the comparison is sensitive
    # folded (remove_weight_norm) checkpoints give the same result
    close(O.hifigan_generator(O.resolve_weight_norm(hifigan_state_dict), T(g["mel"])), g["wav"], 2e-6)


def test_predict_inference(golden, ref_state_dict):
    """StyleModeling.predict_inference (modules.py:285-309) for two control settings; durations `round(..) * 1.3` are
    truncated by the LengthRegulator's int() (case b)."""
    from golden.make_golden_inference import CASES, NAMES
    g, P = golden("predict_inference"), ref_state_dict
    enc = {k: T(g["in_" + k]) for k in ("text", "pitch", "energy", "duration", "speaker", "noise")}
    for tag, kw in CASES.items():
        out = O.predict_inference(P, enc["text"], enc["pitch"], enc["energy"], enc["duration"], enc["speaker"],
                                  enc["noise"], T(g["src_mask"]), None, **kw)
        for n, v in zip(NAMES, out):
            if n == "mel_mask":
                assert np.array_equal(v.numpy(), g[f"{tag}_{n}"])
            else:
                close(v, g[f"{tag}_{n}"], 2e-5)


def test_deepspeaker_restatement_sanity():
    """oracle/deepspeaker_oracle.py is self-consistent only (PARITY UNPINNED: TensorFlow / python_speech_features / weights
    absent).  What can be checked without them: the published structure of the python_speech_features filterbank, the
    framing arithmetic, and the shapes / normalisation of the ResCNN."""
    import numpy as np
    import torch
    from oracle import deepspeaker_oracle as D
    fb = D.get_filterbanks()
    assert fb.shape == (64, 513) and fb.min() >= 0 and fb.max() <= 1
    peaks = fb.argmax(axis=1)
    assert (np.diff(peaks) > 0).all() and peaks[0] >= 1 and peaks[-1] < 513        # triangular filters, increasing centres
    assert D.round_half_up(0.025 * 22050) == 551 and D.round_half_up(0.010 * 22050) == 221
    sig = np.random.RandomState(0).randn(22050).astype(np.float32)
    feat, energy = D.fbank(sig)
    assert feat.shape == (1 + int(np.ceil((22050 - 551) / 221)), 64) and (feat > 0).all()
    mf = D.mfcc_fbank(sig)
    assert np.abs(mf.mean(axis=1)).max() < 1e-4 and np.abs(mf.std(axis=1) - 1).max() < 1e-3
    g = torch.Generator().manual_seed(0)
    P = {k: (torch.rand(s, generator=g) * 0.2 + (0.9 if k.endswith(("gamma", "moving_variance")) else -0.1))
         for k, s in D.layer_shapes().items()}
    e = D.rescnn(P, torch.randn(2, 160, 64, generator=g))
    assert e.shape == (2, 512) and float((e.norm(dim=1) - 1).abs().max()) < 1e-5
    s, t = D.vad_bounds(np.concatenate([np.zeros(100), np.ones(50), np.zeros(100)]).astype(np.float32))
    assert (s, t) == (0, 0) or s >= 100                                            # only the burst can exceed the percentile


def test_mel_filterbank_published_closed_forms():
    """The librosa-0.7.2 filterbank stays PARITY UNPINNED (librosa is absent; stft.py:128-129).  What librosa documents in closed
    form is pinned here, so a regression of the restatement is visible: the Slaney scale (linear 200/3 Hz per mel below 1 kHz,
    log above with 27 mels per factor 6.4), band edges equally spaced on that scale between 0 and 8000 Hz, triangles that peak at
    their centre frequency, `2 / (f[i+2] - f[i])` area normalisation, nothing above fmax."""
    sr, n_fft, n_mels, fmax = 22050, 1024, 80, 8000.0
    fb = O.mel_filterbank(sr, n_fft, n_mels, 0.0, fmax).double().numpy()
    assert fb.shape == (80, 513) and fb.min() >= 0.0
    # the scale itself, written out independently of the oracle's helpers
    f_sp, min_log_hz, logstep = 200.0 / 3.0, 1000.0, np.log(6.4) / 27.0

    def hz2mel(f):
        return f / f_sp if f < min_log_hz else 15.0 + np.log(f / min_log_hz) / logstep

    def mel2hz(m):
        return f_sp * m if m < 15.0 else min_log_hz * np.exp(logstep * (m - 15.0))

    assert hz2mel(1000.0) == 15.0 and abs(hz2mel(6400.0) - 42.0) < 1e-12 and abs(mel2hz(42.0) - 6400.0) < 1e-9
    edges = np.array([mel2hz(m) for m in np.linspace(0.0, hz2mel(fmax), n_mels + 2)])
    assert edges[0] == 0.0 and abs(edges[-1] - fmax) < 1e-9
    assert np.allclose(np.diff(edges[edges <= 1000.0]), np.diff(edges)[0])           # linear region: equal steps in Hz
    hi = edges[edges >= 1000.0]
    assert np.allclose(hi[1:] / hi[:-1], hi[1] / hi[0])                              # log region: equal ratios
    df = sr / n_fft
    freqs = np.arange(513) * df
    for i in range(n_mels):
        lo, c, up = edges[i], edges[i + 1], edges[i + 2]
        norm = 2.0 / (up - lo)
        # the triangle evaluated at the FFT bin frequencies, scaled by the Slaney area normalisation
        tri = np.maximum(0.0, np.minimum((freqs - lo) / (c - lo), (up - freqs) / (up - c))) * norm
        assert np.abs(fb[i] - tri).max() <= 1e-6 * norm, i                          # fp32 storage of the oracle's matrix
        nz = np.nonzero(fb[i])[0]
        assert nz.size and freqs[nz[0]] > lo and freqs[nz[-1]] < up                  # support strictly inside (lo, up)
        assert abs(freqs[fb[i].argmax()] - c) < df                                   # peak at a bin adjacent to the centre
        assert fb[i].max() <= norm * (1 + 1e-6)
        if up - lo > 12 * df:                                                        # wide filters: unit area under the triangle
            assert abs(fb[i].sum() * df - 1.0) < 0.02, (i, fb[i].sum() * df)
    assert fb[:, freqs >= fmax].max() == 0.0                                         # bins 372.. (>= 8000 Hz) carry nothing
    assert np.all(np.diff(fb.argmax(axis=1)) >= 0)


def test_deepspeaker_keras_weight_dump_round_trip():
    """f2 stays PARITY UNPINNED (TensorFlow / the pretrained .h5 are absent; deepspeaker/conv_models.py:28-135).  The loader is
    pinned to the Keras layouts it claims: a synthetic `get_weights()`-style dump (Conv2D kernels [kh, kw, cin, cout], Dense
    [in, out], BatchNormalization gamma / beta / moving_mean / moving_variance) loads by name, a wrong shape is refused, the GEMM
    repacking [taps, cin, cout] -> [cout, taps * cin] agrees with an einsum over the Keras kernel, and the folded BatchNorm
    scale / shift equal Keras' inference formula with eps = 1e-3."""
    import torch
    from styler_amd import deepspeaker as DS
    rs = np.random.RandomState(3)
    shapes = DS.layer_shapes()
    dump = {k: (rs.rand(*s).astype(np.float32) + 0.5 if k.endswith(("gamma", "moving_variance"))
                else rs.randn(*s).astype(np.float32) * 0.1) for k, s in shapes.items()}
    m = DS.DeepSpeaker()
    m.load_keras_weights(dump)
    for k, v in dump.items():
        assert np.array_equal(m.weight(k).numpy(), v), k
    bad = dict(dump)
    bad["affine/kernel"] = dump["affine/kernel"].T.copy()
    with pytest.raises(ValueError):
        DS.DeepSpeaker().load_keras_weights(bad)
    P = m._pack(torch.device("cpu"), 0)                                              # fp32 packing runs on the host
    k = dump["res2_1_branch_2a/kernel"]                                              # [3, 3, 128, 128]
    x = rs.randn(3, 128).astype(np.float32)                                          # one output position: 3 taps x cin
    for i in range(3):
        want = np.einsum("wc,wco->o", x, k[i])                                       # Keras: sum over (kw, cin) of x * kernel[kh=i]
        got = P["res2_1_branch_2a"][i].numpy() @ x.reshape(-1)
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    k5 = dump["conv128-s/kernel"]                                                    # [5, 5, 64, 128]: even / odd column phases
    ev, od = P["s2"][2]
    assert np.array_equal(ev.numpy(), np.transpose(k5[2, 0::2], (2, 0, 1)).reshape(128, -1))
    assert np.array_equal(od.numpy(), np.transpose(k5[2, 1::2], (2, 0, 1)).reshape(128, -1))
    scale, shift = P["res2_1_branch_2a_ss"]
    g, b = dump["res2_1_branch_2a_bn/gamma"], dump["res2_1_branch_2a_bn/beta"]
    mu, var = dump["res2_1_branch_2a_bn/moving_mean"], dump["res2_1_branch_2a_bn/moving_variance"]
    z = rs.randn(128).astype(np.float32)                                             # conv output before bias
    keras = g * ((z + dump["res2_1_branch_2a/bias"]) - mu) / np.sqrt(var + 1e-3) + b
    assert np.abs((z * scale.numpy() + shift.numpy()) - keras).max() <= 1e-5
    assert np.array_equal(P["affine"].numpy(), dump["affine/kernel"].T * np.float32(0.1))
