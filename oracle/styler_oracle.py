"""CPU oracle for the STYLER hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional, plain PyTorch-CPU (fp32, eager) restatement of the reference's
`styler.STYLER.forward` / `loss.STYLERLoss` path.  It consumes a *reference-format*
state dict (the 328 keys of `STYLER().state_dict()`, conv weights `[C_out, C_in, k]`)
and plain tensors; it holds no nn.Module state.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module, and only as the checker / the timed CPU baseline.  Nothing under
`styler_amd/` may import it (tests/test_01_host_cpu.py::test_product_never_imports_oracle enforces that).

Pinning: the reference ships no tests or golden vectors, so this
oracle is pinned against outputs of the reference itself, generated in the build
container by `tests/golden/make_golden.py` (imports /root/reference, closed-form
weights) and committed as `tests/golden/*.npz`; `tests/test_00_oracle_golden.py` checks
every function below against them.

Every function cites the reference file:line it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# ----------------------------------------------------------------------------------
# hyper-parameters of the path (hparams.py:21-71, 93-101)
# ----------------------------------------------------------------------------------
HIDDEN = 256            # encoder_hidden / decoder_hidden (hparams.py:45,49)
N_HEAD = 4              # encoder_head / decoder_head (hparams.py:44,48)
D_INNER = 1024          # fft_conv1d_filter_size (hparams.py:50)
ENC_LAYERS = 2          # hparams.py:43
DEC_LAYERS = 4          # hparams.py:47
MAX_SEQ_LEN = 1000      # hparams.py:58
N_MEL = 80              # hparams.py:37
N_BINS = 256            # hparams.py:34
LOG_OFFSET = 1.0        # hparams.py:104
NECK_D, NECK_P, NECK_E, NECK_R = 80, 64, 64, 64     # hparams.py:63-67
ENC_D, ENC_P, ENC_E, ENC_R = 256, 320, 320, 256     # hparams.py:69-72
DIM_F0 = DIM_EN = 257   # hparams.py:74-75
GN_CH = 16              # va_chs_grp (hparams.py:76)
DAT_WEIGHT = 1.0        # hparams.py:60


# ----------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------
def length_mask(lengths: Tensor, max_len: Optional[int] = None) -> Tensor:
    """True = padded position.  utils.py:223-232 (get_mask_from_lengths)."""
    if max_len is None:
        max_len = int(lengths.max().item())
    pos = torch.arange(int(max_len), device=lengths.device)
    return pos[None, :] >= lengths[:, None]


def sinusoid_table(n_position: int, d_hid: int) -> Tensor:
    """transformer/Models.py:11-30: float64 numpy angles, sin on even / cos on odd
    channels, cast to float32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    angle = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)
    table = np.empty_like(angle)
    table[:, 0::2] = np.sin(angle[:, 0::2])
    table[:, 1::2] = np.cos(angle[:, 1::2])
    return torch.from_numpy(table).float()


def _linear(P: Params, pre: str, x: Tensor) -> Tensor:
    return F.linear(x, P[pre + ".weight"], P[pre + ".bias"])


def _conv_cl(P: Params, pre: str, x: Tensor, pad: int) -> Tensor:
    """Conv1d on a channels-last [B, L, C] tensor (the reference transposes to
    [B, C, L] around every nn.Conv1d: SubLayers.py:83-85, modules.py:502-507)."""
    y = F.conv1d(x.transpose(1, 2), P[pre + ".weight"], P[pre + ".bias"], padding=pad)
    return y.transpose(1, 2)


def _dropout(x: Tensor, p: float, training) -> Tensor:
    """`training` is False (eval), True (train), or "bn_only" (train-mode BatchNorm /
    position-table rules but dropout disabled -- the mode the gradient fixtures use, since
    the reference's dropout RNG stream cannot be reproduced)."""
    return F.dropout(x, p, True) if (training is True and p > 0.0) else x


# ----------------------------------------------------------------------------------
# FFT block (transformer/)
# ----------------------------------------------------------------------------------
def attention(P: Params, pre: str, x: Tensor, key_pad: Tensor, training=False, p_drop=0.2
              ) -> Tensor:
    """MultiHeadAttention.forward, SubLayers.py:31-61 + ScaledDotProductAttention,
    Modules.py:14-25.  q = k = v = x (self-attention, Layers.py:27-28).  Keys at padded
    positions get -inf before the softmax; query rows are not masked here."""
    B, L, _ = x.shape
    dk = HIDDEN // N_HEAD
    q = _linear(P, pre + ".w_qs", x).view(B, L, N_HEAD, dk).permute(0, 2, 1, 3)
    k = _linear(P, pre + ".w_ks", x).view(B, L, N_HEAD, dk).permute(0, 2, 1, 3)
    v = _linear(P, pre + ".w_vs", x).view(B, L, N_HEAD, dk).permute(0, 2, 1, 3)
    score = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dk)   # temperature = sqrt(d_k)
    score = score.masked_fill(key_pad[:, None, None, :], float("-inf"))
    prob = torch.softmax(score, dim=-1)
    ctx = torch.matmul(prob, v).permute(0, 2, 1, 3).reshape(B, L, HIDDEN)
    out = _dropout(_linear(P, pre + ".fc", ctx), p_drop, training)
    return F.layer_norm(out + x, (HIDDEN,), P[pre + ".layer_norm.weight"],
                        P[pre + ".layer_norm.bias"])


def pos_ffn(P: Params, pre: str, x: Tensor, training=False, p_drop=0.2) -> Tensor:
    """PositionwiseFeedForward.forward, SubLayers.py:81-89: Conv1d(k=9,p=4) -> ReLU ->
    Conv1d(k=1) -> dropout -> LayerNorm(out + residual)."""
    h = F.relu(_conv_cl(P, pre + ".w_1", x, 4))
    h = _conv_cl(P, pre + ".w_2", h, 0)
    h = _dropout(h, p_drop, training)
    return F.layer_norm(h + x, (HIDDEN,), P[pre + ".layer_norm.weight"],
                        P[pre + ".layer_norm.bias"])


def fft_block(P: Params, pre: str, x: Tensor, pad: Tensor, training=False) -> Tensor:
    """FFTBlock.forward, Layers.py:26-34: both sub-layers are followed by
    masked_fill(pad, 0)."""
    x = attention(P, pre + ".slf_attn", x, pad, training).masked_fill(pad[..., None], 0.0)
    x = pos_ffn(P, pre + ".pos_ffn", x, training).masked_fill(pad[..., None], 0.0)
    return x


def _position(P: Params, pre: str, L: int, training: bool) -> Tensor:
    """Models.py:69-74 / 120-125: eval mode regenerates the table when L > max_seq_len;
    train mode slices the stored [1, 1001, 256] parameter (and fails to broadcast when
    L > 1001)."""
    if (not training) and L > MAX_SEQ_LEN:
        return sinusoid_table(L, HIDDEN)[None]
    tab = P[pre + ".position_enc"][:, :L]
    if tab.shape[1] != L:
        raise RuntimeError(f"sequence length {L} exceeds position table {tab.shape[1]} in train mode")
    return tab


def text_encoder(P: Params, pre: str, text: Tensor, src_pad: Tensor, training=False) -> Tensor:
    """Encoder.forward, Models.py:60-84."""
    x = F.embedding(text, P[pre + ".src_word_emb.weight"], padding_idx=0)
    x = x + _position(P, pre, text.shape[1], training)
    for i in range(ENC_LAYERS):
        x = fft_block(P, f"{pre}.layer_stack.{i}", x, src_pad, training)
    return x


def decoder(P: Params, pre: str, x: Tensor, mel_pad: Tensor, training=False) -> Tensor:
    """Decoder.forward, Models.py:111-135."""
    x = x + _position(P, pre, x.shape[1], training)
    for i in range(DEC_LAYERS):
        x = fft_block(P, f"{pre}.layer_stack.{i}", x, mel_pad, training)
    return x


def postnet(P: Params, pre: str, mel: Tensor, training=False, update_stats=None) -> Tensor:
    """PostNet.forward, Layers.py:121-130: 5x [Conv1d(k=5,p=2) -> BatchNorm1d], tanh on
    all but the last, F.dropout(0.5) after each.  Train mode uses batch statistics over
    (B, T) including padded frames (torch BatchNorm1d semantics); eval mode the running
    statistics."""
    x = mel
    for i in range(5):
        cp = f"{pre}.convolutions.{i}"
        x = _conv_cl(P, cp + ".0.conv", x, 2)
        xt = x.transpose(1, 2)
        xt = F.batch_norm(xt, P[cp + ".1.running_mean"].clone(), P[cp + ".1.running_var"].clone(),
                          P[cp + ".1.weight"], P[cp + ".1.bias"], training=bool(training),
                          momentum=0.1, eps=1e-5)
        x = xt.transpose(1, 2)
        if i < 4:
            x = torch.tanh(x)
        x = _dropout(x, 0.5, training)
    return x


def decode(P: Params, x: Tensor, mel_pad: Tensor, training=False) -> Tuple[Tensor, Tensor]:
    """STYLER.decode, styler.py:29-37."""
    h = decoder(P, "decoder", x, mel_pad, training)
    mel = _linear(P, "mel_linear", h)
    return mel, postnet(P, "postnet", mel, training) + mel


# ----------------------------------------------------------------------------------
# style encoders (modules.py)
# ----------------------------------------------------------------------------------
def quantize_index(x: Tensor, num_bins: int = 256) -> Tensor:
    """utils.py:417-429 (quantize_1D_torch), index half: 0 where x <= 0 else
    round(x * 255) + 1 (torch.round = half-to-even).  Raises like the reference's
    assert when x is outside [0, 1]."""
    uv = x <= 0
    xc = torch.where(uv, torch.zeros_like(x), x)
    if not (bool((xc >= 0).all()) and bool((xc <= 1).all())):
        raise AssertionError("quantize_1D_torch: input outside [0, 1]")
    idx = torch.round(xc * (num_bins - 1)) + 1
    idx = torch.where(uv, torch.zeros_like(idx), idx)
    return idx.long()


def encoder_input_cat(mel: Tensor, p_norm: Tensor, e_input: Tensor, mel_aug: Tensor) -> Tensor:
    """StyleEncoder.encoder_input_cat, modules.py:218-223, kept channels-last
    [B, T, 80+257+257+80] (the reference transposes to [B, 674, T])."""
    f0_1h = F.one_hot(quantize_index(p_norm), DIM_F0).to(mel.dtype)
    en_1h = F.one_hot(quantize_index(e_input), DIM_EN).to(mel.dtype)
    return torch.cat((mel, f0_1h, en_1h, mel_aug), dim=-1)


def segment_sizes(src: int, tgt: int):
    """utils.py:351-352 (get_scale)."""
    return [src // tgt + (1 if i < src % tgt else 0) for i in range(tgt)]


def mel_calibrate(x: Tensor, mel_len: Tensor, src_len: Tensor) -> Tensor:
    """utils.py:355-384 (mel_calibrator): per item resample the frame axis from mel_len
    to src_len by near-equal segment means (compression) or repeats (expansion); re-pad
    to the batch max of src_len with zeros."""
    B, _, C = x.shape
    S = int(src_len.max().item())
    out = x.new_zeros(B, S, C)
    for b in range(B):
        ml, sl = int(mel_len[b].item()), int(src_len[b].item())
        m = x[b, :ml]
        if ml == sl:
            out[b, :sl] = m
        elif ml > sl:
            start = 0
            for i, n in enumerate(segment_sizes(ml, sl)):
                out[b, i] = m[start:start + n].sum(dim=0) / n
                start += n
        else:
            reps = torch.tensor(segment_sizes(sl, ml))
            out[b, :sl] = torch.repeat_interleave(m, reps, dim=0)
    return out


def lstm_direction(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor,
                   reverse: bool) -> Tensor:
    """One direction of one nn.LSTM layer (gate order i, f, g, o), zero initial state,
    run over the whole padded sequence (no packing: modules.py:179-182)."""
    B, L, _ = x.shape
    H = w_hh.shape[1]
    gx = F.linear(x, w_ih, b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * L
    steps = range(L - 1, -1, -1) if reverse else range(L)
    for t in steps:
        g = gx[:, t] + F.linear(h, w_hh)
        i, f, gg, o = g.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def bilstm2(P: Params, pre: str, x: Tensor) -> Tensor:
    """nn.LSTM(in, H, 2, batch_first=True, bidirectional=True)(x)[0], modules.py:117-162."""
    for layer in range(2):
        outs = []
        for sfx, rev in (("", False), ("_reverse", True)):
            outs.append(lstm_direction(
                x, P[f"{pre}.weight_ih_l{layer}{sfx}"], P[f"{pre}.weight_hh_l{layer}{sfx}"],
                P[f"{pre}.bias_ih_l{layer}{sfx}"], P[f"{pre}.bias_hh_l{layer}{sfx}"], rev))
        x = torch.cat(outs, dim=-1)
    return x


def audio_encoder(P: Params, pre: str, enc_cat: Tensor, mel_len: Tensor, src_len: Tensor
                  ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """AudioEncoder.forward, modules.py:164-201.  enc_cat is channels-last [B, T, 674].
    GroupNorm statistics span 16 channels x the whole padded T (modules.py:113)."""
    streams = list(torch.split(enc_cat, [N_MEL, DIM_F0, DIM_EN, N_MEL], dim=-1))
    widths = [ENC_D, ENC_P, ENC_E, ENC_R]
    for layer in range(3):
        for s in range(4):
            cp = f"{pre}.convolutions_{s + 1}.{layer}"
            h = _conv_cl(P, cp + ".0.conv", streams[s], 2)
            h = F.group_norm(h.transpose(1, 2), widths[s] // GN_CH, P[cp + ".1.weight"],
                             P[cp + ".1.bias"], eps=1e-5).transpose(1, 2)
            streams[s] = F.relu(h)
    cat = mel_calibrate(torch.cat(streams, dim=-1), mel_len, src_len)
    d, f0, e, r = torch.split(cat, widths, dim=-1)
    return (bilstm2(P, pre + ".lstm_1", d), bilstm2(P, pre + ".lstm_2", f0),
            bilstm2(P, pre + ".lstm_3", e), bilstm2(P, pre + ".lstm_4", r))


def aug_classifier(P: Params, pre: str, x: Tensor) -> Tensor:
    """AugmentationClassifier.forward, modules.py:38-45 (GRL is identity in forward;
    RevGrad.backward, modules.py:61-66, negates the gradient -- see grad_reverse)."""
    x = grad_reverse(x)
    h = _linear(P, pre + ".classifier.d_fc1", x)
    h = F.relu(F.layer_norm(h, (HIDDEN,), P[pre + ".classifier.d_bn1.weight"],
                            P[pre + ".classifier.d_bn1.bias"]))
    score = F.log_softmax(_linear(P, pre + ".classifier.d_fc2", h), dim=-1)
    return score.mean(dim=1) if score.dim() > 2 else score


class _GradReverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -g


def grad_reverse(x: Tensor) -> Tensor:
    return _GradReverse.apply(x)


def style_predictor(P: Params, pre: str, x: Tensor, pad: Optional[Tensor], training=False
                    ) -> Tensor:
    """StylePredictor.forward, modules.py:457-465."""
    h = x
    for n in (1, 2):
        h = F.relu(_conv_cl(P, f"{pre}.conv_layer.conv1d_{n}.conv", h, 1))
        h = F.layer_norm(h, (HIDDEN,), P[f"{pre}.conv_layer.layer_norm_{n}.weight"],
                         P[f"{pre}.conv_layer.layer_norm_{n}.bias"])
        h = _dropout(h, 0.5, training)
    out = _linear(P, pre + ".linear_layer", h).squeeze(-1)
    return out.masked_fill(pad, 0.0) if pad is not None else out


def frame_to_phoneme(dur: Tensor, max_len: Optional[int]) -> Tuple[Tensor, Tensor]:
    """Index form of LengthRegulator.LR/expand (modules.py:396-417) + utils.pad
    (utils.py:332-348).  `dur` [B, S] int64 or float; each entry is truncated with int()
    as `vec.expand(int(expand_size), -1)` does.  Returns (index [B, T] int64 with -1 for
    zero-filled frames, mel_len [B] int64 = un-cropped sum of durations).  T = max_len
    if given (longer expansions are cropped by the negative F.pad) else max(mel_len)."""
    d = dur.to(torch.float64).trunc().to(torch.int64).clamp_min(0) if dur.is_floating_point() \
        else dur.to(torch.int64).clamp_min(0)
    csum = torch.cumsum(d, dim=1)
    mel_len = csum[:, -1].clone()
    T = int(max_len) if max_len is not None else int(mel_len.max().item())
    t = torch.arange(T)[None, :].expand(d.shape[0], -1).contiguous()
    idx = torch.searchsorted(csum, t, right=True)
    idx = torch.where(t < mel_len[:, None], idx, torch.full_like(idx, -1))
    return idx, mel_len


def length_regulate(x: Tensor, dur: Tensor, max_len: Optional[int]) -> Tuple[Tensor, Tensor]:
    """LengthRegulator.forward, modules.py:419-423."""
    idx, mel_len = frame_to_phoneme(dur, max_len)
    gathered = torch.gather(x, 1, idx.clamp_min(0)[..., None].expand(-1, -1, x.shape[-1]))
    return gathered * (idx >= 0)[..., None].to(x.dtype), mel_len


def _mlp2(P: Params, pre: str, x: Tensor) -> Tensor:
    """Linear-ReLU-Linear-ReLU stacks, modules.py:250-269."""
    return F.relu(_linear(P, pre + ".2", F.relu(_linear(P, pre + ".0", x))))


def rounded_duration(log_d: Tensor, d_control: float) -> Tensor:
    """modules.py:357-358: clamp(round(exp(log_d) - 1) * d_control, min=0)."""
    return torch.clamp(torch.round(torch.exp(log_d) - LOG_OFFSET) * d_control, min=0)


def style_modeling(P: Params, text, speaker_embed, mel_target, mel_aug, p_norm, e_input,
                   src_len, mel_len, src_pad, mel_pad, d_target=None, p_target=None,
                   e_target=None, max_len=None, d_control=1.0, p_control=1.0, e_control=1.0,
                   training=False):
    """StyleModeling.forward, modules.py:311-387 (+ StyleEncoder.forward, 225-235)."""
    pre = "style_modeling"
    se = pre + ".style_encoder"
    text_enc = text_encoder(P, se + ".text_encoder", text, src_pad, training)
    text_neck = F.relu(_linear(P, se + ".text_linear_down.0", text_enc))
    spk_p = F.relu(_linear(P, se + ".speaker_linear_p.0", speaker_embed))
    spk = F.relu(_linear(P, se + ".speaker_linear.0", speaker_embed))
    enc_cat = encoder_input_cat(mel_target, p_norm, e_input, mel_aug)
    dur_enc, pit_enc, en_enc, noise_enc = audio_encoder(P, se + ".audio_encoder", enc_cat,
                                                        mel_len, src_len)
    S = text_enc.shape[1]

    aug = (aug_classifier(P, pre + ".augmentation_classifier_d", dur_enc),
           aug_classifier(P, pre + ".augmentation_classifier_p", pit_enc),
           aug_classifier(P, pre + ".augmentation_classifier_e", en_enc))

    spk = spk[:, None, :].expand(-1, S, -1)
    spk_p = spk_p[:, None, :].expand(-1, S, -1)
    pit_enc = pit_enc + spk_p

    dur_up = _mlp2(P, pre + ".duration_linear", dur_enc)
    pit_up = _mlp2(P, pre + ".pitch_linear", pit_enc)
    en_up = _mlp2(P, pre + ".energy_linear", en_enc)
    noise_up = _mlp2(P, pre + ".residual_linear", noise_enc)[:, :S]
    neck_up = F.relu(_linear(P, pre + ".text_linear_up.0", text_neck))

    enc = torch.cat((text_enc, neck_up + pit_up, spk, neck_up + en_up, noise_up), dim=-1)

    log_d = style_predictor(P, pre + ".duration_predictor", neck_up + dur_up, src_pad, training)
    if d_target is not None:
        enc, out_mel_len = length_regulate(enc, d_target, max_len)
    else:
        enc, out_mel_len = length_regulate(enc, rounded_duration(log_d, d_control), max_len)
        mel_pad = length_mask(out_mel_len)

    t_e, p_e, s_e, e_e, n_e = torch.split(enc, HIDDEN, dim=-1)

    e_pred = style_predictor(P, pre + ".energy_predictor", e_e, mel_pad, training)
    if e_target is not None:
        e_idx = torch.bucketize(e_target, P[pre + ".energy_bins"])
    else:
        e_pred = e_pred * e_control
        e_idx = torch.bucketize(e_pred, P[pre + ".energy_bins"])
    e_emb = F.embedding(e_idx, P[pre + ".energy_embedding.weight"])

    p_pred = style_predictor(P, pre + ".pitch_predictor", p_e + s_e, mel_pad, training)
    if p_target is not None:
        p_idx = torch.bucketize(p_target, P[pre + ".pitch_bins"])
    else:
        p_pred = p_pred * p_control
        p_idx = torch.bucketize(p_pred, P[pre + ".pitch_bins"])
    p_emb = F.embedding(p_idx, P[pre + ".pitch_embedding.weight"])

    out = t_e + p_emb + s_e + e_emb
    return out, n_e, log_d, p_pred, e_pred, out_mel_len, mel_pad, aug


def predict_inference(P: Params, text_enc, pitch_enc, energy_enc, duration_enc, speaker_enc, noise_enc, src_pad,
                      max_len=None, speaker_normalized=True, d_control=1.0, p_control=1.0, e_control=1.0):
    """StyleModeling.predict_inference, modules.py:285-309 (the synthesize.py inspection / control path): encodings
    [B, S, 256] each -> (text, pitch_embedding, speaker, energy_embedding, noise) [B, T, 256], log_d [B, S],
    pitch / energy predictions [B, T], mel_pad [B, T]."""
    pre = "style_modeling"
    enc = torch.cat((text_enc, pitch_enc, speaker_enc, energy_enc, noise_enc), dim=-1)
    log_d = style_predictor(P, pre + ".duration_predictor", duration_enc, src_pad)
    enc, mel_len = length_regulate(enc, rounded_duration(log_d, d_control), max_len)
    mel_pad = length_mask(mel_len)
    t_e, p_e, s_e, e_e, n_e = torch.split(enc, HIDDEN, dim=-1)
    e_pred = style_predictor(P, pre + ".energy_predictor", e_e, mel_pad) * e_control
    e_emb = F.embedding(torch.bucketize(e_pred, P[pre + ".energy_bins"]), P[pre + ".energy_embedding.weight"])
    p_pred = style_predictor(P, pre + ".pitch_predictor", p_e if speaker_normalized else p_e + s_e, mel_pad) * p_control
    p_emb = F.embedding(torch.bucketize(p_pred, P[pre + ".pitch_bins"]), P[pre + ".pitch_embedding.weight"])
    return t_e, p_emb, s_e, e_emb, n_e, log_d, p_pred, e_pred, mel_pad


def styler_forward(P: Params, src_seq, mel_target, mel_aug, p_norm, e_input, src_len, mel_len,
                   d_target=None, p_target=None, e_target=None, max_src_len=None,
                   max_mel_len=None, speaker_embed=None, d_control=1.0, p_control=1.0,
                   e_control=1.0, training=False, noisy_branch=True):
    """STYLER.forward, styler.py:39-58.  Returns the same 9-tuple.  `noisy_branch=False`
    (BASELINE config 2, "clean branch only") skips the second decode and returns the
    clean outputs in both slots."""
    src_pad = length_mask(src_len, max_src_len)
    mel_pad = length_mask(mel_len, max_mel_len)
    x, noise, log_d, p_pred, e_pred, new_len, new_pad, aug = style_modeling(
        P, src_seq, speaker_embed, mel_target, mel_aug, p_norm, e_input, src_len, mel_len,
        src_pad, mel_pad, d_target, p_target, e_target, max_mel_len, d_control, p_control,
        e_control, training)
    if d_target is None:
        mel_len, mel_pad = new_len, new_pad
    mel, mel_post = decode(P, x, mel_pad, training)
    if noisy_branch:
        mel_n, mel_post_n = decode(P, x.detach() + noise, mel_pad, training)
    else:
        mel_n, mel_post_n = mel, mel_post
    return (mel, mel_n), (mel_post, mel_post_n), log_d, p_pred, e_pred, src_pad, mel_pad, \
        mel_len, aug


# ----------------------------------------------------------------------------------
# losses (loss.py) and the training-step composition (train.py:135-160)
# ----------------------------------------------------------------------------------
def _masked_mean(err: Tensor, valid: Tensor) -> Tensor:
    """masked_select(...) followed by a mean-reduced loss == sum over valid / count."""
    v = valid.to(err.dtype)
    while v.dim() < err.dim():
        v = v[..., None]
    return (err * v).sum() / v.expand_as(err).sum()


def mel_losses(mel, mel_post, mel_target, mel_valid):
    """STYLERLoss.cal_mel_loss, loss.py:16-24 (MSE over valid frames x 80 bins)."""
    return (_masked_mean((mel - mel_target) ** 2, mel_valid),
            _masked_mean((mel_post - mel_target) ** 2, mel_valid))


def nll3(aug, label):
    """3 x nn.NLLLoss summed, loss.py:46-48 / 64-67."""
    return sum(F.nll_loss(a, label) for a in aug)


def styler_loss(log_d_pred, log_d_tgt, p_pred, p_tgt, e_pred, e_tgt, mel, mel_post, mel_tgt,
                src_valid, mel_valid, aug, aug_label):
    """STYLERLoss.forward, loss.py:26-50."""
    mel_l, post_l = mel_losses(mel, mel_post, mel_tgt, mel_valid)
    d_l = _masked_mean((log_d_pred - log_d_tgt).abs(), src_valid)
    p_l = _masked_mean((p_pred - p_tgt).abs(), mel_valid)
    e_l = _masked_mean((e_pred - e_tgt).abs(), mel_valid)
    return mel_l, post_l, d_l, p_l, e_l, nll3(aug, aug_label)


def dat_pass(P: Params, mel_aug, p_norm_aug, e_input_aug, mel_len, src_len):
    """train.py:149-153: the augmented input through encoder_input_cat -> audio encoder ->
    the three classifiers."""
    enc_cat = encoder_input_cat(mel_aug, p_norm_aug, e_input_aug, mel_aug)
    d, p, e, _ = audio_encoder(P, "style_modeling.style_encoder.audio_encoder", enc_cat,
                               mel_len, src_len)
    return (aug_classifier(P, "style_modeling.augmentation_classifier_d", d),
            aug_classifier(P, "style_modeling.augmentation_classifier_p", p),
            aug_classifier(P, "style_modeling.augmentation_classifier_e", e))


def train_losses(P: Params, batch: Dict[str, Tensor], training=True, max_mel_len=None):
    """train.py:135-160: the ten scalars of one step (total first).  `max_mel_len`: the padded mel extent when the batch
    was collated to more than its own maximum (train.py:132 passes np.max(mel_len); a collate that pads further passes its
    own extent -- the tensors must be padded to it)."""
    B = batch["text"].shape[0]
    out = styler_forward(P, batch["text"], batch["mel_target"], batch["mel_aug"],
                         batch["f0_norm"], batch["energy_input"], batch["src_len"],
                         batch["mel_len"], batch["D"], batch["f0"], batch["energy"],
                         int(batch["src_len"].max()),
                         int(batch["mel_len"].max()) if max_mel_len is None else int(max_mel_len),
                         speaker_embed=batch["speaker_embed"], training=training)
    (mel, mel_n), (post, post_n), log_d, p_pred, e_pred, src_pad, mel_pad, _, aug = out
    zeros = torch.zeros(B, dtype=torch.long)
    ones = torch.ones(B, dtype=torch.long)
    mel_l, post_l, d_l, p_l, e_l, cls = styler_loss(
        log_d, batch["log_D"], p_pred, batch["f0"], e_pred, batch["energy"], mel, post,
        batch["mel_target"], ~src_pad, ~mel_pad, aug, zeros)
    mel_nl, post_nl = mel_losses(mel_n, post_n, batch["mel_aug"], ~mel_pad)
    aug_dat = dat_pass(P, batch["mel_aug"], batch["f0_norm_aug"], batch["energy_input_aug"],
                       batch["mel_len"], batch["src_len"])
    cls_dat = nll3(aug_dat, ones)
    total = mel_l + post_l + mel_nl + post_nl + d_l + p_l + e_l + DAT_WEIGHT * (cls + cls_dat)
    return total, mel_l, post_l, mel_nl, post_nl, d_l, p_l, e_l, cls, cls_dat


def noam_lr(step: int, d_model: int = 256, warmup: int = 4000) -> float:
    """optimizer.py:21-32: the step counter is incremented BEFORE this is evaluated."""
    return float(d_model ** -0.5 * min(step ** -0.5, warmup ** -1.5 * step))


# ----------------------------------------------------------------------------------
# STFT -> mel (audio/stft.py, audio/tools.py)
# ----------------------------------------------------------------------------------
def stft_basis(n_fft: int = 1024) -> Tensor:
    """STFT.__init__, stft.py:26-49: rows [Re; Im] of fft(eye(n))[:n/2+1], each multiplied
    by the periodic Hann window (scipy get_window('hann', n, fftbins=True)), fp32."""
    n = np.arange(n_fft)
    four = np.fft.fft(np.eye(n_fft))
    cutoff = n_fft // 2 + 1
    basis = np.vstack([np.real(four[:cutoff]), np.imag(four[:cutoff])])
    basis = torch.from_numpy(basis).float()
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)       # periodic Hann
    return basis * torch.from_numpy(win).float()[None, :]


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=22050, n_fft=1024, n_mels=80, fmin=0.0, fmax=8000.0) -> Tensor:
    """librosa==0.7.2 `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` as called at
    stft.py:128-129 (third-party, absent from /root/reference and from this image):
    Slaney mel scale (htk=False), triangular filters, Slaney area normalisation
    (norm=1: each filter scaled by 2 / (f_hi - f_lo)).  PARITY UNPINNED at this boundary:
    the reference holds no vector for mel_basis.  Cross-checked (not pinned) against a second, independent derivation --
    HF transformers' librosa-compatible `audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")`, fixture
    tests/golden/mel_filterbank_hf.npz: max |diff| = 3.5e-8 of the peak weight -- plus self-checks in
    tests/test_00_oracle_golden.py (peaks monotone, area norm)."""
    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_pts = np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2)
    hz = _mel_to_hz(mel_pts)
    fdiff = np.diff(hz)
    ramps = hz[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, fft_f.size))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[:, None]
    return torch.from_numpy(w).float()


def stft_magnitude(wav: Tensor, n_fft=1024, hop=256) -> Tensor:
    """STFT.transform, stft.py:51-79 (magnitude only): reflect-pad n_fft/2 both sides,
    strided conv against the windowed DFT basis, sqrt(re^2 + im^2).  [B, N] ->
    [B, n_fft/2+1, 1 + N // hop]."""
    x = F.pad(wav[:, None, :], (n_fft // 2, n_fft // 2), mode="reflect")
    spec = F.conv1d(x, stft_basis(n_fft)[:, None, :], stride=hop)
    c = n_fft // 2 + 1
    return torch.sqrt(spec[:, :c] ** 2 + spec[:, c:] ** 2)


def mel_spectrogram(wav: Tensor, mel_basis: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """TacotronSTFT.mel_spectrogram, stft.py:141-160 + dynamic_range_compression,
    audio_processing.py:80-86.  wav in [-1, 1].  Returns (log-mel [B, 80, T],
    energy [B, T])."""
    if float(wav.min()) < -1 or float(wav.max()) > 1:
        raise AssertionError("mel_spectrogram: wav outside [-1, 1]")
    mag = stft_magnitude(wav)
    mb = mel_filterbank() if mel_basis is None else mel_basis
    mel = torch.log(torch.clamp(torch.matmul(mb, mag), min=1e-5))
    return mel, torch.norm(mag, dim=1)


# ----------------------------------------------------------------------------------
# HiFi-GAN generator, mel -> waveform
# ----------------------------------------------------------------------------------
HIFIGAN_V1 = dict(upsample_rates=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4),
                  upsample_initial_channel=512, resblock_kernel_sizes=(3, 7, 11),
                  resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)))    # hifigan/config.json
LRELU_SLOPE = 0.1                                                                # hifigan/models.py:7


def resolve_weight_norm(P: Params) -> Params:
    """`torch.nn.utils.weight_norm` (dim 0) as the checkpoints store it (hifigan/models.py:26-91,115-146;
    utils.py:259-261): weight = weight_g * weight_v / ||weight_v|| with the norm over every dim but 0.
    Keys already folded (`.weight`, after remove_weight_norm) pass through."""
    out = {}
    for k, v in P.items():
        if k.endswith(".weight_g"):
            base = k[:-len("_g")]
            wv = P[base + "_v"]
            norm = wv.reshape(wv.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (wv.dim() - 1))
            out[base] = v * wv / norm
        elif not k.endswith(".weight_v"):
            out[k] = v
    return out


def hifigan_generator(P: Params, mel: Tensor, h: dict = HIFIGAN_V1) -> Tensor:
    """Generator.forward, hifigan/models.py:155-169 with ResBlock.forward (94-101).
    mel [B, 80, T] -> wav [B, 1, T * prod(upsample_rates)]."""
    P = resolve_weight_norm(P)
    nk = len(h["resblock_kernel_sizes"])
    x = F.conv1d(mel, P["conv_pre.weight"], P["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(F.leaky_relu(x, LRELU_SLOPE), P[f"ups.{i}.weight"], P[f"ups.{i}.bias"], stride=u,
                               padding=(k - u) // 2)
        acc = None
        for j, (ks, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            pre = f"resblocks.{i * nk + j}"
            r = x
            for m, d in enumerate(dil):
                t = F.conv1d(F.leaky_relu(r, LRELU_SLOPE), P[f"{pre}.convs1.{m}.weight"], P[f"{pre}.convs1.{m}.bias"],
                             dilation=d, padding=(ks * d - d) // 2)
                t = F.conv1d(F.leaky_relu(t, LRELU_SLOPE), P[f"{pre}.convs2.{m}.weight"], P[f"{pre}.convs2.{m}.bias"],
                             padding=(ks - 1) // 2)
                r = t + r
            acc = r if acc is None else acc + r
        x = acc / nk
    x = F.conv1d(F.leaky_relu(x), P["conv_post.weight"], P["conv_post.bias"], padding=3)    # default slope 0.01 (165)
    return torch.tanh(x)
