"""Host-side mirror of the reference's `modules.py`: StyleModeling and its blocks, with the reference's
module tree / parameter names, built from libstyler_hip.so kernels only.

Layout notes (all channels-last fp32):
  * the reference's `enc_cat [B, 674, T]` (mel | one-hot f0 | one-hot energy | mel_aug) is never
    built: `encoder_input_cat` returns an `EncoderInput` record and the one-hot streams enter through
    the 5-tap weight-gather kernel (styler_onehot_conv5);
  * the four conv stacks write their last GroupNorm+ReLU straight into one [B, T, 1152] buffer
    (channel slices), which the mel calibrator resamples to [B, S, 1152];
  * the five style encodings are written straight into channel slices of one [B, S, 1280] buffer that
    the LengthRegulator expands to [B, T, 1280]; predictors read its slices in place.
"""
from collections import OrderedDict, namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import autograd as AG
from . import hparams as hp
from . import ops
from .runtime import Seg, rt, seg_rows, x3
from .transformer import ConvNorm, Encoder, _HipModule

EncoderInput = namedtuple("EncoderInput", "mel p_norm e_input mel_aug")


class GradientReversalLayer(nn.Module):
    """modules.py:48-81: identity forward, gradient * (-alpha) backward (applied in the backward ops)."""

    def __init__(self, alpha=1):
        super().__init__()
        self._alpha = torch.tensor(alpha, requires_grad=False)

    def forward(self, x):
        return x


class AugmentationClassifier(_HipModule):
    """modules.py:23-45."""

    def __init__(self, input_dim=hp.encoder_hidden):
        super().__init__()
        self.grl = GradientReversalLayer()
        self.hidden = hp.encoder_hidden
        self.classifier = nn.Sequential(OrderedDict([
            ("d_fc1", nn.Linear(input_dim, self.hidden)),
            ("d_bn1", nn.LayerNorm(self.hidden)),
            ("d_relu1", nn.ReLU()),
            ("d_fc2", nn.Linear(self.hidden, 2)),
            ("d_softmax", nn.LogSoftmax(dim=-1)),
        ]))

    def forward(self, x):
        c = self.classifier
        h = self._gemm("fc1", x, c.d_fc1, neg_dx=True)       # GRL: dX of the first Linear is negated (modules.py:61-66)
        if (self.training and torch.is_grad_enabled()):
            return AG.AugTailFn.apply(h, c.d_fc2.weight, c)
        return ops.aug_classifier_tail(h, c.d_bn1.weight, c.d_bn1.bias, c.d_fc2.weight, c.d_fc2.bias)


class AudioEncoder(_HipModule):
    """modules.py:84-201."""

    def __init__(self):
        super().__init__()
        self.widths = [hp.va_enc_dim_d, hp.va_enc_dim_p, hp.va_enc_dim_e, hp.va_enc_dim_r]
        self.necks = [hp.va_neck_hidden_d, hp.va_neck_hidden_p, hp.va_neck_hidden_e, hp.va_neck_hidden_r]
        in_dims = [hp.n_mel_channels, hp.va_dim_f0, hp.va_dim_energy, hp.n_mel_channels]
        for s in range(4):
            convs = []
            for i in range(3):
                convs.append(nn.Sequential(
                    ConvNorm(in_dims[s] if i == 0 else self.widths[s], self.widths[s], kernel_size=5, stride=1,
                             padding=2, dilation=1, w_init_gain="relu"),
                    nn.GroupNorm(self.widths[s] // hp.va_chs_grp, self.widths[s])))
            setattr(self, f"convolutions_{s + 1}", nn.ModuleList(convs))
            setattr(self, f"lstm_{s + 1}",
                    nn.LSTM(self.widths[s], self.necks[s], 2, batch_first=True, bidirectional=True))

    def _lstm_weights(self, lstm, layer, key, cin):
        """Derived tensors of one BiLSTM layer: fused input weights [8H, in] (fp32 or bf16), summed biases
        [8H] (b_ih + b_hh, forward | reverse), stacked recurrent weights [2, 4H, H]."""
        d = self._derived
        names = [f"weight_ih_l{layer}", f"weight_ih_l{layer}_reverse", f"weight_hh_l{layer}",
                 f"weight_hh_l{layer}_reverse", f"bias_ih_l{layer}", f"bias_hh_l{layer}",
                 f"bias_ih_l{layer}_reverse", f"bias_hh_l{layer}_reverse"]
        wi, wir, wh, whr, bi, bh, bir, bhr = [getattr(lstm, n) for n in names]
        n4, H = wh.shape
        bias = d.get_spec(key + "b", (2 * n4,), False,
                          lambda: [Seg(bi, (n4,), (1,), (1,), src2=bh), Seg(bir, (n4,), (1,), (1,), dst_off=n4, src2=bhr)])
        w_hh = d.get_spec(key + "wh", (2, n4, H), False, lambda: [seg_rows(wh, 0), seg_rows(whr, n4)])
        if rt.prec == ops.PREC_BF16X3 and cin % 8 == 0:
            w = d.get_spec(key + "wix3", (2 * n4, 3 * cin), True, lambda: x3([seg_rows(wi, 0), seg_rows(wir, n4)], cin))
            return w, bias, w_hh, ops.PREC_BF16X3
        bf16 = rt.prec == ops.PREC_BF16 and cin % 8 == 0
        w = d.get_spec(key + ("wi16" if bf16 else "wi"), (2 * n4, cin), bf16,
                       lambda: [seg_rows(wi, 0), seg_rows(wir, n4)])
        return w, bias, w_hh, ops.PREC_BF16 if bf16 else ops.PREC_F32

    def _lstm(self, s, x):
        """2-layer BiLSTM: per layer one MFMA GEMM for both directions' input projections, then the
        persistent recurrent kernel."""
        lstm = getattr(self, f"lstm_{s + 1}")
        H = self.necks[s]
        for layer in range(2):
            key = f"lstm{s}_{layer}"
            if (self.training and torch.is_grad_enabled()):
                x = AG.LstmLayerFn.apply(x, getattr(lstm, f"weight_ih_l{layer}"), lstm, layer, H, self, key)
            else:
                w, bias, w_hh, prec = self._lstm_weights(lstm, layer, key, x.shape[-1])
                x = ops.lstm_bidir(ops.conv_gemm(x, w, bias, n=8 * H, prec=prec), w_hh, H)
        return x

    def forward(self, cat, len_org, seq_len, mask=None, max_seq_len=None, noise_items=None):
        """cat: EncoderInput (from StyleEncoder.encoder_input_cat); len_org = mel_len, seq_len = src_len.
        Returns (duration, f0, energy, noise) encodings [B, S, 2*neck].  S = max(seq_len) as in
        utils.mel_calibrator's re-padding; pass `max_seq_len` to avoid the host sync that needs.
        `noise_items` (round 6, training with the stacked main + DAT batch): only the first `noise_items` items need the
        noise stream -- the DAT pass discards its fourth output (train.py:150 `..., _ = audio_encoder(...)`), so the three
        convolution + GroupNorm stages of stream 4 run on those items only and the other items' columns of the concatenation
        are zeros (their rows of the noise encoding are never read, their gradient is exactly zero)."""
        if not isinstance(cat, EncoderInput):
            raise TypeError("audio_encoder expects the EncoderInput returned by encoder_input_cat "
                            "(the dense one-hot [B, 674, T] tensor is never materialised)")
        mel, p_norm, e_input, mel_aug = cat
        B, T, _ = mel.shape
        dev = mel.device
        W = self.widths
        offs = [0, W[0], W[0] + W[1], W[0] + W[1] + W[2]]
        grad = (self.training and torch.is_grad_enabled())
        catbuf = None if grad else torch.empty(B, T, sum(W), device=dev, dtype=torch.float32)
        err = torch.zeros(1, device=dev, dtype=torch.int32) if rt.strict_inputs else None
        finals, last = [], []
        Bn = B
        if noise_items is not None and grad and rt.fused_cat and rt.skip_dat_noise and 0 < noise_items < B:
            Bn = int(noise_items)
        for s in range(4):
            convs = getattr(self, f"convolutions_{s + 1}")
            x = None
            for i in range(3):
                conv, gn = convs[i][0].conv, convs[i][1]
                if i == 0 and s in (1, 2):
                    v = p_norm if s == 1 else e_input
                    if grad:
                        y = AG.OnehotConv5Fn.apply(v, conv.weight, conv, self._derived, f"oh{s}", err)
                    else:
                        wt = AG.onehot_weight(self._derived, f"oh{s}", conv.weight)
                        y = torch.empty(B, T, W[s], device=dev, dtype=torch.float32)
                        ops.onehot_conv5(v, wt, conv.bias, y, err_flag=err)
                else:
                    src = (mel if s == 0 else mel_aug[:Bn]) if i == 0 else x
                    if grad and i == 2 and rt.fused_cat:    # the four last stages: one node writing into the concatenation
                        last.append((src, conv, gn, f"c{s}_{i}"))
                        continue
                    if grad:                             # conv + GroupNorm + ReLU as one tape node (bf16 between the convs)
                        x = AG.ConvNormFn.apply(src, conv.weight, conv.bias, self._derived, f"c{s}_{i}", 5, gn, "gn",
                                                ops.ACT_RELU, 0.0, 1, i < 2)
                        continue
                    # eval: the conv output is read once, by GroupNorm -- as bf16 in throughput mode (rt.bf16_z) when the item
                    # fits the single-pass kernel (styler_groupnorm_fused_rows)
                    z16 = (rt.bf16_z and rt.bf16_acts and rt.prec == ops.PREC_BF16 and W[s] % 8 == 0
                           and 0 < T <= ops.lib.styler_groupnorm_fused_rows(0))
                    y = self._gemm(f"c{s}_{i}", src, conv, kw=5, out_bf16=z16)
                if grad:
                    # after a one-hot convolution (its backward takes an fp32 gradient): bf16 output only
                    b16 = rt.bf16_acts and rt.prec == ops.PREC_BF16 and W[s] % 8 == 0
                    x = AG.GroupNormReluFn.apply(y, gn.weight, gn, b16 and i < 2, False)
                else:
                    if i == 2:
                        out = catbuf[..., offs[s]:offs[s] + W[s]]
                    elif rt.bf16_acts and rt.prec == ops.PREC_BF16 and W[s] % 8 == 0:
                        out = torch.empty_like(y, dtype=torch.bfloat16)      # consumed only by the next conv (bf16 operand)
                    else:
                        out = y
                    x = ops.groupnorm_relu(y, gn.weight, gn.bias, out=out)
            finals.append(x)
        if grad and len(last) == 4:
            flat = []
            for src, conv, gn, key in last:
                flat += [src, conv.weight, conv.bias]
            catbuf = AG.ConvNormCatFn.apply(self._derived, tuple(k for _, _, _, k in last), tuple(g for _, _, g, _ in last), *flat)
        elif grad:
            catbuf = AG.CatFn.apply(*finals)
        if err is not None and int(err.item()) != 0:
            raise AssertionError("quantize_1D_torch: input outside [0, 1] (utils.py:423)")
        S = int(max_seq_len) if max_seq_len is not None else int(seq_len.max().item())
        cal = AG.MelCalibrateFn.apply(catbuf, len_org, seq_len, S) if grad else ops.mel_calibrate(catbuf, len_org, seq_len, S)
        # the four 2-layer BiLSTMs advance layer by layer together: 4 input GEMMs + ONE recurrent launch per layer
        xs = list(AG.SplitWidthsFn.apply(cal, tuple(W))) if grad else [cal[..., offs[s]:offs[s] + W[s]] for s in range(4)]
        for layer in range(2):
            if grad:
                xs = list(AG.LstmMultiLayerFn.apply(self, layer, self.lstm_1.weight_hh_l0, *xs))
            else:
                gxs, w_hhs = [], []
                for s in range(4):
                    w, bias, w_hh, prec = self._lstm_weights(getattr(self, f"lstm_{s + 1}"), layer, f"lstm{s}_{layer}",
                                                             xs[s].shape[-1])
                    gxs.append(ops.conv_gemm(xs[s], w, bias, n=8 * self.necks[s], prec=prec))
                    w_hhs.append(w_hh)
                xs = ops.lstm_bidir_multi(gxs, w_hhs, self.necks, parts=rt.lstm_parts())
        return tuple(xs)


class StyleEncoder(_HipModule):
    """modules.py:204-235."""

    def __init__(self):
        super().__init__()
        self.text_encoder = Encoder()
        self.audio_encoder = AudioEncoder()
        self.text_linear_down = nn.Sequential(nn.Linear(hp.encoder_hidden, hp.va_neck_hidden_t), nn.ReLU())
        self.speaker_linear_p = nn.Sequential(nn.Linear(hp.speaker_embed_dim, hp.va_neck_hidden_p * 2), nn.ReLU())
        self.speaker_linear = nn.Sequential(nn.Linear(hp.speaker_embed_dim, hp.encoder_hidden), nn.ReLU())
        self.dat_inputs = None        # (mel_aug, f0_norm_aug, energy_input_aug) of the DAT pass, set by train_losses
        self.dat_encodings = None     # its (d, p, e) encodings when the forward ran both passes as one batch
        self.stacked_encodings = None # (rt.pair_classifiers) the [2B, S, C] encodings of both passes, for the classifiers

    def encoder_input_cat(self, mel_target, p_norm, e_input, mel_aug):
        return EncoderInput(mel_target.contiguous(), p_norm.contiguous(), e_input.contiguous(),
                            mel_aug.contiguous())

    def forward(self, text, speaker_embed, mel_target, p_norm, e_input, mel_aug, mel_len, src_len, src_mask,
                text_out=None):
        side = None
        if rt.text_stream and not rt.ar_text_point and self.training and torch.is_grad_enabled():
            # the text encoder (two FFT blocks on [B, S] rows: ~40 launch-latency-bound kernels) on a side stream next to
            # the AudioEncoder's T-domain convolutions; autograd replays each node's backward on its forward stream, so
            # the two backward chains overlap the same way (rt.text_stream)
            # The embedding stays on the MAIN stream: its tape node is the last one of the text encoder's backward chain, autograd
            # runs it on the stream of its forward and orders that stream behind the node's input gradient -- so the side
            # stream's whole backward chain (its parameter gradients are written by the kernels themselves, no AccumulateGrad
            # node of that stream tells the engine to wait for it) is joined into the main stream before backward() returns.
            # (Round 5: with the chain delayed on purpose, the B = 48 eager step folded the embedding's slots before they were
            # written -- the join had been a matter of timing.)
            main = torch.cuda.current_stream()
            if "_text_stream" not in self.__dict__:
                self.__dict__["_text_stream"] = torch.cuda.Stream(device=text.device)
            side = self.__dict__["_text_stream"]
            x0 = self.text_encoder.embed(text)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                text_encoding = self.text_encoder(text, src_len, out=text_out, x0=x0)
                text_encoding_neck = self._gemm("tld", text_encoding, self.text_linear_down[0], act=ops.ACT_RELU)
        else:
            text_encoding = self.text_encoder(text, src_len, out=text_out)
            text_encoding_neck = self._gemm("tld", text_encoding, self.text_linear_down[0], act=ops.ACT_RELU)
        spk_in = speaker_embed.unsqueeze(1)
        speaker_encoding_p = self._gemm("slp", spk_in, self.speaker_linear_p[0], act=ops.ACT_RELU).squeeze(1)
        speaker_encoding = self._gemm("sl", spk_in, self.speaker_linear[0], act=ops.ACT_RELU).squeeze(1)
        dat = self.dat_inputs if (rt.pair_audio and self.training and torch.is_grad_enabled()) else None
        self.dat_inputs, self.dat_encodings, self.stacked_encodings = None, None, None
        if dat is not None:
            # rt.pair_audio: the DAT pass of train.py:149-150 runs the same AudioEncoder on (mel_aug, f0_norm_aug,
            # energy_input_aug, mel_aug); every op in it is per item, so both passes are one batch of 2B items (one BiLSTM
            # chain instead of two; test_experimental_switches_match_default pins it to the two-pass result)
            B = mel_target.shape[0]
            pr = dat[3] if len(dat) > 3 else None
            if pr is not None:
                # the stacked [2B, ...] inputs come with the batch (training.add_pair_inputs / the feeder's collate): no
                # concatenation kernels on the step (six aten cat launches behind idle gaps: 55 us per step)
                enc_cat = self.encoder_input_cat(pr["pair_mel"], pr["pair_f0n"], pr["pair_ein"], pr["pair_mela"])
                len2, src2 = pr["pair_mel_len"], pr["pair_src_len"]
            else:
                enc_cat = self.encoder_input_cat(torch.cat([mel_target, dat[0]]), torch.cat([p_norm, dat[1]]),
                                                 torch.cat([e_input, dat[2]]), torch.cat([mel_aug, dat[0]]))
                len2, src2 = torch.cat([mel_len, mel_len]), torch.cat([src_len, src_len])
            d, p, e, n = self.audio_encoder(enc_cat, len2, src2, mask=None, max_seq_len=text.shape[1], noise_items=B)
            if rt.pair_classifiers:
                # round 4: the augmentation classifiers run ONCE on the stacked [2B, S, C] encodings (StyleModeling.forward;
                # every op of a classifier is per item) -- half the classifier launches, and the DAT halves never need to be
                # cut out: their only consumer is the classifier.  The main halves are views whose gradient comes back
                # zero-extended (SplitBatchFn with an unused second output).
                (da, d), (pa, p), (ea, e) = (AG.StackedFanoutFn.apply(t) for t in (d, p, e))
                self.stacked_encodings = (da, pa, ea)
            else:
                (d, d2), (p, p2), (e, e2) = (AG.SplitBatchFn.apply(t) for t in (d, p, e))
                self.dat_encodings = (d2, p2, e2)
            n, _ = AG.SplitBatchFn.apply(n)            # the DAT pass has no use for the noise stream (train.py:150-153);
            # (a plain n[:B] makes autograd zero-fill the full tensor and copy the half into it: a fill + a memcpy node)
        else:
            enc_cat = self.encoder_input_cat(mel_target, p_norm, e_input, mel_aug)
            d, p, e, n = self.audio_encoder(enc_cat, mel_len, src_len, mask=None, max_seq_len=text.shape[1])
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return text_encoding, text_encoding_neck, speaker_encoding_p, speaker_encoding, d, p, e, n


class LengthRegulator(nn.Module):
    """modules.py:390-423."""

    def forward(self, x, duration, max_len):
        B, S, _ = x.shape
        csum, mel_len, _ = ops.duration_scan(B, S, x.device, dur=duration.contiguous())
        T = int(max_len) if max_len is not None else int(mel_len.max().item())
        return ops.length_regulate(x, csum, T), mel_len


class Conv(nn.Module):
    """modules.py:468-507 (parameter holder: `.conv`)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, bias=True,
                 w_init="linear"):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)


class StylePredictor(_HipModule):
    """modules.py:426-465: [Conv1d(k=3) -> ReLU -> LayerNorm -> dropout] x 2 -> Linear(256,1) -> mask.
    ReLU rides in the GEMM epilogue; the second LayerNorm, the Linear(256,1) and the mask are one kernel."""

    def __init__(self):
        super().__init__()
        k = hp.style_predictor_kernel_size
        f = hp.style_predictor_filter_size
        self.conv_layer = nn.Sequential(OrderedDict([
            ("conv1d_1", Conv(hp.encoder_hidden, f, kernel_size=k, padding=(k - 1) // 2)),
            ("relu_1", nn.ReLU()),
            ("layer_norm_1", nn.LayerNorm(f)),
            ("dropout_1", nn.Dropout(hp.style_predictor_dropout)),
            ("conv1d_2", Conv(f, f, kernel_size=k, padding=1)),
            ("relu_2", nn.ReLU()),
            ("layer_norm_2", nn.LayerNorm(f)),
            ("dropout_2", nn.Dropout(hp.style_predictor_dropout)),
        ]))
        self.linear_layer = nn.Linear(f, 1)

    def forward(self, encoder_output, lens):
        c = self.conv_layer
        k = hp.style_predictor_kernel_size
        p = hp.style_predictor_dropout
        grad = (self.training and torch.is_grad_enabled())
        drop = self.training and p > 0 and not rt.disable_dropout
        if grad and rt.fused_predictor:
            # each stage one tape node: conv (+ ReLU) GEMM and one LayerNorm kernel forward; one LayerNorm-backward kernel
            # (dropout mask + ReLU mask inside), weight gradient and dX GEMM backward
            c1, c2 = c.conv1d_1.conv, c.conv1d_2.conv
            h = AG.PredictorStageFn.apply(encoder_output, c1.weight, c1.bias, self._derived, "c1", k, c.layer_norm_1,
                                          None, None, p if drop else 0.0)
            return AG.PredictorStageFn.apply(h, c2.weight, c2.bias, self._derived, "c2", k, c.layer_norm_2,
                                             self.linear_layer, lens, p if drop else 0.0)
        h = self._gemm("c1", encoder_output, c.conv1d_1.conv, kw=k, act=ops.ACT_RELU)
        h = self._ln(h, None, c.layer_norm_1, None)
        h = AG.dropout(h, p, self.training)
        h = self._gemm("c2", h, c.conv1d_2.conv, kw=k, act=ops.ACT_RELU)
        dp, seed = (p, AG.next_dropout_seed()) if drop else (0.0, 0)
        if grad:
            return AG.LayerNormDotFn.apply(h, self.linear_layer.weight, c.layer_norm_2, self.linear_layer, lens, dp, seed)
        return ops.add_layernorm(h, c.layer_norm_2.weight, c.layer_norm_2.bias, lens=lens,
                                 dot_w=self.linear_layer.weight, dot_b=self.linear_layer.bias, drop_p=dp, drop_seed=seed)


def _mlp(in_dim):
    return nn.Sequential(nn.Linear(in_dim, hp.encoder_hidden), nn.ReLU(),
                         nn.Linear(hp.encoder_hidden, hp.encoder_hidden), nn.ReLU())


class StyleModeling(_HipModule):
    """modules.py:238-387."""

    def __init__(self):
        super().__init__()
        self.style_encoder = StyleEncoder()
        self.augmentation_classifier_d = AugmentationClassifier(input_dim=hp.va_neck_hidden_d * 2)
        self.augmentation_classifier_p = AugmentationClassifier(input_dim=hp.va_neck_hidden_p * 2)
        self.augmentation_classifier_e = AugmentationClassifier(input_dim=hp.va_neck_hidden_e * 2)
        self.duration_linear = _mlp(hp.va_neck_hidden_d * 2)
        self.pitch_norm_linear = _mlp(hp.va_neck_hidden_p * 2)
        self.pitch_linear = _mlp(hp.va_neck_hidden_p * 2)
        self.energy_linear = _mlp(hp.va_neck_hidden_e * 2)
        self.residual_linear = _mlp(hp.va_neck_hidden_r * 2)
        self.text_linear_up = nn.Sequential(nn.Linear(hp.va_neck_hidden_t, hp.encoder_hidden), nn.ReLU())
        self.duration_predictor = StylePredictor()
        self.length_regulator = LengthRegulator()
        self.pitch_predictor = StylePredictor()
        self.energy_predictor = StylePredictor()
        self.pitch_bins = nn.Parameter(torch.exp(torch.linspace(
            np.log(hp.f0_min), np.log(hp.f0_max), hp.n_bins - 1)), requires_grad=False)
        self.energy_bins = nn.Parameter(torch.linspace(hp.energy_min, hp.energy_max, hp.n_bins - 1),
                                        requires_grad=False)
        self.pitch_embedding = nn.Embedding(hp.n_bins, hp.encoder_hidden)
        self.energy_embedding = nn.Embedding(hp.n_bins, hp.encoder_hidden)

    # -- helpers ------------------------------------------------------------------------------
    def _loss_only_stream(self, on):
        """Context for work whose results only feed the LOSS in a teacher-forced training step (the augmentation classifiers, the
        duration / pitch / energy predictors): with rt.pred_stream it runs on ONE side stream next to the main chain (LengthRegulator,
        decoder, PostNet) -- a single fork here, a single join in STYLER.forward -- and autograd replays the backward of these nodes
        on that stream next to the decoder's backward.  The side stream first waits for everything enqueued so far on the current
        stream (its inputs).  `on` False (eval, free-running, switch off): a null context."""
        import contextlib
        # Only inside training.forward_backward (it owns the joins: the forward one in STYLER.forward, the backward one in every
        # WgradArena.flush, and it resets ops.loss_side_stream).  A caller that runs StyleModeling / loss.backward() on its own
        # gets everything on its current stream (round-5 advisor).
        if not on or ops.zero_slab is None:
            return contextlib.nullcontext()
        main = torch.cuda.current_stream()
        if "_pred_stream" not in self.__dict__:       # (created once: setdefault() would build and drop a stream per forward)
            self.__dict__["_pred_stream"] = torch.cuda.Stream(device=main.device)
        side = self.__dict__["_pred_stream"]
        side.wait_stream(main)
        self._pred_side = side
        ops.loss_side_stream = side                   # (every WgradArena.flush joins it: partial tiles are written there too)
        return torch.cuda.stream(side)

    def _mlp2(self, key, seq, x, res=None, out=None):
        h = self._gemm(key + "0", x, seq[0], act=ops.ACT_RELU)
        return self._gemm(key + "2", h, seq[2], act=ops.ACT_RELU, res=res, out=out)

    def _mlp2_multi(self, specs):
        """[(key, Sequential(Linear, ReLU, Linear, ReLU), x), ...] -> their outputs: the first Linears of all MLPs as one tape
        node / grouped launch, then the second ones (autograd.ConvGemmMultiFn; the MLPs are independent of one another)."""
        for layer in (0, 2):
            flat = []
            for (key, seq, x) in specs:
                flat += [x, seq[layer].weight, seq[layer].bias]
            metas = tuple((self._derived, key + str(layer), ops.ACT_RELU, False) for key, _, _ in specs)
            ys = AG.ConvGemmMultiFn.apply(metas, *flat)
            specs = [(key, seq, y) for (key, seq, _), y in zip(specs, ys)]
        return [y for _, _, y in specs]

    def _expand_and_predict(self, encodings, src_len, duration_target, log_d, max_len, mel_len_in, mel_mask,
                            pitch_target, energy_target, d_control, p_control, e_control, pitch_plus_speaker=True,
                            want_noise_sum=True, ids=None):
        """LengthRegulator + energy/pitch predictors + bucketise/embed/add (modules.py:352-385)."""
        B, S, _ = encodings.shape
        H = hp.encoder_hidden
        if duration_target is not None:
            csum, mel_len, _ = ops.duration_scan(B, S, encodings.device, dur=duration_target.contiguous())
            T = int(max_len) if max_len is not None else int(mel_len.max().item())
            lens = mel_len_in
            out_len, out_mask = mel_len_in, mel_mask
        else:
            csum, mel_len, _ = ops.duration_scan(B, S, encodings.device, log_d=log_d, d_control=d_control)
            T = int(max_len) if max_len is not None else int(mel_len.max().item())
            if T <= 0:
                raise ValueError("free-running synthesis: every predicted duration is zero (empty mel)")
            lens = mel_len
            out_len, out_mask = mel_len, ops.length_mask(mel_len, T)
        grad = (self.training and torch.is_grad_enabled()) and encodings.requires_grad
        lr = AG.LengthRegulateFn.apply(encodings, csum, T) if grad else ops.length_regulate(encodings, csum, T)
        if grad and rt.fused_split:
            t_e, p_e, s_e, e_e, n_e = AG.SplitChannelsFn.apply(lr, H)
        else:
            t_e, p_e, s_e, e_e, n_e = (lr[..., i * H:(i + 1) * H] for i in range(5))       # [B, T, 1280] slices

        # (teacher-forced training: the two predictions only feed the loss -- rt.pred_stream; joined by STYLER.forward)
        with self._loss_only_stream(grad and rt.pred_stream and energy_target is not None and pitch_target is not None):
            energy_prediction = self.energy_predictor(e_e, lens)
            if pitch_plus_speaker:
                p_in = AG.Add2Fn.apply(p_e, s_e) if grad else ops.add2(p_e, s_e)
            else:
                p_in = p_e
            pitch_prediction = self.pitch_predictor(p_in, lens)
        if energy_target is not None:
            e_src, e_scale = energy_target.contiguous(), 1.0
        else:
            e_src, e_scale = energy_prediction, e_control
        if pitch_target is not None:
            p_src, p_scale = pitch_target.contiguous(), 1.0
        else:
            p_src, p_scale = pitch_prediction, p_control
        if grad:
            out, out_noisy = AG.BucketEmbedAddFn.apply(t_e, s_e, n_e if want_noise_sum else None,
                                                       self.pitch_embedding.weight, p_src.detach(), p_scale,
                                                       e_src.detach(), e_scale, self)
        else:
            out, out_noisy = ops.bucket_embed_add(
                t_e, s_e, p_src, p_scale, e_src, e_scale, self.pitch_bins, self.energy_bins,
                self.pitch_embedding.weight, self.energy_embedding.weight, noise=n_e if want_noise_sum else None,
                p_ids=ids[0] if ids else None, e_ids=ids[1] if ids else None)
        if energy_target is None and e_control != 1.0:
            energy_prediction = energy_prediction * e_control
        if pitch_target is None and p_control != 1.0:
            pitch_prediction = pitch_prediction * p_control
        return out, out_noisy, n_e, pitch_prediction, energy_prediction, out_len, out_mask, lr

    # -- reference entry points ----------------------------------------------------------------
    def forward(self, text, speaker_embed, mel_target, mel_aug, p_norm, e_input, src_len, mel_len, src_mask,
                mel_mask=None, duration_target=None, pitch_target=None, energy_target=None, max_len=None,
                d_control=1.0, p_control=1.0, e_control=1.0):
        B, S = text.shape
        H = hp.encoder_hidden
        grad = (self.training and torch.is_grad_enabled())
        # eval: the five encodings are written straight into channel slices of one [B, S, 1280] buffer;
        # under autograd they are separate tape tensors joined by CatFn (backward = slice views)
        encodings = None if grad else torch.empty(B, S, 5 * H, device=text.device, dtype=torch.float32)
        sl = [None] * 5 if grad else [encodings[..., i * H:(i + 1) * H] for i in range(5)]

        (text_encoding, text_encoding_neck, speaker_encoding_p, speaker_encoding, duration_encoding,
         pitch_encoding, energy_encoding, noise_encoding) = self.style_encoder(
            text, speaker_embed, mel_target, p_norm, e_input, mel_aug, mel_len, src_len, src_mask, text_out=sl[0])
        max_seq_len = S

        se = self.style_encoder
        self.dat_posteriors = None
        self._pred_side = None
        side_ok = grad and rt.pred_stream and duration_target is not None
        with self._loss_only_stream(side_ok and rt.pred_stream_cls):
            if se.stacked_encodings is not None:
                # main + DAT pass of the classifiers (train.py:135-136 and 149-153) as one batch of 2B items; the rows of the two
                # passes are cut apart on the [2B, 2] log-probabilities
                (d_all, p_all, e_all), se.stacked_encodings = se.stacked_encodings, None
                cls = (self.augmentation_classifier_d, self.augmentation_classifier_p, self.augmentation_classifier_e)
                if rt.grouped_mlps:                      # the three first Linears (GRL: negated dX) as one grouped launch
                    flat = []
                    for c, t in zip(cls, (d_all, p_all, e_all)):
                        flat += [t, c.classifier.d_fc1.weight, c.classifier.d_fc1.bias]
                    hs = AG.ConvGemmMultiFn.apply(tuple((c._derived, "fc1", ops.ACT_NONE, True) for c in cls), *flat)
                    post = [AG.SplitBatchFn.apply(AG.AugTailFn.apply(h, c.classifier.d_fc2.weight, c.classifier))
                            for c, h in zip(cls, hs)]
                else:
                    post = [AG.SplitBatchFn.apply(c(t)) for c, t in zip(cls, (d_all, p_all, e_all))]
                aug_posterior_d, aug_posterior_p, aug_posterior_e = (pp[0] for pp in post)
                self.dat_posteriors = tuple(pp[1] for pp in post)
            else:
                aug_posterior_d = self.augmentation_classifier_d(duration_encoding)
                aug_posterior_p = self.augmentation_classifier_p(pitch_encoding)
                aug_posterior_e = self.augmentation_classifier_e(energy_encoding)

        # for the inspection (modules.py:327-333)
        self.max_seq_len = max_seq_len
        self.pitch_encoding = pitch_encoding
        self.speaker_encoding = speaker_encoding.unsqueeze(1).expand(-1, S, -1)
        self.speaker_encoding_p = speaker_encoding_p.unsqueeze(1).expand(-1, S, -1)

        if grad:
            pitch_in = AG.AddRowvecFn.apply(pitch_encoding, speaker_encoding_p, S)
            text_neck_up = self._gemm("tlu", text_encoding_neck, self.text_linear_up[0], act=ops.ACT_RELU)
            fused_cat = rt.grouped_mlps and rt.style_cat
            if rt.grouped_mlps:
                # round 4: the four style MLPs advance layer by layer together (one grouped launch per layer and direction)
                duration_up, pitch_up, energy_up, residual_up = self._mlp2_multi([
                    ("dl", self.duration_linear, duration_encoding), ("pl", self.pitch_linear, pitch_in),
                    ("el", self.energy_linear, energy_encoding), ("rl", self.residual_linear, noise_encoding)])
                if not fused_cat:
                    sl[1] = AG.Add2Fn.apply(pitch_up, text_neck_up)
                    sl[4] = residual_up
            else:
                duration_up = self._mlp2("dl", self.duration_linear, duration_encoding)
                sl[1] = self._mlp2("pl", self.pitch_linear, pitch_in, res=text_neck_up)
                energy_up = self._mlp2("el", self.energy_linear, energy_encoding)
                sl[4] = self._mlp2("rl", self.residual_linear, noise_encoding)
            if fused_cat:                               # round 6: the five slices and the predictor's input in one launch
                encodings, dp_in = AG.StyleCatFn.apply(text_encoding, pitch_up, text_neck_up, speaker_encoding, energy_up,
                                                       residual_up, duration_up)
                sl[0], sl[4] = text_encoding, residual_up
            else:
                dp_in = AG.Add2Fn.apply(text_neck_up, duration_up)
                sl[0] = text_encoding
                sl[2] = AG.AddRowvecFn.apply(None, speaker_encoding, S)
                sl[3] = AG.Add2Fn.apply(text_neck_up, energy_up)
                encodings = AG.CatFn.apply(*sl)
        else:
            pitch_in = ops.add_rowvec(pitch_encoding, speaker_encoding_p, S)
            text_neck_up = self._gemm("tlu", text_encoding_neck, self.text_linear_up[0], act=ops.ACT_RELU)
            duration_up = self._mlp2("dl", self.duration_linear, duration_encoding)
            dp_in = ops.add2(text_neck_up, duration_up)
            self._mlp2("pl", self.pitch_linear, pitch_in, res=text_neck_up, out=sl[1])
            ops.add_rowvec(None, speaker_encoding, S, out=sl[2])
            energy_up = self._mlp2("el", self.energy_linear, energy_encoding)
            ops.add2(text_neck_up, energy_up, out=sl[3])
            self._mlp2("rl", self.residual_linear, noise_encoding, out=sl[4])

        # for the inspection (modules.py:341-348)
        self.text_encoding_neck = text_neck_up
        self.duration_encoding = duration_up
        self.energy_encoding = energy_up
        self.noise_encoding = sl[4]
        self.text_encoding = sl[0]
        self.src_mask = src_mask
        self.max_len = max_len

        with self._loss_only_stream(side_ok and duration_target is not None):
            log_duration_prediction = self.duration_predictor(dp_in, src_len)
        out, out_noisy, n_e, pitch_prediction, energy_prediction, out_len, out_mask, _ = self._expand_and_predict(
            encodings, src_len, duration_target, log_duration_prediction, max_len, mel_len, mel_mask, pitch_target,
            energy_target, d_control, p_control, e_control)
        self._out_noisy = out_noisy
        return (out, n_e, log_duration_prediction, pitch_prediction, energy_prediction, out_len, out_mask,
                (aug_posterior_d, aug_posterior_p, aug_posterior_e))

    def predict_inference(self, text_encoding, pitch_encoding, energy_encoding, duration_encoding,
                          speaker_encoding, noise_encoding, src_mask, max_len, speaker_normalized=True,
                          d_control=1.0, p_control=1.0, e_control=1.0):
        """modules.py:285-309 (the synthesize.py inspection path)."""
        B, S, H = text_encoding.shape
        encodings = torch.empty(B, S, 5 * H, device=text_encoding.device, dtype=torch.float32)
        for i, part in enumerate((text_encoding, pitch_encoding, speaker_encoding, energy_encoding,
                                  noise_encoding)):
            ops.add2(part.contiguous() if part.stride(-1) != 1 or part.stride(0) != S * part.stride(1)
                     else part, None, out=encodings[..., i * H:(i + 1) * H])
        src_len = (~src_mask).sum(dim=1).to(torch.int64)
        log_d = self.duration_predictor(duration_encoding.contiguous(), src_len)
        csum, mel_len, _ = ops.duration_scan(B, S, encodings.device, log_d=log_d, d_control=d_control)
        T = int(max_len) if max_len is not None else int(mel_len.max().item())
        ids = (torch.empty(B, T, device=encodings.device, dtype=torch.int32),
               torch.empty(B, T, device=encodings.device, dtype=torch.int32))
        out, _, n_e, pitch_prediction, energy_prediction, _, mel_mask, lr = self._expand_and_predict(
            encodings, src_len, None, log_d, T, None, None, None, None, d_control, p_control, e_control,
            pitch_plus_speaker=not speaker_normalized, want_noise_sum=False, ids=ids)
        t_e, _, s_e, _, _ = (lr[..., i * H:(i + 1) * H] for i in range(5))
        # inspection-only entry: the two embedding tables are returned un-summed, as the reference does
        pitch_embedding = ops.gather_rows(ids[0], self.pitch_embedding.weight)
        energy_embedding = ops.gather_rows(ids[1], self.energy_embedding.weight)
        return (t_e, pitch_embedding, s_e, energy_embedding, n_e, log_d, pitch_prediction, energy_prediction,
                mel_mask)
