"""Host-side mirror of the reference's `transformer/` package (Models.py, Layers.py, SubLayers.py):
same module tree and parameter names/shapes (so `state_dict()` is the reference's), forward passes
built only from libstyler_hip.so kernels."""
import numpy as np
import torch
import torch.nn as nn

from . import hparams as hp
from . import autograd as AG
from . import ops
from .runtime import Derived, Seg, gemm_weight, rt, seg_rows, x3


def get_sinusoid_encoding_table(n_position, d_hid, padding_idx=None):
    """transformer/Models.py:11-30 (float64 numpy, then cast) -- host-side, construction time only."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    tab = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    if padding_idx is not None:
        tab[padding_idx] = 0.0
    return torch.FloatTensor(tab)


class _HipModule(nn.Module):
    """Base of the HIP-backed modules.  The autograd tape (HIP backward kernels) is recorded only in train()
    mode with grad enabled; eval() forwards are inference-only and use the fully fused kernels."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_derived", Derived())

    def _gemm(self, key, x, lin, *, kw=1, act=ops.ACT_NONE, res=None, out=None, lens=None, scale=None,
              shift=None, neg_dx=False, plan=None, out_bf16=False):
        """conv_gemm with the weight of an nn.Linear / nn.Conv1d parameter holder.  Under autograd the call
        goes through ConvGemmFn (HIP backward: dX conv, wgrad, bias column sums)."""
        if (self.training and torch.is_grad_enabled()) and (lin.weight.requires_grad or x.requires_grad):
            assert out is None and scale is None and shift is None and lens is None
            return AG.ConvGemmFn.apply(x, res, lin.weight, lin.bias, self._derived, key, kw, act, neg_dx, plan)
        w, prec = gemm_weight(self._derived, key, lin.weight, x.shape[-1])
        bias = lin.bias if shift is None else shift
        return ops.conv_gemm(x, w, bias, kw=kw, n=lin.weight.shape[0], act=act, prec=prec, scale=scale, res=res,
                             out=out, lens=lens, plan=plan, out_bf16=out_bf16 and prec == ops.PREC_BF16)

    def _ln(self, x, res, ln, lens, out=None, drop_p=0.0):
        """LayerNorm(dropout(x) + res) + pad mask, tape-aware (drop_p only in train mode)."""
        if (self.training and torch.is_grad_enabled()) and (x.requires_grad or ln.weight.requires_grad):
            return AG.LayerNormFn.apply(x, res, ln.weight, ln, lens, drop_p)
        if drop_p > 0 and not rt.disable_dropout:
            return ops.add_layernorm(x, ln.weight, ln.bias, res=res, lens=lens, out=out, in_drop_p=drop_p,
                                     in_drop_seed=AG.next_dropout_seed())
        return ops.add_layernorm(x, ln.weight, ln.bias, res=res, lens=lens, out=out)


class MultiHeadAttention(_HipModule):
    """SubLayers.py:9-61.  Fused QKV projection (one N=768 GEMM), LDS-tiled online-softmax attention,
    output projection with the residual folded into its epilogue, LayerNorm + pad mask in one pass."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super().__init__()
        assert (n_head, d_model, d_k, d_v) == (4, 256, 64, 64), "kernels are specialised for 4 x 64"
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = nn.Linear(d_model, n_head * d_k)
        self.w_ks = nn.Linear(d_model, n_head * d_k)
        self.w_vs = nn.Linear(d_model, n_head * d_v)
        self.layer_norm = nn.LayerNorm(d_model)
        self.fc = nn.Linear(n_head * d_v, d_model)
        self.dropout = nn.Dropout(dropout)

    def _qkv(self):
        d = self._derived
        srcs_w = [self.w_qs.weight, self.w_ks.weight, self.w_vs.weight]
        srcs_b = [self.w_qs.bias, self.w_ks.bias, self.w_vs.bias]
        b = d.get_spec("qkv_b", (768,), False, lambda: [Seg(u, (256,), (1,), (1,), dst_off=k * 256)
                                                        for k, u in enumerate(srcs_b)])
        if rt.prec == ops.PREC_BF16X3:                       # [768, 3 * 256]: rows [w_hi | w_hi | w_lo] against the activation blocks (hi, lo, hi) (runtime.x3)
            w = d.get_spec("qkv_wx3", (768, 768), True, lambda: x3([seg_rows(u, k * 256) for k, u in enumerate(srcs_w)], 256))
            return w, b, ops.PREC_BF16X3
        bf16 = rt.prec == ops.PREC_BF16
        w = d.get_spec("qkv_w16" if bf16 else "qkv_w", (768, 256), bf16,
                       lambda: [seg_rows(u, k * 256) for k, u in enumerate(srcs_w)])
        return w, b, ops.PREC_BF16 if bf16 else ops.PREC_F32

    def forward(self, x, lens, out=None, plan=None, want16=False):
        """x [B, L, 256]; lens int64 [B]; returns LayerNorm(dropout(fc(attn)) + x) with padded rows zeroed
        (the masked_fill of Layers.py:29 is fused into the LayerNorm kernel).  With `plan` (ops.PackPlan) x is the
        packed [1, B*T, 256] tensor and lens = plan.nrows.  `want16` (throughput mode): returns (y, y16) -- y16 the bf16
        copy of y the LayerNorm kernel writes for the FFN's first convolution (None when not produced)."""
        grad = (self.training and torch.is_grad_enabled())
        drop = self.training and self.dropout.p > 0
        asked = want16                                        # the caller unpacks a pair whenever it asked for one
        want16 = want16 and rt.prec == ops.PREC_BF16 and rt.ln_bf16_copy and x.dtype != torch.bfloat16   # (a bf16 stream needs no copy)
        y16 = None
        if grad:                                              # the whole sublayer is one tape node
            y = AG.AttnSublayerFn.apply(x, self.w_qs.weight, self, lens, plan, self.dropout.p, want16)
            if want16:
                y, y16 = y
            return (y, y16) if asked else y
        w, b, prec = self._qkv()
        ctx = ops.attention_fwd(ops.conv_gemm(x, w, b, n=768, prec=prec, plan=plan,
                                              out_bf16=prec == ops.PREC_BF16 and rt.bf16_qkv), lens, plan=plan,
                                out_bf16=prec == ops.PREC_BF16 and rt.bf16_att)
        if drop:                                              # dropout + residual + LayerNorm + mask: one kernel
            y = self._ln(self._gemm("fc", ctx, self.fc, plan=plan), x, self.layer_norm, lens, out,
                         drop_p=self.dropout.p if self.training else 0.0)
            return (y, None) if asked else y
        s16 = x.dtype == torch.bfloat16                       # bf16 residual stream (the packed decoder): y is bf16
        if AG._linear_ln_takes(ctx, prec) and out is None:    # projection + residual + LayerNorm + mask: one launch (csrc/linear_ln.hip)
            wfc, _ = gemm_weight(self._derived, "fc", self.fc.weight, 256)
            if want16 and not s16:
                y16 = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
            y = ops.linear_ln(ctx, wfc, self.fc.bias, x, self.layer_norm.weight, self.layer_norm.bias, lens=lens, out16=y16,
                              packed=plan is not None)
            return (y, y16) if asked else y
        o = self._gemm("fc", ctx, self.fc, res=x, plan=plan)  # eval: residual rides in the GEMM epilogue
        if want16 and not s16:
            y16 = torch.empty_like(o, dtype=torch.bfloat16)
        y = ops.add_layernorm(o, self.layer_norm.weight, self.layer_norm.bias, lens=lens, out=out, out16=y16, out_bf16=s16)
        if rt.sim_bf16_stream and rt.prec == ops.PREC_BF16 and not s16:
            y.copy_(y.to(torch.bfloat16))
        return (y, y16) if asked else y


class PositionwiseFeedForward(_HipModule):
    """SubLayers.py:64-89: Conv1d(256->1024, k=9) + ReLU and Conv1d(1024->256, k=1) as implicit GEMMs."""

    def __init__(self, d_in, d_hid, dropout=0.1):
        super().__init__()
        k = hp.fft_conv1d_kernel_size
        self.w_1 = nn.Conv1d(d_in, d_hid, kernel_size=k[0], padding=(k[0] - 1) // 2)
        self.w_2 = nn.Conv1d(d_hid, d_in, kernel_size=k[1], padding=(k[1] - 1) // 2)
        self.layer_norm = nn.LayerNorm(d_in)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, lens, out=None, plan=None, x16=None):
        """`x16` (optional): the bf16 copy of x (MultiHeadAttention.forward(want16=True)) -- the k = 9 convolution reads it
        instead of x (same values after the operand rounding)."""
        k = hp.fft_conv1d_kernel_size
        if self.training and torch.is_grad_enabled():        # the whole sublayer is one tape node
            return AG.FfnSublayerFn.apply(x, self.w_1.weight, self, lens, plan, self.dropout.p, x16)
        xa = x16 if (x16 is not None and rt.prec == ops.PREC_BF16) else x
        h = self._gemm("w_1", xa, self.w_1, kw=k[0], act=ops.ACT_RELU, plan=plan, out_bf16=True)   # bf16 in throughput mode
        if (self.training and torch.is_grad_enabled()) or (self.training and self.dropout.p > 0):
            return self._ln(self._gemm("w_2", h, self.w_2, kw=k[1], plan=plan), x, self.layer_norm, lens, out,
                            drop_p=self.dropout.p if self.training else 0.0)
        if k[1] == 1 and AG._linear_ln_takes(h, rt.prec) and out is None:
            w2, _ = gemm_weight(self._derived, "w_2", self.w_2.weight, h.shape[-1])
            return ops.linear_ln(h, w2, self.w_2.bias, x, self.layer_norm.weight, self.layer_norm.bias, lens=lens,
                                 packed=plan is not None)
        o = self._gemm("w_2", h, self.w_2, kw=k[1], res=x, plan=plan)
        s16 = x.dtype == torch.bfloat16
        y = ops.add_layernorm(o, self.layer_norm.weight, self.layer_norm.bias, lens=lens, out=out, out_bf16=s16)
        if rt.sim_bf16_stream and rt.prec == ops.PREC_BF16 and not s16:
            y.copy_(y.to(torch.bfloat16))
        return y


class FFTBlock(nn.Module):
    """Layers.py:10-34."""

    def __init__(self, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=dropout)

    def forward(self, x, lens, out=None, plan=None):
        a, a16 = self.slf_attn(x, lens, plan=plan, want16=True)
        return self.pos_ffn(a, lens, out=out, plan=plan, x16=a16)


class _PositionMixin:
    def _pe(self, L, device):
        """Models.py:69-74 / 120-125: stored table for L <= 1001; eval mode regenerates longer ones
        (on device, float64 angles); train mode cannot exceed the stored table."""
        if L <= self.position_enc.shape[1] and not ((not self.training) and L > hp.max_seq_len):
            return self.position_enc[0]
        if self.training:
            raise RuntimeError(f"sequence length {L} exceeds the position table "
                               f"({self.position_enc.shape[1]}) in train mode (Models.py:124-125)")
        cache = self.__dict__.setdefault("_pe_cache", {})
        if cache.get("L") != L or cache["pe"].device != device:
            cache["L"], cache["pe"] = L, ops.sinusoid_table(L, hp.encoder_hidden, device)
        return cache["pe"]


class Encoder(nn.Module, _PositionMixin):
    """Models.py:33-84."""

    def __init__(self, n_src_vocab=hp.n_src_vocab, len_max_seq=hp.max_seq_len, d_word_vec=hp.encoder_hidden,
                 n_layers=hp.encoder_layer, n_head=hp.encoder_head, d_k=hp.encoder_hidden // hp.encoder_head,
                 d_v=hp.encoder_hidden // hp.encoder_head, d_model=hp.encoder_hidden,
                 d_inner=hp.fft_conv1d_filter_size, dropout=hp.encoder_dropout):
        super().__init__()
        self.src_word_emb = nn.Embedding(n_src_vocab, d_word_vec, padding_idx=0)
        self.position_enc = nn.Parameter(
            get_sinusoid_encoding_table(len_max_seq + 1, d_word_vec).unsqueeze(0), requires_grad=False)
        self.layer_stack = nn.ModuleList(
            [FFTBlock(d_model, d_inner, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers)])

    def embed(self, src_seq):
        """Token embedding + positional encoding (Models.py:72-73): the first node of forward()."""
        pe = self._pe(src_seq.shape[1], src_seq.device)
        if (self.training and torch.is_grad_enabled()) and self.src_word_emb.weight.requires_grad:
            return AG.EmbedPosFn.apply(src_seq, self.src_word_emb.weight, self.src_word_emb, pe)
        return ops.embed_pos(src_seq, self.src_word_emb.weight, pe)

    def forward(self, src_seq, lens, out=None, x0=None):
        """x0: embed(src_seq) computed by the caller (StyleEncoder: on the main stream, the FFT blocks on a side stream)."""
        x = self.embed(src_seq) if x0 is None else x0
        for i, layer in enumerate(self.layer_stack):
            x = layer(x, lens, out=out if i == len(self.layer_stack) - 1 else None)
        return x


class Decoder(nn.Module, _PositionMixin):
    """Models.py:87-135."""

    def __init__(self, len_max_seq=hp.max_seq_len, d_word_vec=hp.encoder_hidden, n_layers=hp.decoder_layer,
                 n_head=hp.decoder_head, d_k=hp.decoder_hidden // hp.decoder_head,
                 d_v=hp.decoder_hidden // hp.decoder_head, d_model=hp.decoder_hidden,
                 d_inner=hp.fft_conv1d_filter_size, dropout=hp.decoder_dropout):
        super().__init__()
        self.position_enc = nn.Parameter(
            get_sinusoid_encoding_table(len_max_seq + 1, d_word_vec).unsqueeze(0), requires_grad=False)
        self.layer_stack = nn.ModuleList(
            [FFTBlock(d_model, d_inner, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers)])

    def forward(self, enc_seq, lens):
        pe = self._pe(enc_seq.shape[1], enc_seq.device)
        tape = (self.training and torch.is_grad_enabled()) and enc_seq.requires_grad
        if rt.pack_decoder and enc_seq.shape[0] <= 4096:
            # Every block zeroes its padded rows (Layers.py:29,32) and masks padded keys: the four blocks run on the valid
            # frames only, stored back to back (ops.PackPlan / csrc/pack.hip), and the result is padded again with zeros.
            B, T, _ = enc_seq.shape
            plan = ops.PackPlan(lens, B, T)
            s16 = rt.bf16_stream and rt.prec == ops.PREC_BF16      # throughput mode: the packed residual stream is bf16
            x = AG.PackRowsFn.apply(enc_seq, pe, plan, s16) if tape else ops.pack_rows(enc_seq, plan, add=pe, out_bf16=s16)
            for layer in self.layer_stack:
                x = layer(x, plan.nrows, plan=plan)
            return AG.UnpackRowsFn.apply(x, plan) if tape else ops.unpack_rows(x, plan)
        x = AG.AddPosFn.apply(enc_seq, pe) if tape else ops.add_pos(enc_seq, pe)
        for layer in self.layer_stack:
            x = layer(x, lens)
        return x

    def forward_pair(self, seq_a, seq_b, lens):
        """The clean and the noisy decode (styler.py:52,55) as ONE packed batch of 2B items: the FFT blocks are row- and
        item-wise, so the result is the two separate calls stacked, [2B, T, C] -- with half the launches and twice
        the rows per GEMM (at B=48 the N=256 GEMMs of one decode are 212 tiles on 256 CUs)."""
        B, T, _ = seq_a.shape
        pe = self._pe(T, seq_a.device)
        tape = (self.training and torch.is_grad_enabled()) and (seq_a.requires_grad or seq_b.requires_grad)
        pl = getattr(rt, "pair_lens", None)            # (mel_len, its stacked copy) of the current batch, training.train_losses
        lens2 = pl[1] if (pl is not None and pl[0] is lens) else torch.cat([lens, lens])
        plan = ops.PackPlan(lens2, 2 * B, T)
        s16 = rt.bf16_stream and rt.prec == ops.PREC_BF16
        x = (AG.PackPairFn.apply(seq_a, seq_b, pe, plan, s16) if tape
             else ops.pack_rows_pair(seq_a, seq_b, plan, add=pe, out_bf16=s16))
        for layer in self.layer_stack:
            x = layer(x, plan.nrows, plan=plan)
        return AG.UnpackRowsFn.apply(x, plan) if tape else ops.unpack_rows(x, plan)


class ConvNorm(nn.Module):
    """Layers.py:37-64 (parameter holder: `.conv`)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain="linear"):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)


class PostNet(_HipModule):
    """Layers.py:67-130: five Conv1d(k=5) + BatchNorm1d (+tanh).  Eval: BatchNorm folded into the GEMM
    epilogue (scale/shift), the final residual `+ mel` (styler.py:34) folded into the last one."""

    def __init__(self, n_mel_channels=80, postnet_embedding_dim=512, postnet_kernel_size=5,
                 postnet_n_convolutions=5):
        super().__init__()
        self.convolutions = nn.ModuleList()
        dims = [n_mel_channels] + [postnet_embedding_dim] * (postnet_n_convolutions - 1) + [n_mel_channels]
        for i in range(postnet_n_convolutions):
            self.convolutions.append(nn.Sequential(
                ConvNorm(dims[i], dims[i + 1], kernel_size=postnet_kernel_size, stride=1,
                         padding=int((postnet_kernel_size - 1) / 2), dilation=1),
                nn.BatchNorm1d(dims[i + 1])))
        self.kernel_size = postnet_kernel_size

    def forward(self, x, add_residual=None, segs=1):
        """x [B, T, 80] channels-last -> [B, T, 80]; `add_residual` (the mel) is added to the output.  `segs` > 1: x stacks
        the inputs of `segs` calls along the batch axis (clean + noisy decode, styler.py:52,55); train-mode BatchNorm then
        normalises each of them with its own batch statistics and updates the running statistics once per segment, i.e.
        the result is that of `segs` separate calls, with one GEMM / norm launch per layer instead of `segs`."""
        n = len(self.convolutions)
        for i, seq in enumerate(self.convolutions):
            conv, bn = seq[0].conv, seq[1]
            last = i == n - 1
            act = ops.ACT_NONE if last else ops.ACT_TANH
            if self.training:
                if torch.is_grad_enabled():
                    # conv + BatchNorm (batch statistics) + tanh + F.dropout(.., 0.5, self.training) (Layers.py:126-128) as
                    # one tape node; throughput mode keeps the activations between the convolutions as bf16
                    y = AG.ConvNormFn.apply(x, conv.weight, conv.bias, self._derived, f"c{i}", self.kernel_size, bn, "bn", act,
                                            0.5, segs, not last)
                else:
                    y = self._gemm(f"c{i}", x, conv, kw=self.kernel_size)
                    p = 0.0 if rt.disable_dropout else 0.5
                    y, _, _ = ops.batchnorm_train(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, act,
                                                  drop_p=p, drop_seed=AG.next_dropout_seed() if p > 0 else 0, segs=segs)
                if last and add_residual is not None:
                    y = AG.Add2Fn.apply(y, add_residual) if (self.training and torch.is_grad_enabled()) else ops.add2(y, add_residual)
                x = y
            else:
                scale, shift = self._derived.get(
                    f"bn{i}", [bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.bias],
                    lambda g, b, rm, rv, cb: ops.bn_fold(g, b, rm, rv, cb))
                # throughput mode: the activation between two convolutions is stored as bf16 (the next GEMM rounds its
                # operand to bf16 anyway: same values, half the bytes, and the 256 x 256 LDS-DMA engine can take it)
                x = self._gemm(f"c{i}", x, conv, kw=self.kernel_size, act=act, scale=scale, shift=shift,
                               res=add_residual if last else None, out_bf16=(not last) and rt.bf16_acts)
        if self.training:                                  # BatchNorm1d bookkeeping: one multi-tensor launch per call
            with torch.no_grad():
                torch._foreach_add_([seq[1].num_batches_tracked for seq in self.convolutions], segs)
        return x
