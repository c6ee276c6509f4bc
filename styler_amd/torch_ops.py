"""The fused ops of the path as `torch.library` custom operators (namespace `styler`), SURVEY 8(b): schema'd,
dispatcher-visible entry points over libstyler_hip.so with registered autograd formulas and fake (meta) kernels, so that
the kernels compose with stock PyTorch code (`torch.ops.styler.conv_gemm(...)`, `torch.library.opcheck`, tracing):

    torch.ops.styler.conv_gemm(x, weight, bias, act, prec)               Linear / Conv1d('same') + activation
    torch.ops.styler.attention(qkv, lens, prec)                          4 x 64 masked self-attention (+ lse for backward)
    torch.ops.styler.add_layernorm(x, res, gamma, beta, lens)            LayerNorm(x + res) with pad mask
    torch.ops.styler.length_regulate(x, durations, max_len)              LengthRegulator expand (+ mel_len)
    torch.ops.styler.stft_mel(wav, basis, mel_basis, prec)               STFT -> log-mel / energy

Round 6: every operator above has its backward as a dispatcher-visible operator of its own (`conv_gemm_bwd`, `attention_bwd`,
`add_layernorm_bwd`, `length_regulate_bwd`: the registered autograd formulas call THOSE, so a traced backward consists of `styler::`
nodes too), and the other fused kernels of the path are registered the same way, forward + backward:

    groupnorm_relu / _bwd        GroupNorm(16 ch per group over padded T) + ReLU      modules.py:103-161
    batchnorm_act / _bwd         train-mode BatchNorm1d + tanh + dropout (PostNet)    transformer/Layers.py:121-130
    lstm_bidir / _bwd            one BiLSTM layer's recurrence (gate pre-activations in)   modules.py:100-101,179-182
    linear_ln                    Linear + residual + LayerNorm + pad mask             transformer/SubLayers.py:55-61,86-89
    masked_err_mean / _bwd       masked MSE / L1 mean                                  loss.py:16-50
    nll3 / _bwd                  three NLLLoss(mean) terms, summed                     loss.py:46-48,60-68
    dropout                      counter-based dropout (its own backward)
    embed_pos / _bwd             token embedding + sinusoid positions                  transformer/Models.py:60-84
    mel_calibrate / _bwd         frame -> phoneme pooling of the AudioEncoder          modules.py:203-230
    aug_classifier_tail / _bwd   LayerNorm -> ReLU -> Linear(256, 2) -> log-softmax -> time mean   modules.py:30-45
    clip_adam_step               clip_grad_norm_ + Adam over flat buffers (in place)   train.py:181-185

Weights are the reference's parameter layouts ([n, cin] or [n, cin, kw]); the kernel layouts are derived inside the op.
`styler_amd`'s own modules do NOT route through the dispatcher: a Python-registered operator costs tens of microseconds of
host time per call, and the train step issues ~700 launches (they call the same C entry points directly, with the derived
layouts cached per optimiser step -- runtime.Derived).  The two routes are the same kernels; tests/test_10_hip_parity.py
checks them against each other."""
import torch
from torch.library import Library, impl, register_autograd, register_fake

from . import ops

_LIB = Library("styler", "DEF")


def _kernel_weight(weight, prec):
    """[n, cin] / [n, cin, kw] (reference layout) -> [n, kw * cin] kernel layout, bf16 in throughput mode."""
    w = weight if weight.dim() == 2 else weight.permute(0, 2, 1).reshape(weight.shape[0], -1)
    w = w.contiguous()
    return ops.cast_bf16(w) if prec == ops.PREC_BF16 else w


def _kernel_weight_bwd(weight, prec):
    """dX weight: [cin, kw * n] with the taps flipped."""
    if weight.dim() == 2:
        w = weight.t().contiguous()
    else:
        n, cin, kw = weight.shape
        w = weight.flip(2).permute(1, 2, 0).reshape(cin, kw * n).contiguous()
    return ops.cast_bf16(w) if prec == ops.PREC_BF16 else w


# ---- conv_gemm --------------------------------------------------------------------------------------------------------------
_LIB.define("conv_gemm(Tensor x, Tensor weight, Tensor? bias, int act, int prec) -> Tensor")


@impl(_LIB, "conv_gemm", "CUDA")
def _conv_gemm(x, weight, bias, act, prec):
    kw = weight.shape[2] if weight.dim() == 3 else 1
    return ops.conv_gemm(x.contiguous(), _kernel_weight(weight, prec), bias, kw=kw, n=weight.shape[0], act=act, prec=prec)


@register_fake("styler::conv_gemm")
def _conv_gemm_fake(x, weight, bias, act, prec):
    return x.new_empty(x.shape[0], x.shape[1], weight.shape[0])


def _conv_gemm_setup(ctx, inputs, output):
    x, weight, bias, act, prec = inputs
    ctx.save_for_backward(x, weight, output)
    ctx.act, ctx.prec, ctx.has_bias = act, prec, bias is not None


_LIB.define("conv_gemm_bwd(Tensor dy, Tensor x, Tensor weight, Tensor y, int act, int prec) -> (Tensor, Tensor, Tensor)")


@impl(_LIB, "conv_gemm_bwd", "CUDA")
def _conv_gemm_bwd(dy, x, weight, y, act, prec):
    """-> (dx, dw, db): the activation's backward, the weight gradient of all taps + the bias column sums (one launch), and the dX
    convolution on the forward engine with the transposed, tap-flipped weight."""
    kw = weight.shape[2] if weight.dim() == 3 else 1
    n, cin = weight.shape[0], weight.shape[1]
    dz = ops.act_bwd(dy.contiguous(), y, act) if act != ops.ACT_NONE else dy.contiguous()
    dw = torch.zeros_like(weight)
    db = torch.zeros(n, device=x.device, dtype=torch.float32)
    ops.wgrad(dz, x.contiguous(), dw, n, cin, kw=kw, db=db, prec=prec)
    dx = ops.conv_gemm(dz, _kernel_weight_bwd(weight, prec), None, kw=kw, n=cin, prec=prec)
    return dx, dw, db


@register_fake("styler::conv_gemm_bwd")
def _conv_gemm_bwd_fake(dy, x, weight, y, act, prec):
    return torch.empty_like(x), torch.empty_like(weight), weight.new_empty(weight.shape[0])


def _conv_gemm_backward(ctx, dy):
    x, weight, y = ctx.saved_tensors
    dx, dw, db = torch.ops.styler.conv_gemm_bwd(dy, x, weight, y, ctx.act, ctx.prec)
    return dx, dw, (db if ctx.has_bias else None), None, None


register_autograd("styler::conv_gemm", _conv_gemm_backward, setup_context=_conv_gemm_setup)


# ---- attention ---------------------------------------------------------------------------------------------------------------
_LIB.define("attention(Tensor qkv, Tensor lens, int prec) -> (Tensor, Tensor)")


@impl(_LIB, "attention", "CUDA")
def _attention(qkv, lens, prec):
    B, L, _ = qkv.shape
    lse = torch.empty(B, 4, L, device=qkv.device, dtype=torch.float32)
    out = ops.attention_fwd(qkv.contiguous(), lens.contiguous(), lse=lse, prec=prec)
    return out, lse


@register_fake("styler::attention")
def _attention_fake(qkv, lens, prec):
    B, L, _ = qkv.shape
    return qkv.new_empty(B, L, 256), qkv.new_empty(B, 4, L)


def _attention_setup(ctx, inputs, output):
    qkv, lens, prec = inputs
    ctx.save_for_backward(qkv, lens, output[0], output[1])
    ctx.prec = prec


_LIB.define("attention_bwd(Tensor qkv, Tensor out, Tensor dout, Tensor lse, Tensor lens, int prec) -> Tensor")


@impl(_LIB, "attention_bwd", "CUDA")
def _attention_bwd(qkv, out, dout, lse, lens, prec):
    return ops.attention_bwd(qkv.contiguous(), out, dout.contiguous(), lse, lens.contiguous(), prec=prec)


@register_fake("styler::attention_bwd")
def _attention_bwd_fake(qkv, out, dout, lse, lens, prec):
    return torch.empty_like(qkv)


def _attention_backward(ctx, dout, dlse):
    qkv, lens, out, lse = ctx.saved_tensors
    return torch.ops.styler.attention_bwd(qkv, out, dout, lse, lens, ctx.prec), None, None


register_autograd("styler::attention", _attention_backward, setup_context=_attention_setup)


# ---- add_layernorm -------------------------------------------------------------------------------------------------------------
_LIB.define("add_layernorm(Tensor x, Tensor? res, Tensor gamma, Tensor beta, Tensor? lens) -> (Tensor, Tensor)")


@impl(_LIB, "add_layernorm", "CUDA")
def _add_layernorm(x, res, gamma, beta, lens):
    s = torch.empty_like(x) if res is not None else x
    y = ops.add_layernorm(x.contiguous(), gamma, beta, res=res, lens=lens, sum_out=s if res is not None else None)
    return y, s


@register_fake("styler::add_layernorm")
def _add_layernorm_fake(x, res, gamma, beta, lens):
    return torch.empty_like(x), torch.empty_like(x)


def _add_layernorm_setup(ctx, inputs, output):
    x, res, gamma, beta, lens = inputs
    ctx.save_for_backward(output[1], gamma, beta, *([lens] if lens is not None else []))
    ctx.has_res, ctx.has_lens = res is not None, lens is not None


_LIB.define("add_layernorm_bwd(Tensor s, Tensor dy, Tensor gamma, Tensor beta, Tensor? lens) -> (Tensor, Tensor, Tensor)")


@impl(_LIB, "add_layernorm_bwd", "CUDA")
def _add_layernorm_bwd(s, dy, gamma, beta, lens):
    """s = the pre-norm sum the forward saved -> (dx (= dres), dgamma, dbeta); statistics are recomputed from s."""
    dg, db = torch.zeros_like(gamma), torch.zeros_like(beta)
    dx = ops.layernorm_bwd(s.contiguous(), dy.contiguous(), gamma, beta, dg, db, lens=lens)
    return dx, dg, db


@register_fake("styler::add_layernorm_bwd")
def _add_layernorm_bwd_fake(s, dy, gamma, beta, lens):
    return torch.empty_like(s), torch.empty_like(gamma), torch.empty_like(beta)


def _add_layernorm_backward(ctx, dy, ds):
    saved = ctx.saved_tensors
    s, gamma, beta = saved[:3]
    lens = saved[3] if ctx.has_lens else None
    dx, dg, db = torch.ops.styler.add_layernorm_bwd(s, dy, gamma, beta, lens)
    return dx, (dx if ctx.has_res else None), dg, db, None


register_autograd("styler::add_layernorm", _add_layernorm_backward, setup_context=_add_layernorm_setup)


# ---- length_regulate -------------------------------------------------------------------------------------------------------------
_LIB.define("length_regulate(Tensor x, Tensor durations, int max_len) -> (Tensor, Tensor)")


@impl(_LIB, "length_regulate", "CUDA")
def _length_regulate(x, durations, max_len):
    B, S, _ = x.shape
    csum, mel_len, _ = ops.duration_scan(B, S, x.device, dur=durations.contiguous())
    return ops.length_regulate(x.contiguous(), csum, max_len), mel_len


@register_fake("styler::length_regulate")
def _length_regulate_fake(x, durations, max_len):
    return x.new_empty(x.shape[0], max_len, x.shape[2]), durations.new_empty(x.shape[0], dtype=torch.int64)


def _length_regulate_setup(ctx, inputs, output):
    x, durations, max_len = inputs
    ctx.save_for_backward(durations)
    ctx.S = x.shape[1]


_LIB.define("length_regulate_bwd(Tensor dy, Tensor durations, int S) -> Tensor")


@impl(_LIB, "length_regulate_bwd", "CUDA")
def _length_regulate_bwd(dy, durations, S):
    B, S_ = durations.shape
    csum, _, _ = ops.duration_scan(B, S_, dy.device, dur=durations.contiguous())
    return ops.length_regulate_bwd(dy.contiguous(), csum, S)


@register_fake("styler::length_regulate_bwd")
def _length_regulate_bwd_fake(dy, durations, S):
    return dy.new_empty(dy.shape[0], S, dy.shape[2])


def _length_regulate_backward(ctx, dy, dlen):
    (durations,) = ctx.saved_tensors
    return torch.ops.styler.length_regulate_bwd(dy, durations, ctx.S), None, None


register_autograd("styler::length_regulate", _length_regulate_backward, setup_context=_length_regulate_setup)


# ---- stft_mel (inference front end, no gradient) ----------------------------------------------------------------------------------
_LIB.define("stft_mel(Tensor wav, Tensor? wav_len) -> (Tensor, Tensor, Tensor, Tensor)")
_STFT = {}


@impl(_LIB, "stft_mel", "CUDA")
def _stft_mel(wav, wav_len):
    from .audio import TacotronSTFT
    key = str(wav.device)
    if key not in _STFT:
        _STFT[key] = TacotronSTFT().to(wav.device)
    f = _STFT[key].features(wav, wav_len)
    return f["mel"], f["energy"], f["e_input"], f["mel_len"]


@register_fake("styler::stft_mel")
def _stft_mel_fake(wav, wav_len):
    B, F = wav.shape[0], 1 + wav.shape[1] // 256
    return (wav.new_empty(B, F, 80), wav.new_empty(B, F), wav.new_empty(B, F), wav.new_empty(B, dtype=torch.int64))





# =====================================================================================================================================
# Round 6: the other fused kernels of the path, forward + backward operators each
# =====================================================================================================================================
def _pair(name, schema, impl_fn, fake_fn):
    _LIB.define(f"{name}{schema}")
    impl(_LIB, name, "CUDA")(impl_fn)
    register_fake(f"styler::{name}")(fake_fn)


# ---- groupnorm_relu ------------------------------------------------------------------------------------------------------------------
def _gn_fwd(x, gamma, beta):
    B, L, C = x.shape
    stats = torch.empty(B, C // 16, 2, device=x.device, dtype=torch.float32)
    y = ops.groupnorm_relu(x.contiguous(), gamma, beta, out=torch.empty_like(x), stats=stats)
    return y, stats


def _gn_bwd(x, dy, gamma, beta, stats):
    dg, db = torch.zeros_like(gamma), torch.zeros_like(beta)
    dx = ops.groupnorm_relu_bwd(x.contiguous(), dy.contiguous(), gamma, beta, stats, dg, db)
    return dx, dg, db


_pair("groupnorm_relu", "(Tensor x, Tensor gamma, Tensor beta) -> (Tensor, Tensor)", _gn_fwd,
      lambda x, gamma, beta: (torch.empty_like(x), x.new_empty(x.shape[0], x.shape[2] // 16, 2)))
_pair("groupnorm_relu_bwd", "(Tensor x, Tensor dy, Tensor gamma, Tensor beta, Tensor stats) -> (Tensor, Tensor, Tensor)", _gn_bwd,
      lambda x, dy, gamma, beta, stats: (torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(beta)))


def _gn_setup(ctx, inputs, output):
    x, gamma, beta = inputs
    ctx.save_for_backward(x, gamma, beta, output[1])


def _gn_backward(ctx, dy, dstats):
    x, gamma, beta, stats = ctx.saved_tensors
    return torch.ops.styler.groupnorm_relu_bwd(x, dy, gamma, beta, stats)


register_autograd("styler::groupnorm_relu", _gn_backward, setup_context=_gn_setup)


# ---- batchnorm_act (train mode: batch statistics; the updated running statistics are outputs) ---------------------------------------------------
def _bn_fwd(x, gamma, beta, running_mean, running_var, act, drop_p, drop_seed, segs):
    """Functional form (the dispatcher's autograd registration takes no mutable arguments): the updated running statistics are
    returned as outputs four and five; the caller copies them back (`nn.BatchNorm1d` semantics, momentum 0.1)."""
    rm, rv = running_mean.clone(), running_var.clone()
    y, mean, rstd = ops.batchnorm_train(x.contiguous(), gamma, beta, rm, rv, act, drop_p=drop_p, drop_seed=drop_seed, segs=segs)
    return y, mean, rstd, rm, rv


def _bn_bwd(x, dy, gamma, beta, mean, rstd, act, drop_p, drop_seed, segs):
    dg, db = torch.zeros_like(gamma), torch.zeros_like(beta)
    dx = ops.batchnorm_bwd(x.contiguous(), None, dy.contiguous(), gamma, mean, rstd, dg, db, act, beta=beta, drop_p=drop_p,
                           drop_seed=drop_seed, segs=segs)
    return dx, dg, db


_pair("batchnorm_act", "(Tensor x, Tensor gamma, Tensor beta, Tensor running_mean, Tensor running_var, int act, float drop_p, "
      "int drop_seed, int segs) -> (Tensor, Tensor, Tensor, Tensor, Tensor)", _bn_fwd,
      lambda x, gamma, beta, rm, rv, act, p, seed, segs: (torch.empty_like(x), x.new_empty(segs, x.shape[-1]),
                                                          x.new_empty(segs, x.shape[-1]), torch.empty_like(rm), torch.empty_like(rv)))
_pair("batchnorm_act_bwd", "(Tensor x, Tensor dy, Tensor gamma, Tensor beta, Tensor mean, Tensor rstd, int act, float drop_p, "
      "int drop_seed, int segs) -> (Tensor, Tensor, Tensor)", _bn_bwd,
      lambda x, dy, gamma, beta, mean, rstd, act, p, seed, segs: (torch.empty_like(x), torch.empty_like(gamma),
                                                                  torch.empty_like(beta)))


def _bn_setup(ctx, inputs, output):
    x, gamma, beta, rm, rv, act, p, seed, segs = inputs
    ctx.save_for_backward(x, gamma, beta, output[1], output[2])
    ctx.cfg = (act, p, seed, segs)


def _bn_backward(ctx, dy, dmean, drstd, drm, drv):
    x, gamma, beta, mean, rstd = ctx.saved_tensors
    dx, dg, db = torch.ops.styler.batchnorm_act_bwd(x, dy, gamma, beta, mean, rstd, *ctx.cfg)
    return dx, dg, db, None, None, None, None, None, None


register_autograd("styler::batchnorm_act", _bn_backward, setup_context=_bn_setup)


# ---- lstm_bidir: the recurrence of one bidirectional layer; gx = x W_ih^T + b of both directions ([B, S, 8H], gate order i f g o) --------
def _lstm_fwd(gx, w_hh, H):
    B, S, _ = gx.shape
    cell = torch.empty(B, S, 2 * H, device=gx.device, dtype=torch.float32)
    gates = torch.empty(B, S, 8 * H, device=gx.device, dtype=torch.float32)
    out = ops.lstm_bidir(gx.contiguous(), w_hh.contiguous(), H, cell_out=cell, gates_out=gates)
    return out, gates, cell


_pair("lstm_bidir", "(Tensor gx, Tensor w_hh, int H) -> (Tensor, Tensor, Tensor)", _lstm_fwd,
      lambda gx, w_hh, H: (gx.new_empty(gx.shape[0], gx.shape[1], 2 * H), torch.empty_like(gx),
                           gx.new_empty(gx.shape[0], gx.shape[1], 2 * H)))
_pair("lstm_bidir_bwd", "(Tensor dout, Tensor gates, Tensor cell, Tensor w_hh, int H) -> Tensor",
      lambda dout, gates, cell, w_hh, H: ops.lstm_bidir_bwd(dout, gates, cell, w_hh.contiguous(), H),
      lambda dout, gates, cell, w_hh, H: torch.empty_like(gates))


# ---- linear_ln: y = LayerNorm(a W^T + b + res) * pad mask, the pre-norm sum kept (eval form; no dropout) --------------------------------------
def _linear_ln_fwd(a, weight, bias, res, gamma, beta, lens):
    B, L, K = a.shape
    wb = ops.cast_bf16(weight.contiguous())
    a16 = a if a.dtype == torch.bfloat16 else ops.cast_bf16(a.contiguous()).view(B, L, K)
    s = torch.empty(B, L, 256, device=a.device, dtype=torch.float32)
    if not ops.linear_ln_ok(a16, weight.shape[0]):
        raise ops.StylerHipError("styler::linear_ln: n = 256, K % 64 == 0 and bf16-castable rows expected")
    y = ops.linear_ln(a16, wb, bias, res.contiguous(), gamma, beta, lens=lens, sum_out=s)
    return y, s


_pair("linear_ln", "(Tensor a, Tensor weight, Tensor bias, Tensor res, Tensor gamma, Tensor beta, Tensor? lens) -> (Tensor, Tensor)",
      _linear_ln_fwd, lambda a, weight, bias, res, gamma, beta, lens: (torch.empty_like(res), torch.empty_like(res)))


# ---- losses ----------------------------------------------------------------------------------------------------------------------------
def _mem_fwd(a, b, kind, lens):
    mean, acc = ops.masked_err_mean(a.contiguous(), b.contiguous(), kind, lens)
    return mean.view(()), acc


_pair("masked_err_mean", "(Tensor a, Tensor b, int kind, Tensor? lens) -> (Tensor, Tensor)", _mem_fwd,
      lambda a, b, kind, lens: (a.new_empty(()), a.new_empty(ops.MASKED_ACC_DOUBLES, dtype=torch.float64)))
_pair("masked_err_mean_bwd", "(Tensor a, Tensor b, Tensor acc, Tensor g, int kind, Tensor? lens) -> Tensor",
      lambda a, b, acc, g, kind, lens: ops.masked_err_bwd(a.contiguous(), b.contiguous(), acc, g.reshape(1).float().contiguous(), kind, lens),
      lambda a, b, acc, g, kind, lens: torch.empty_like(a))


def _mem_setup(ctx, inputs, output):
    a, b, kind, lens = inputs
    ctx.save_for_backward(a, b, output[1], *([lens] if lens is not None else []))
    ctx.kind, ctx.has_lens = kind, lens is not None


def _mem_backward(ctx, g, gacc):
    t = ctx.saved_tensors
    da = torch.ops.styler.masked_err_mean_bwd(t[0], t[1], t[2], g, ctx.kind, t[3] if ctx.has_lens else None)
    return da, None, None, None


register_autograd("styler::masked_err_mean", _mem_backward, setup_context=_mem_setup)

_pair("nll3", "(Tensor lp0, Tensor lp1, Tensor lp2, Tensor label) -> Tensor",
      lambda lp0, lp1, lp2, label: ops.nll3([lp0.contiguous(), lp1.contiguous(), lp2.contiguous()], label.contiguous()).view(()),
      lambda lp0, lp1, lp2, label: lp0.new_empty(()))


def _nll3_bwd(lp0, lp1, lp2, label, g):
    d3 = ops.nll3([lp0.contiguous(), lp1.contiguous(), lp2.contiguous()], label.contiguous(), gscale=g.reshape(1).float().contiguous(),
                  want_grad=True)
    return d3[0], d3[1], d3[2]


_pair("nll3_bwd", "(Tensor lp0, Tensor lp1, Tensor lp2, Tensor label, Tensor g) -> (Tensor, Tensor, Tensor)", _nll3_bwd,
      lambda lp0, lp1, lp2, label, g: (torch.empty_like(lp0), torch.empty_like(lp1), torch.empty_like(lp2)))


def _nll3_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _nll3_backward(ctx, g):
    lp0, lp1, lp2, label = ctx.saved_tensors
    return (*torch.ops.styler.nll3_bwd(lp0, lp1, lp2, label, g), None)


register_autograd("styler::nll3", _nll3_backward, setup_context=_nll3_setup)


# ---- dropout: y = x * keep / (1 - p), keep from the counter-based stream (seed, element index); the same call on dy is the backward -----
_pair("dropout", "(Tensor x, float p, int seed) -> Tensor", lambda x, p, seed: ops.dropout(x.contiguous(), p, seed).view(x.shape),
      lambda x, p, seed: torch.empty_like(x))


def _dropout_setup(ctx, inputs, output):
    ctx.p, ctx.seed = inputs[1], inputs[2]


register_autograd("styler::dropout", lambda ctx, dy: (torch.ops.styler.dropout(dy, ctx.p, ctx.seed), None, None),
                  setup_context=_dropout_setup)


# ---- embed_pos -------------------------------------------------------------------------------------------------------------------------
_pair("embed_pos", "(Tensor text, Tensor emb, Tensor pe) -> Tensor", lambda text, emb, pe: ops.embed_pos(text.contiguous(), emb, pe),
      lambda text, emb, pe: emb.new_empty(text.shape[0], text.shape[1], emb.shape[1]))


def _embed_pos_bwd(text, dy, V):
    demb = torch.zeros(V, dy.shape[-1], device=dy.device, dtype=torch.float32)
    ops.embed_bwd(text.contiguous(), dy.contiguous(), demb)
    return demb


_pair("embed_pos_bwd", "(Tensor text, Tensor dy, int V) -> Tensor", _embed_pos_bwd,
      lambda text, dy, V: dy.new_empty(V, dy.shape[-1]))


def _embed_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.V = inputs[1].shape[0]


register_autograd("styler::embed_pos", lambda ctx, dy: (None, torch.ops.styler.embed_pos_bwd(ctx.saved_tensors[0], dy, ctx.V), None),
                  setup_context=_embed_setup)


# ---- mel_calibrate ---------------------------------------------------------------------------------------------------------------------
_pair("mel_calibrate", "(Tensor x, Tensor mel_len, Tensor src_len, int S) -> Tensor",
      lambda x, mel_len, src_len, S: ops.mel_calibrate(x.contiguous(), mel_len.contiguous(), src_len.contiguous(), S),
      lambda x, mel_len, src_len, S: x.new_empty(x.shape[0], S, x.shape[2]))
_pair("mel_calibrate_bwd", "(Tensor dy, Tensor mel_len, Tensor src_len, int T) -> Tensor",
      lambda dy, mel_len, src_len, T: ops.mel_calibrate_bwd(dy.contiguous(), mel_len.contiguous(), src_len.contiguous(), T),
      lambda dy, mel_len, src_len, T: dy.new_empty(dy.shape[0], T, dy.shape[2]))


def _melcal_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1], inputs[2])
    ctx.T = inputs[0].shape[1]


register_autograd("styler::mel_calibrate",
                  lambda ctx, dy: (torch.ops.styler.mel_calibrate_bwd(dy, ctx.saved_tensors[0], ctx.saved_tensors[1], ctx.T), None, None, None),
                  setup_context=_melcal_setup)


# ---- aug_classifier_tail ---------------------------------------------------------------------------------------------------------------
_pair("aug_classifier_tail", "(Tensor h, Tensor ln_g, Tensor ln_b, Tensor w2, Tensor b2) -> Tensor",
      lambda h, ln_g, ln_b, w2, b2: ops.aug_classifier_tail(h.contiguous(), ln_g, ln_b, w2.contiguous(), b2),
      lambda h, ln_g, ln_b, w2, b2: h.new_empty(h.shape[0], 2))


def _aug_bwd(h, ln_g, ln_b, w2, b2, dout):
    dg, db, dw2, db2 = torch.zeros_like(ln_g), torch.zeros_like(ln_b), torch.zeros_like(w2), torch.zeros_like(b2)
    dh = ops.aug_classifier_tail_bwd(h.contiguous(), ln_g, ln_b, w2.contiguous(), b2, dout, dg, db, dw2, db2)
    return dh, dg, db, dw2, db2


_pair("aug_classifier_tail_bwd", "(Tensor h, Tensor ln_g, Tensor ln_b, Tensor w2, Tensor b2, Tensor dout) -> "
      "(Tensor, Tensor, Tensor, Tensor, Tensor)", _aug_bwd,
      lambda h, ln_g, ln_b, w2, b2, dout: (torch.empty_like(h), torch.empty_like(ln_g), torch.empty_like(ln_b), torch.empty_like(w2),
                                           torch.empty_like(b2)))


def _aug_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


register_autograd("styler::aug_classifier_tail", lambda ctx, dout: torch.ops.styler.aug_classifier_tail_bwd(*ctx.saved_tensors, dout),
                  setup_context=_aug_setup)


# ---- clip_adam_step: nn.utils.clip_grad_norm_(params, max_norm) + Adam over flat buffers, in place; returns the gradient norm -----------------
def _clip_adam(p, g, m, v, max_norm, lr, beta1, beta2, eps, step, grad_scale):
    ss = torch.zeros(1, device=p.device, dtype=torch.float64)
    ops.sumsq(g, ss)
    ops.adam_step(p, g, m, v, ss, max_norm, lr, beta1, beta2, eps, step, grad_scale=grad_scale)
    return ss.sqrt().float().view(())


_pair("clip_adam_step", "(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float max_norm, float lr, float beta1, float beta2, "
      "float eps, int step, float grad_scale) -> Tensor", _clip_adam,
      lambda p, g, m, v, max_norm, lr, beta1, beta2, eps, step, grad_scale: p.new_empty(()))


OPS = ("conv_gemm", "conv_gemm_bwd", "attention", "attention_bwd", "add_layernorm", "add_layernorm_bwd", "length_regulate",
       "length_regulate_bwd", "stft_mel", "groupnorm_relu", "groupnorm_relu_bwd", "batchnorm_act", "batchnorm_act_bwd", "lstm_bidir",
       "lstm_bidir_bwd", "linear_ln", "masked_err_mean", "masked_err_mean_bwd", "nll3", "nll3_bwd", "dropout", "embed_pos",
       "embed_pos_bwd", "mel_calibrate", "mel_calibrate_bwd", "aug_classifier_tail", "aug_classifier_tail_bwd", "clip_adam_step")
