"""The fused ops of the path as `torch.library` custom operators (namespace `styler`), SURVEY 8(b): schema'd,
dispatcher-visible entry points over libstyler_hip.so with registered autograd formulas and fake (meta) kernels, so that
the kernels compose with stock PyTorch code (`torch.ops.styler.conv_gemm(...)`, `torch.library.opcheck`, tracing):

    torch.ops.styler.conv_gemm(x, weight, bias, act, prec)               Linear / Conv1d('same') + activation
    torch.ops.styler.attention(qkv, lens, prec)                          4 x 64 masked self-attention (+ lse for backward)
    torch.ops.styler.add_layernorm(x, res, gamma, beta, lens)            LayerNorm(x + res) with pad mask
    torch.ops.styler.length_regulate(x, durations, max_len)              LengthRegulator expand (+ mel_len)
    torch.ops.styler.stft_mel(wav, basis, mel_basis, prec)               STFT -> log-mel / energy

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


def _conv_gemm_backward(ctx, dy):
    x, weight, y = ctx.saved_tensors
    kw = weight.shape[2] if weight.dim() == 3 else 1
    n, cin = weight.shape[0], weight.shape[1]
    dz = ops.act_bwd(dy.contiguous(), y, ctx.act) if ctx.act != ops.ACT_NONE else dy.contiguous()
    dw = torch.zeros_like(weight)
    db = torch.zeros(n, device=x.device, dtype=torch.float32) if ctx.has_bias else None
    ops.wgrad(dz, x.contiguous(), dw, n, cin, kw=kw, db=db, prec=ctx.prec)
    dx = ops.conv_gemm(dz, _kernel_weight_bwd(weight, ctx.prec), None, kw=kw, n=cin, prec=ctx.prec)
    return dx, dw, db, None, None


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


def _attention_backward(ctx, dout, dlse):
    qkv, lens, out, lse = ctx.saved_tensors
    return ops.attention_bwd(qkv.contiguous(), out, dout.contiguous(), lse, lens, prec=ctx.prec), None, None


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


def _add_layernorm_backward(ctx, dy, ds):
    saved = ctx.saved_tensors
    s, gamma, beta = saved[:3]
    lens = saved[3] if ctx.has_lens else None
    dg, db = torch.zeros_like(gamma), torch.zeros_like(beta)
    dx = ops.layernorm_bwd(s, dy.contiguous(), gamma, beta, dg, db, lens=lens)
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


def _length_regulate_backward(ctx, dy, dlen):
    (durations,) = ctx.saved_tensors
    B, S = durations.shape
    csum, _, _ = ops.duration_scan(B, S, dy.device, dur=durations.contiguous())
    return ops.length_regulate_bwd(dy.contiguous(), csum, ctx.S), None, None


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


OPS = ("conv_gemm", "attention", "add_layernorm", "length_regulate", "stft_mel")
