"""Autograd bindings: every differentiable op of the path is a torch.autograd.Function whose forward and
backward are libstyler_hip.so kernels.  torch.autograd only supplies the tape.

Parameter gradients are accumulated by the kernels (atomicAdd) straight into `param.grad` -- a view of the
flat fp32 gradient buffer owned by `training.TrainState` -- so the Functions return None for parameters
(the fused wgrad-accumulation scheme); activations' gradients flow through the tape as usual."""
import torch
from torch.autograd import Function

from . import ops
from .runtime import Seg, gemm_weight, gemm_weight_bwd, gemm_weight_bwd_auto, rt, seg_transposed, x3


def onehot_weight(cache, key, w):
    """Conv1d(257 -> C, k5) weight [C, 257, 5] -> gather layout [5, 257, C] of styler_onehot_conv5."""
    C, Q, K = w.shape
    return cache.get_spec(key, (K, Q, C), False, lambda: [Seg(w, (K, Q, C), (1, K, Q * K), (Q * C, C, 1))])


def lstm_wi_transposed(cache, key, wi, wir):
    """[cin, 8H] = transposed cat of the two directions' input weights (dX of the fused input projection); a bf16 shadow in
    throughput mode (rt.lstm_dx_bf16), like the dX weight of every other Linear.  Returns (weight, precision)."""
    n4, cin = wi.shape
    if rt.prec == ops.PREC_BF16X3 and (2 * n4) % 8 == 0 and cin % 4 == 0:
        w = cache.get_spec(key + "x3", (cin, 6 * n4), True,
                           lambda: x3([seg_transposed(wi, 0, 2 * n4), seg_transposed(wir, n4, 2 * n4)], 2 * n4))
        return w, ops.PREC_BF16X3
    bf16 = rt.prec == ops.PREC_BF16 and rt.lstm_dx_bf16 and (2 * n4) % 8 == 0 and cin % 4 == 0
    w = cache.get_spec(key + ("16" if bf16 else ""), (cin, 2 * n4), bf16,
                       lambda: [seg_transposed(wi, 0, 2 * n4), seg_transposed(wir, n4, 2 * n4)])
    return w, (ops.PREC_BF16 if bf16 else ops.PREC_F32)

NONE, RELU, TANH = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_TANH


def _x3_split(t, n, plan=None):
    """bf16x3 arithmetic: a gradient that feeds BOTH the weight gradient and the dX GEMM of its node is split into
    [hi | lo (| hi)] once (ops.split3; the single definition of the bf16x3 layouts is include/styler_hip.h, styler_split3_bf16) -> (the tensor to hand to ops.conv_gemm, the (hi, lo) views to hand to ops.wgrad as `dz_parts`).
    Any other arithmetic (or a width the bf16 engines do not take): (t, None)."""
    if rt.prec == ops.PREC_BF16X3 and n % 8 == 0 and t.dtype == torch.float32:
        t3 = ops.split3(t, plan)
        return t3, ops.split3_parts(t3, n)
    return t, None


def G(p):
    """Gradient buffer of a parameter (allocated zeroed on first use when no TrainState owns it)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


_neg_ones = {}


def _neg(n, device):
    key = (n, str(device))
    if key not in _neg_ones:
        _neg_ones[key] = torch.full((n,), -1.0, device=device)
    return _neg_ones[key]


class ConvGemmFn(Function):
    """y = act(conv_same(x, W) + b) (+ res when act is NONE).  W / b are nn.Parameters in reference layout."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, cache, key, kw, act, neg_dx, plan=None):
        w, prec = gemm_weight(cache, key, weight, x.shape[-1])
        fuse_res = res is not None and act == NONE
        y = ops.conv_gemm(x, w, bias, kw=kw, n=weight.shape[0], act=act, prec=prec, res=res if fuse_res else None,
                          plan=plan)
        ctx.save_for_backward(x, y if act != NONE else None)
        ctx.weight, ctx.bias, ctx.cache, ctx.key, ctx.plan = weight, bias, cache, key, plan
        ctx.kw, ctx.act, ctx.neg_dx, ctx.has_res = kw, act, neg_dx, res is not None
        if res is not None and not fuse_res:
            return ops.add2(y, res)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias, kw = ctx.weight, ctx.bias, ctx.kw
        n, cin = weight.shape[0], x.shape[-1]
        dy = ops._rows_view(dy)
        plan = ctx.plan
        dz = ops.act_bwd(dy, y, ctx.act, lens=plan.nrows if plan is not None else None) if ctx.act != NONE else dy
        dzg, dzp = _x3_split(dz, n, plan)
        if weight.requires_grad:
            ops.wgrad(dz, x, G(weight), n, cin, kw=kw,
                      db=G(bias) if (bias is not None and bias.requires_grad) else None, plan=plan, dz_parts=dzp)
        dx = None
        if ctx.needs_input_grad[0]:
            wt, prec = gemm_weight_bwd_auto(ctx.cache, ctx.key, weight)
            dz = dzg if prec == ops.PREC_BF16X3 else dz
            # a bf16 activation gets a bf16 gradient (autograd would cast an fp32 one to the input's dtype anyway -- with a
            # kernel of its own): written as bf16 by the GEMM, read as bf16 by the norm backward that consumes it
            dx = ops.conv_gemm(dz, wt, None, kw=kw, n=cin, prec=prec,
                               scale=_neg(cin, dz.device) if ctx.neg_dx else None, plan=plan,
                               out_bf16=x.dtype == torch.bfloat16 and prec == ops.PREC_BF16)
        return dx, (dy if ctx.has_res else None), None, None, None, None, None, None, None, None


class ConvGemmMultiFn(Function):
    """K INDEPENDENT Linear layers y_k = act_k(x_k W_k^T + b_k) as ONE tape node: one grouped launch forward
    (ops.conv_gemm_multi), one grouped launch for the K dX GEMMs backward; the weight gradients join the step's grouped
    weight-gradient launch as before.  apply(metas, x_0, W_0, b_0, x_1, W_1, b_1, ...) with metas[k] = (cache, key, act,
    neg_dx) -> tuple of K outputs.  (The first / second Linear of the four style MLPs, modules.py:250-271,335-348; the three
    classifiers' first Linear, modules.py:38-45.)"""

    @staticmethod
    def forward(ctx, metas, *flat):
        xs, ws, bs = flat[0::3], flat[1::3], flat[2::3]
        calls = []
        for (cache, key, act, _), x, weight, bias in zip(metas, xs, ws, bs):
            w, prec = gemm_weight(cache, key, weight, x.shape[-1])
            calls.append(dict(x=x, w=w, bias=bias, n=weight.shape[0], act=act, prec=prec))
        ys = ops.conv_gemm_multi(calls)
        ctx.save_for_backward(*xs, *[y if m[2] != NONE else x.new_empty(0) for y, m, x in zip(ys, metas, xs)])
        ctx.metas, ctx.ws, ctx.bs = metas, ws, bs
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        K = len(ctx.metas)
        saved = ctx.saved_tensors
        xs, ys = saved[:K], saved[K:]
        calls, slots = [], []
        grads = [None] * (3 * K)
        live = [k for k in range(K) if dys[k] is not None and ctx.metas[k][2] != NONE]
        dzs = dict(zip(live, ops.act_bwd_multi([(dys[k], ys[k], ctx.metas[k][2]) for k in live]))) if live else {}
        for k, ((cache, key, act, neg_dx), x, y, weight, bias, dy) in enumerate(zip(ctx.metas, xs, ys, ctx.ws, ctx.bs, dys)):
            if dy is None:
                continue
            n, cin = weight.shape[0], x.shape[-1]
            dz = dzs[k] if act != NONE else ops._rows_view(dy)
            dzg, dzp = _x3_split(dz, n)
            if weight.requires_grad:
                ops.wgrad(dz, x, G(weight), n, cin, db=G(bias) if (bias is not None and bias.requires_grad) else None,
                          dz_parts=dzp)
            if ctx.needs_input_grad[1 + 3 * k]:
                wt, prec = gemm_weight_bwd_auto(cache, key, weight)
                calls.append(dict(x=dzg if prec == ops.PREC_BF16X3 else dz, w=wt, n=cin, prec=prec,
                                  scale=_neg(cin, dz.device) if neg_dx else None))
                slots.append(k)
        for k, dx in zip(slots, ops.conv_gemm_multi(calls) if calls else []):
            grads[3 * k] = dx
        return (None, *grads)


class LayerNormFn(Function):
    """LayerNorm(x + res) with pad mask (SubLayers.py:59,87 + Layers.py:29,32)."""

    @staticmethod
    def forward(ctx, x, res, anchor, ln, lens, drop_p=0.0):
        """LayerNorm(dropout(x, drop_p) + res): the dropout, the residual add, the normalisation and the pad mask are
        one kernel, which also writes the pre-norm sum the backward needs."""
        drop_p = 0.0 if rt.disable_dropout else drop_p
        seed = next_dropout_seed() if drop_p > 0 else 0
        if res is not None or drop_p > 0:
            s = torch.empty_like(x)
            y = ops.add_layernorm(x, ln.weight, ln.bias, res=res, lens=lens, in_drop_p=drop_p, in_drop_seed=seed,
                                  sum_out=s)
        else:
            s = x
            y = ops.add_layernorm(s, ln.weight, ln.bias, lens=lens)
        ctx.save_for_backward(s, lens)
        ctx.ln, ctx.has_res, ctx.drop = ln, res is not None, (drop_p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        s, lens = ctx.saved_tensors
        ln = ctx.ln
        if ctx.drop[0] > 0:
            dx, dxd = ops.layernorm_bwd(s, dy, ln.weight, ln.bias, G(ln.weight), G(ln.bias), lens=lens,
                                        in_drop_p=ctx.drop[0], in_drop_seed=ctx.drop[1])
            return dxd, (dx if ctx.has_res else None), None, None, None, None
        dx = ops.layernorm_bwd(s, dy, ln.weight, ln.bias, G(ln.weight), G(ln.bias), lens=lens)
        return dx, (dx if ctx.has_res else None), None, None, None, None


def _ln_tail_fwd(o, x, ln, lens, drop_p, want16=False, plan=None):
    """dropout(o) + x -> LayerNorm -> pad mask in one kernel; returns (y, pre-norm sum, (p, seed)) -- and with `want16` a
    fourth value, the bf16 copy of y for the GEMM that consumes it."""
    drop_p = 0.0 if rt.disable_dropout else drop_p
    seed = next_dropout_seed() if drop_p > 0 else 0
    s = torch.empty_like(x)                           # bf16 when the residual stream is (x bf16): y comes out as bf16 too
    y16 = torch.empty_like(x, dtype=torch.bfloat16) if (want16 and x.dtype != torch.bfloat16) else None
    y = ops.add_layernorm(o, ln.weight, ln.bias, res=x, lens=lens, in_drop_p=drop_p, in_drop_seed=seed, sum_out=s,
                          out16=y16, x3=True, plan=plan)   # (bf16x3: y feeds the next sublayer's GEMM -- split written here)
    if rt.sim_bf16_stream and rt.prec == ops.PREC_BF16 and x.dtype != torch.bfloat16:
        y, s = _r16(y), _r16(s)
    return (y, s, (drop_p, seed), y16) if want16 else (y, s, (drop_p, seed))


def _linear_ln_fwd(a, w, bias, x, ln, lens, drop_p, want16=False, plan=None):
    """_ln_tail_fwd with the Linear in front of it in the same launch (ops.linear_ln, csrc/linear_ln.hip): the same seed
    sequence, the same saved sum, so _ln_tail_bwd and the Linear's own backward GEMMs run unchanged."""
    drop_p = 0.0 if rt.disable_dropout else drop_p
    seed = next_dropout_seed() if drop_p > 0 else 0
    s = torch.empty_like(x)
    y16 = torch.empty_like(x, dtype=torch.bfloat16) if (want16 and x.dtype != torch.bfloat16) else None
    y = ops.linear_ln(a, w, bias, x, ln.weight, ln.bias, lens=lens, drop_p=drop_p, drop_seed=seed, sum_out=s, out16=y16,
                      packed=plan is not None)
    return (y, s, (drop_p, seed), y16) if want16 else (y, s, (drop_p, seed))


def _linear_ln_takes(a, prec):
    """The fused Linear + LayerNorm launch: throughput mode, bf16 rows, no simulated stream."""
    return rt.linear_ln and prec == ops.PREC_BF16 and not rt.sim_bf16_stream and ops.linear_ln_ok(a, 256)


def _r16(t):
    """rt.sim_bf16_stream: the value a bf16-stored tensor would hold."""
    return t.to(torch.bfloat16).to(torch.float32)


def _ln_tail_bwd(s, dy, ln, lens, drop, plan=None):
    """-> (gradient of the residual input, gradient of the dropout branch)."""
    sim = rt.sim_bf16_stream and rt.prec == ops.PREC_BF16 and s.dtype != torch.bfloat16
    if sim:
        dy = _r16(dy)
    if drop[0] > 0:
        dx, dxd = ops.layernorm_bwd(s, dy, ln.weight, ln.bias, G(ln.weight), G(ln.bias), lens=lens, in_drop_p=drop[0],
                                    in_drop_seed=drop[1], x3=True, plan=plan)
        return (_r16(dx), _r16(dxd)) if sim else (dx, dxd)
    dx = ops.layernorm_bwd(s, dy, ln.weight, ln.bias, G(ln.weight), G(ln.bias), lens=lens, x3=True, plan=plan)
    if sim:
        dx = _r16(dx)
    return dx, dx


class FfnSublayerFn(Function):
    """PositionwiseFeedForward with its residual (SubLayers.py:81-89) as ONE tape node: Conv1d(k=9)+ReLU, Conv1d(k=1),
    dropout, + x, LayerNorm, pad mask.  Backward keeps everything in GEMM epilogues: the ReLU mask rides on the dX GEMM of
    the second conv, the residual's gradient is added in the epilogue of the first conv's dX GEMM (no act_bwd launch, no
    autograd accumulation kernel)."""

    @staticmethod
    def forward(ctx, x, anchor, ffn, lens, plan, drop_p, x16=None):
        kw1, kw2 = ffn.w_1.weight.shape[2], ffn.w_2.weight.shape[2]
        w1, p1 = gemm_weight(ffn._derived, "w_1", ffn.w_1.weight, x.shape[-1])
        # throughput mode: the hidden activation (and its gradient in backward) live in HBM as bf16 -- they are only ever
        # consumed as bf16 MFMA operands or as a sign mask, so this changes no result and halves the widest tensors
        h16 = p1 == ops.PREC_BF16 and ffn.w_1.weight.shape[0] % 8 == 0
        # x16: the bf16 copy of x written by the LayerNorm that produced it (same values the GEMM would round x to)
        xa = x16 if (x16 is not None and p1 == ops.PREC_BF16) else x
        h = ops.conv_gemm(xa, w1, ffn.w_1.bias, kw=kw1, n=ffn.w_1.weight.shape[0], act=RELU, prec=p1, plan=plan,
                          out_bf16=h16, x3_out=True)             # (bf16x3: h is the second conv's operand -- split in the epilogue)
        w2, p2 = gemm_weight(ffn._derived, "w_2", ffn.w_2.weight, h.shape[-1])
        if kw2 == 1 and ffn.w_2.weight.shape[0] == 256 and _linear_ln_takes(h, p2):
            y, s, ctx.drop = _linear_ln_fwd(h, w2, ffn.w_2.bias, x, ffn.layer_norm, lens, drop_p, plan=plan)
        else:
            o = ops.conv_gemm(h, w2, ffn.w_2.bias, kw=kw2, n=ffn.w_2.weight.shape[0], prec=p2, plan=plan)
            y, s, ctx.drop = _ln_tail_fwd(o, x, ffn.layer_norm, lens, drop_p, plan=plan)
        ctx.save_for_backward(x, h, s, lens)
        ctx.ffn, ctx.plan = ffn, plan
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, s, lens = ctx.saved_tensors
        ffn, plan = ctx.ffn, ctx.plan
        w_1, w_2 = ffn.w_1, ffn.w_2
        kw1, kw2 = w_1.weight.shape[2], w_2.weight.shape[2]
        d_hid, d_in = w_1.weight.shape[0], w_2.weight.shape[0]
        dx_res, d_o = _ln_tail_bwd(s, ops._rows_view(dy), ffn.layer_norm, lens, ctx.drop, plan=plan)
        d_og, d_op = _x3_split(d_o, d_in, plan)
        ops.wgrad(d_o, h, G(w_2.weight), d_in, d_hid, kw=kw2, db=G(w_2.bias), plan=plan, dz_parts=d_op)
        wt2, prec2 = gemm_weight_bwd_auto(ffn._derived, "w_2", w_2.weight)
        dh = ops.conv_gemm(d_og if prec2 == ops.PREC_BF16X3 else d_o, wt2, None, kw=kw2, n=d_hid, prec=prec2, plan=plan,
                           mask=h, out_bf16=h.dtype == torch.bfloat16, x3_out=True)
        dhg, dhp = _x3_split(dh, d_hid, plan)
        ops.wgrad(dh, x, G(w_1.weight), d_hid, d_in, kw=kw1, db=G(w_1.bias), plan=plan, dz_parts=dhp)
        wt1, prec1 = gemm_weight_bwd_auto(ffn._derived, "w_1", w_1.weight)
        dx = ops.conv_gemm(dhg if prec1 == ops.PREC_BF16X3 else dh, wt1, None, kw=kw1, n=d_in, prec=prec1, plan=plan,
                           res=dx_res, out_bf16=x.dtype == torch.bfloat16)      # bf16 stream: bf16 gradient
        return dx, None, None, None, None, None, None


class AttnSublayerFn(Function):
    """MultiHeadAttention with its residual (SubLayers.py:31-61) as ONE tape node: fused QKV GEMM, attention, output
    projection, dropout, + x, LayerNorm, pad mask; the residual's gradient is added in the epilogue of the QKV dX GEMM."""

    @staticmethod
    def forward(ctx, x, anchor, mha, lens, plan, drop_p, want16=False):
        w, b, prec = mha._qkv()
        qkv = ops.conv_gemm(x, w, b, n=768, prec=prec, plan=plan, out_bf16=prec == ops.PREC_BF16 and rt.bf16_qkv)
        B, L = (plan.B, plan.T) if plan is not None else x.shape[:2]
        lse = torch.empty(B, 4, L, device=x.device, dtype=torch.float32)
        att = ops.attention_fwd(qkv, lens, lse=lse, plan=plan, out_bf16=prec == ops.PREC_BF16 and rt.bf16_att, x3=True)
        wfc, pfc = gemm_weight(mha._derived, "fc", mha.fc.weight, 256)
        want16 = want16 and prec == ops.PREC_BF16
        if _linear_ln_takes(att, pfc):                # output projection + dropout + residual + LayerNorm: one launch
            r = _linear_ln_fwd(att, wfc, mha.fc.bias, x, mha.layer_norm, lens, drop_p, want16=want16, plan=plan)
            (y, s, ctx.drop, y16) = r if want16 else (r + (None,))
        else:
            o = ops.conv_gemm(att, wfc, mha.fc.bias, n=256, prec=pfc, plan=plan)
            if want16:                                # (on a bf16 stream the second value is None: y itself is bf16)
                y, s, ctx.drop, y16 = _ln_tail_fwd(o, x, mha.layer_norm, lens, drop_p, want16=True, plan=plan)
            else:
                y, s, ctx.drop = _ln_tail_fwd(o, x, mha.layer_norm, lens, drop_p, plan=plan)
        ctx.save_for_backward(x, qkv, att, lse, s, lens)
        ctx.mha, ctx.plan = mha, plan
        if want16:
            ctx.mark_non_differentiable(y16)
            ctx.set_materialize_grads(False)            # (no zero-filled bf16 tensor for the copy's unused gradient slot)
            return y, y16
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        x, qkv, att, lse, s, lens = ctx.saved_tensors
        mha, plan = ctx.mha, ctx.plan
        dx_res, d_o = _ln_tail_bwd(s, ops._rows_view(dy), mha.layer_norm, lens, ctx.drop, plan=plan)
        bf16 = rt.prec == ops.PREC_BF16
        prec = ops.PREC_BF16 if bf16 else ops.PREC_F32
        d_og, d_op = _x3_split(d_o, 256, plan)
        ops.wgrad(d_o, att, G(mha.fc.weight), 256, 256, db=G(mha.fc.bias), plan=plan, dz_parts=d_op)
        wfc, pfc = gemm_weight_bwd_auto(mha._derived, "fc", mha.fc.weight)
        d_att = ops.conv_gemm(d_og if pfc == ops.PREC_BF16X3 else d_o, wfc, None, n=256, prec=pfc, plan=plan,
                              out_bf16=bf16 and att.dtype == torch.bfloat16)
        dqkv = ops.attention_bwd(qkv, att, d_att, lse, lens, plan=plan, out_bf16=bf16 and rt.bf16_acts and rt.bf16_dqkv,
                                 x3=True)
        srcs = [mha.w_qs.weight, mha.w_ks.weight, mha.w_vs.weight]
        dqg, _ = _x3_split(dqkv, 768, plan)           # [hi(768) | lo(768) (| hi)]: per projection, slices of both parts
        x3p = ops.split3_parts(ops.split3(x, plan), 256) if dqg is not dqkv else None
        for i, lin in enumerate((mha.w_qs, mha.w_ks, mha.w_vs)):
            dzp = ((dqg[..., i * 256:(i + 1) * 256], dqg[..., 768 + i * 256:768 + (i + 1) * 256]) if dqg is not dqkv else None)
            ops.wgrad(dqkv[..., i * 256:(i + 1) * 256], x, G(lin.weight), 256, 256, db=G(lin.bias), plan=plan,
                      dz_parts=dzp, x_parts=x3p)
        if rt.prec == ops.PREC_BF16X3:
            prec = ops.PREC_BF16X3
            wt = mha._derived.get_spec("qkv_wTx3", (256, 3 * 768), True,
                                       lambda: x3([seg_transposed(w, k * 256, 768) for k, w in enumerate(srcs)], 768))
        else:
            wt = mha._derived.get_spec("qkv_wT16" if bf16 else "qkv_wT", (256, 768), bf16,
                                       lambda: [seg_transposed(w, k * 256, 768) for k, w in enumerate(srcs)])
        dx = ops.conv_gemm(dqg if prec == ops.PREC_BF16X3 else dqkv, wt, None, n=256, prec=prec, plan=plan, res=dx_res,
                           out_bf16=x.dtype == torch.bfloat16)
        return dx, None, None, None, None, None, None


class LayerNormDotFn(Function):
    """StylePredictor tail: LayerNorm -> Linear(256,1) -> masked_fill (modules.py:449-465)."""

    @staticmethod
    def forward(ctx, x, anchor, ln, lin, lens, drop_p, seed):
        out = ops.add_layernorm(x, ln.weight, ln.bias, lens=lens, dot_w=lin.weight, dot_b=lin.bias, drop_p=drop_p,
                                drop_seed=seed)
        ctx.save_for_backward(x, lens)
        ctx.ln, ctx.lin, ctx.drop = ln, lin, (drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, lens = ctx.saved_tensors
        ln, lin = ctx.ln, ctx.lin
        dx = ops.layernorm_bwd(x, None, ln.weight, ln.bias, G(ln.weight), G(ln.bias), lens=lens, dot_w=lin.weight,
                               dout=dout, ddot_w=G(lin.weight), ddot_b=G(lin.bias), drop_p=ctx.drop[0],
                               drop_seed=ctx.drop[1])
        return dx, None, None, None, None, None, None


class PredictorStageFn(Function):
    """One stage of the StylePredictor (modules.py:430-447) as ONE tape node: Conv1d(k) -> ReLU -> LayerNorm -> dropout,
    and with `lin` the second stage with its tail -> Linear(256, 1) -> pad mask.  Forward: the conv GEMM (ReLU in its
    epilogue) and one LayerNorm kernel that also drops.  Backward: ONE LayerNorm-backward kernel regenerates the dropout
    mask, and -- its input being the ReLU output -- hands back the gradient w.r.t. the conv's pre-activation output; as
    separate nodes this was a dropout kernel forward and backward and an act_bwd pass per stage."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, key, kw, ln, lin, lens, drop_p):
        drop_p = 0.0 if rt.disable_dropout else drop_p
        seed = next_dropout_seed() if drop_p > 0 else 0
        w, prec = gemm_weight(cache, key, weight, x.shape[-1])
        h = ops.conv_gemm(x, w, bias, kw=kw, n=weight.shape[0], act=RELU, prec=prec)
        if lin is None:
            out = ops.add_layernorm(h, ln.weight, ln.bias, drop_p=drop_p, drop_seed=seed)
        else:
            out = ops.add_layernorm(h, ln.weight, ln.bias, lens=lens, dot_w=lin.weight, dot_b=lin.bias, drop_p=drop_p,
                                    drop_seed=seed)
        ctx.save_for_backward(x, h, lens)
        ctx.weight, ctx.bias, ctx.cache, ctx.key, ctx.kw = weight, bias, cache, key, kw
        ctx.ln, ctx.lin, ctx.drop = ln, lin, (drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h, lens = ctx.saved_tensors
        weight, bias, kw, ln, lin = ctx.weight, ctx.bias, ctx.kw, ctx.ln, ctx.lin
        if lin is None:
            dz = ops.layernorm_bwd(h, dout, ln.weight, ln.bias, G(ln.weight), G(ln.bias), drop_p=ctx.drop[0],
                                   drop_seed=ctx.drop[1], relu_input=True)
        else:
            dz = ops.layernorm_bwd(h, None, ln.weight, ln.bias, G(ln.weight), G(ln.bias), lens=lens, dot_w=lin.weight,
                                   dout=dout, ddot_w=G(lin.weight), ddot_b=G(lin.bias), drop_p=ctx.drop[0],
                                   drop_seed=ctx.drop[1], relu_input=True)
        n, cin = weight.shape[0], x.shape[-1]
        dzg, dzp = _x3_split(dz, n)
        ops.wgrad(dz, x, G(weight), n, cin, kw=kw, db=G(bias) if bias is not None else None, dz_parts=dzp)
        dx = None
        if ctx.needs_input_grad[0]:
            wt, wprec = gemm_weight_bwd_auto(ctx.cache, ctx.key, weight)
            dx = ops.conv_gemm(dzg if wprec == ops.PREC_BF16X3 else dz, wt, None, kw=kw, n=cin, prec=wprec)
        return dx, None, None, None, None, None, None, None, None, None


class ConvNormFn(Function):
    """norm_act(conv_same(x, W) + b) as ONE tape node: Conv1d -> GroupNorm + ReLU (AudioEncoder, modules.py:103-113) or
    Conv1d -> train-mode BatchNorm1d (+ tanh) + dropout (PostNet, Layers.py:91-128).

    One node because of what lives between its kernels in throughput mode: the gradient w.r.t. the convolution's output is
    consumed only by the dX GEMM and the weight-gradient kernel, which round it to bf16 first, so the norm backward writes
    it as bf16 (same results, half the bytes for all three kernels).  As a gradient crossing the tape it would be cast back
    to the fp32 of the convolution's output by autograd.  `out_bf16`: the activation y is stored as bf16 as well (its only
    consumers are the next convolution and that convolution's weight gradient); the gradient that comes back for it is then
    bf16 too (written by the next node's dX GEMM, read here by the norm backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, key, kw, norm, kind, act, drop_p, segs, out_bf16):
        w, prec = gemm_weight(cache, key, weight, x.shape[-1])
        b16 = prec == ops.PREC_BF16 and rt.bf16_acts and weight.shape[0] % 8 == 0
        # the convolution output itself as bf16 (rt.bf16_z): read by the norm forward and again by its backward
        z16 = b16 and rt.bf16_z and (kind != "gn" or ops.groupnorm_z_bf16_ok(x.shape[1]))
        z = ops.conv_gemm(x, w, bias, kw=kw, n=weight.shape[0], prec=prec, out_bf16=z16)
        out_bf16 = out_bf16 and b16
        if kind == "gn":
            aux = torch.empty(z.shape[0], z.shape[2] // 16, 2, device=z.device, dtype=torch.float32)
            y = ops.groupnorm_relu(z, norm.weight, norm.bias, stats=aux, x3=True,    # (bf16x3: y feeds the next conv's GEMM)
                                   out=torch.empty_like(z, dtype=torch.bfloat16 if out_bf16 else torch.float32))
            ctx.save_for_backward(x, z, aux)
            ctx.drop = (0.0, 0)
        else:
            drop_p = 0.0 if rt.disable_dropout else drop_p
            seed = next_dropout_seed() if drop_p > 0 else 0
            y, mean, rstd = ops.batchnorm_train(z, norm.weight, norm.bias, norm.running_mean, norm.running_var, act,
                                                drop_p=drop_p, drop_seed=seed, segs=segs, out_bf16=out_bf16, x3=True)
            ctx.save_for_backward(x, z, mean, rstd)
            ctx.drop = (drop_p, seed)
        ctx.weight, ctx.bias, ctx.cache, ctx.key, ctx.kw = weight, bias, cache, key, kw
        ctx.norm, ctx.kind, ctx.act, ctx.segs, ctx.b16 = norm, kind, act, segs, b16
        return y

    @staticmethod
    def backward(ctx, dy):
        norm, weight, bias, kw = ctx.norm, ctx.weight, ctx.bias, ctx.kw
        if ctx.kind == "gn":
            x, z, stats = ctx.saved_tensors
            dz = ops.groupnorm_relu_bwd(z, dy, norm.weight, norm.bias, stats, G(norm.weight), G(norm.bias), dx_bf16=ctx.b16,
                                        x3=True)     # (bf16x3: dz feeds the dX GEMM and the weight gradient)
        else:
            x, z, mean, rstd = ctx.saved_tensors
            dz = ops.batchnorm_bwd(z, None, dy, norm.weight, mean, rstd, G(norm.weight), G(norm.bias), ctx.act,
                                   beta=norm.bias, drop_p=ctx.drop[0], drop_seed=ctx.drop[1], segs=ctx.segs,
                                   dx_bf16=ctx.b16, x3=True)
        n, cin = weight.shape[0], x.shape[-1]
        dzg, dzp = _x3_split(dz, n)
        if weight.requires_grad:
            ops.wgrad(dz, x, G(weight), n, cin, kw=kw, db=G(bias) if (bias is not None and bias.requires_grad) else None,
                      dz_parts=dzp)
        dx = None
        if ctx.needs_input_grad[0]:
            wt, wprec = gemm_weight_bwd_auto(ctx.cache, ctx.key, weight)
            dx = ops.conv_gemm(dzg if wprec == ops.PREC_BF16X3 else dz, wt, None, kw=kw, n=cin, prec=wprec,
                               out_bf16=wprec == ops.PREC_BF16 and x.dtype == torch.bfloat16)
        return (dx,) + (None,) * 11


class ConvNormCatFn(Function):
    """The LAST conv -> GroupNorm -> ReLU stage of the AudioEncoder's four streams (modules.py:128-160) as one tape node
    whose GroupNorm kernels write straight into the channel slices of the concatenated [B, T, sum C] buffer the
    mel calibrator reads (modules.py:176-177) -- the reference's torch.cat, and the four strided copies that stood in for
    it here (0.1 ms per step), are gone; the backward hands each stream its slice of the incoming gradient as a view.
    apply(cache, keys, norms, x0, w0, b0, x1, w1, b1, ...)."""

    @staticmethod
    def forward(ctx, cache, keys, norms, *flat):
        xs, ws, bs = flat[0::3], flat[1::3], flat[2::3]
        B, L = xs[0].shape[:2]                        # (a stream may cover only the first Bs < B items: AudioEncoder.noise_items)
        widths = [w.shape[0] for w in ws]
        # throughput mode (rt.bf16_cat): the concatenation itself is bf16 -- its only reader, the mel calibrator, averages it in
        # fp32, and the gradient that comes back for it is bf16 too (read by the four GroupNorm backwards)
        cat16 = rt.bf16_cat and rt.bf16_acts and rt.prec == ops.PREC_BF16 and all(wd % 8 == 0 for wd in widths)
        out = torch.empty(B, L, sum(widths), device=xs[0].device, dtype=torch.bfloat16 if cat16 else torch.float32)
        saved, b16s, off = [], [], 0
        for x, weight, bias, key, norm, wd in zip(xs, ws, bs, keys, norms, widths):
            w, prec = gemm_weight(cache, key, weight, x.shape[-1])
            b16 = prec == ops.PREC_BF16 and rt.bf16_acts and wd % 8 == 0
            z16 = b16 and rt.bf16_z and ops.groupnorm_z_bf16_ok(L)
            z = ops.conv_gemm(x, w, bias, kw=5, n=wd, prec=prec, out_bf16=z16)
            Bs = x.shape[0]
            aux = torch.empty(Bs, wd // 16, 2, device=z.device, dtype=torch.float32)
            ops.groupnorm_relu(z, norm.weight, norm.bias, stats=aux, out=out[:Bs, :, off:off + wd])
            if Bs < B:                                # the items this stream skips: zeros (finite inputs for the calibrator / BiLSTM)
                ops.fill_zero(out[Bs:, :, off:off + wd])
            saved += [x, z, aux]
            b16s.append(b16)
            off += wd
        ctx.save_for_backward(*saved)
        ctx.meta = (cache, keys, norms, ws, bs, widths, b16s)
        return out

    @staticmethod
    def backward(ctx, dy):
        cache, keys, norms, ws, bs, widths, b16s = ctx.meta
        saved = ctx.saved_tensors
        grads, off = [], 0
        for i, (weight, bias, key, norm, wd, b16) in enumerate(zip(ws, bs, keys, norms, widths, b16s)):
            x, z, stats = saved[3 * i:3 * i + 3]
            dz = ops.groupnorm_relu_bwd(z, dy[:x.shape[0], :, off:off + wd], norm.weight, norm.bias, stats, G(norm.weight),
                                        G(norm.bias), dx_bf16=b16, x3=True)
            off += wd
            n, cin = weight.shape[0], x.shape[-1]
            dzg, dzp = _x3_split(dz, n)
            if weight.requires_grad:
                ops.wgrad(dz, x, G(weight), n, cin, kw=5, db=G(bias) if (bias is not None and bias.requires_grad) else None,
                          dz_parts=dzp)
            dx = None
            if ctx.needs_input_grad[3 + 3 * i]:
                wt, wprec = gemm_weight_bwd_auto(cache, key, weight)
                dx = ops.conv_gemm(dzg if wprec == ops.PREC_BF16X3 else dz, wt, None, kw=5, n=cin, prec=wprec,
                                   out_bf16=wprec == ops.PREC_BF16 and x.dtype == torch.bfloat16)
            grads += [dx, None, None]
        return (None, None, None) + tuple(grads)


class GroupNormReluFn(Function):
    """relu(GroupNorm(x)).  `out_bf16`: y is stored as bf16 (throughput mode, when the only consumer is the next convolution
    -- which rounds its activation operand to bf16 anyway); `dx_bf16`: so is the gradient handed to the convolution that
    produced x (its dX GEMM and weight gradient round it to bf16 anyway)."""

    @staticmethod
    def forward(ctx, x, anchor, gn, out_bf16=False, dx_bf16=False):
        stats = torch.empty(x.shape[0], x.shape[2] // 16, 2, device=x.device, dtype=torch.float32)
        out = torch.empty_like(x, dtype=torch.bfloat16) if out_bf16 else torch.empty_like(x)
        y = ops.groupnorm_relu(x, gn.weight, gn.bias, out=out, stats=stats)
        ctx.save_for_backward(x, stats)
        ctx.gn, ctx.dx_bf16 = gn, dx_bf16
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        gn = ctx.gn
        return ops.groupnorm_relu_bwd(x, dy, gn.weight, gn.bias, stats, G(gn.weight), G(gn.bias),
                                      dx_bf16=ctx.dx_bf16), None, None, None, None


class BatchNormActFn(Function):
    """dropout(act(BatchNorm1d_train(x))) (Layers.py:91-128) in one pass; backward regenerates the dropout mask and
    recomputes the tanh output from x instead of saving it.  `out_bf16` / `dx_bf16` as in GroupNormReluFn."""

    @staticmethod
    def forward(ctx, x, anchor, bn, act, drop_p=0.0, segs=1, out_bf16=False, dx_bf16=False):
        drop_p = 0.0 if rt.disable_dropout else drop_p
        seed = next_dropout_seed() if drop_p > 0 else 0
        y, mean, rstd = ops.batchnorm_train(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, act,
                                            drop_p=drop_p, drop_seed=seed, segs=segs, out_bf16=out_bf16)
        ctx.save_for_backward(x, mean, rstd)
        ctx.bn, ctx.act, ctx.drop, ctx.segs, ctx.dx_bf16 = bn, act, (drop_p, seed), segs, dx_bf16
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        bn = ctx.bn
        dx = ops.batchnorm_bwd(x, None, dy, bn.weight, mean, rstd, G(bn.weight), G(bn.bias), ctx.act, beta=bn.bias,
                               drop_p=ctx.drop[0], drop_seed=ctx.drop[1], segs=ctx.segs, dx_bf16=ctx.dx_bf16)
        return dx, None, None, None, None, None, None, None


class EmbedPosFn(Function):
    @staticmethod
    def forward(ctx, text, anchor, emb, pe):
        ctx.save_for_backward(text)
        ctx.emb = emb
        return ops.embed_pos(text, emb.weight, pe)

    @staticmethod
    def backward(ctx, dy):
        (text,) = ctx.saved_tensors
        ops.embed_bwd(text, dy, G(ctx.emb.weight))
        if rt.text_ready_hook is not None:      # the text encoder is back-propagated: the first range of the flat gradient is final
            rt.text_ready_hook()
        return None, None, None, None


class OnehotConv5Fn(Function):
    @staticmethod
    def forward(ctx, v, anchor, conv, cache, key, err):
        wt = onehot_weight(cache, key, conv.weight)
        B, L = v.shape
        y = torch.empty(B, L, conv.weight.shape[0], device=v.device, dtype=torch.float32)
        ops.onehot_conv5(v, wt, conv.bias, y, err_flag=err)
        ctx.save_for_backward(v)
        ctx.conv = conv
        return y

    @staticmethod
    def backward(ctx, dy):
        (v,) = ctx.saved_tensors
        ops.onehot_conv5_bwd(v, dy, G(ctx.conv.weight), G(ctx.conv.bias))
        return None, None, None, None, None, None


class MelCalibrateFn(Function):
    @staticmethod
    def forward(ctx, x, mel_len, src_len, S):
        ctx.save_for_backward(mel_len, src_len)
        ctx.T, ctx.x16 = x.shape[1], x.dtype == torch.bfloat16
        return ops.mel_calibrate(x, mel_len, src_len, S)

    @staticmethod
    def backward(ctx, dy):
        mel_len, src_len = ctx.saved_tensors                 # (a bf16 input gets its gradient in bf16)
        return ops.mel_calibrate_bwd(dy, mel_len, src_len, ctx.T, out_bf16=ctx.x16), None, None, None


class LstmLayerFn(Function):
    """One bidirectional nn.LSTM layer: input GEMM (both directions) + recurrent kernel."""

    @staticmethod
    def forward(ctx, x, anchor, lstm, layer, H, enc, key):
        w, bias, w_hh, prec = enc._lstm_weights(lstm, layer, key, x.shape[-1])
        gx = ops.conv_gemm(x, w, bias, n=8 * H, prec=prec)
        B, S, _ = gx.shape
        gates = torch.empty(B, S, 8 * H, device=x.device, dtype=torch.float32)
        cell = torch.empty(B, S, 2 * H, device=x.device, dtype=torch.float32)
        out = ops.lstm_bidir(gx, w_hh, H, cell_out=cell, gates_out=gates)
        ctx.save_for_backward(x, out, gates, cell, w_hh)
        ctx.lstm, ctx.layer, ctx.H, ctx.enc, ctx.key = lstm, layer, H, enc, key
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, gates, cell, w_hh = ctx.saved_tensors
        lstm, layer, H = ctx.lstm, ctx.layer, ctx.H
        cin = x.shape[-1]
        dgp = ops.lstm_bidir_bwd(dout, gates, cell, w_hh, H)
        for d, sfx in enumerate(("", "_reverse")):
            sl = dgp[..., d * 4 * H:(d + 1) * 4 * H]
            # both biases receive colsum(dgates): fused into the input-weight gradient launch
            ops.wgrad(sl, x, G(getattr(lstm, f"weight_ih_l{layer}{sfx}")), 4 * H, cin,
                      db=G(getattr(lstm, f"bias_ih_l{layer}{sfx}")), db2=G(getattr(lstm, f"bias_hh_l{layer}{sfx}")))
            # h_{prev}: forward direction reads out[t-1], reverse direction out[t+1]
            ops.wgrad(sl, out[..., d * H:(d + 1) * H], G(getattr(lstm, f"weight_hh_l{layer}{sfx}")), 4 * H, H,
                      pad_left=1 if d == 0 else -1)
        dx = None
        if ctx.needs_input_grad[0]:
            names = [f"weight_ih_l{layer}", f"weight_ih_l{layer}_reverse"]
            srcs = [getattr(lstm, n) for n in names]
            wt, wp = lstm_wi_transposed(ctx.enc._derived, ctx.key + "wiT", srcs[0], srcs[1])
            dx = ops.conv_gemm(dgp, wt, None, n=cin, prec=wp)
        return dx, None, None, None, None, None, None


class LstmMultiLayerFn(Function):
    """One bidirectional layer of ALL four AudioEncoder LSTMs: four input GEMMs, one recurrent launch; backward: one
    BPTT launch, then the weight / input gradients as GEMMs."""

    @staticmethod
    def forward(ctx, enc, layer, anchor, *xs):
        Hs = enc.necks
        calls, w_hhs = [], []
        for s, x in enumerate(xs):
            lstm = getattr(enc, f"lstm_{s + 1}")
            w, bias, w_hh, prec = enc._lstm_weights(lstm, layer, f"lstm{s}_{layer}", x.shape[-1])
            calls.append(dict(x=x, w=w, bias=bias, n=8 * Hs[s], prec=prec))
            w_hhs.append(w_hh)
        gxs = ops.conv_gemm_multi(calls)             # the four input projections: one grouped launch
        parts = rt.lstm_parts()
        outs, cells, gates = ops.lstm_bidir_multi(gxs, w_hhs, Hs, save=True, parts=parts)
        ctx.save_for_backward(*xs, *outs, *cells, *gates, *w_hhs)
        ctx.enc, ctx.layer, ctx.parts = enc, layer, parts
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        enc, layer = ctx.enc, ctx.layer
        Hs = enc.necks
        t = ctx.saved_tensors
        xs, outs, cells, gates, w_hhs = t[0:4], t[4:8], t[8:12], t[12:16], t[16:20]
        dgps = ops.lstm_bidir_bwd_multi(douts, gates, cells, w_hhs, Hs, parts=ctx.parts)
        dxs, dx_calls, dx_slots = [], [], []
        for s in range(4):
            lstm = getattr(enc, f"lstm_{s + 1}")
            H, x, dgp, cin = Hs[s], xs[s], dgps[s], xs[s].shape[-1]
            for d, sfx in enumerate(("", "_reverse")):
                sl = dgp[..., d * 4 * H:(d + 1) * 4 * H]
                ops.wgrad(sl, x, G(getattr(lstm, f"weight_ih_l{layer}{sfx}")), 4 * H, cin,
                          db=G(getattr(lstm, f"bias_ih_l{layer}{sfx}")), db2=G(getattr(lstm, f"bias_hh_l{layer}{sfx}")))
                ops.wgrad(sl, outs[s][..., d * H:(d + 1) * H], G(getattr(lstm, f"weight_hh_l{layer}{sfx}")), 4 * H, H,
                          pad_left=1 if d == 0 else -1)
            dxs.append(None)
            if ctx.needs_input_grad[3 + s]:
                srcs = [getattr(lstm, f"weight_ih_l{layer}"), getattr(lstm, f"weight_ih_l{layer}_reverse")]
                wt, wp = lstm_wi_transposed(enc._derived, f"lstm{s}_{layer}wiT", srcs[0], srcs[1])
                dx_calls.append(dict(x=dgp, w=wt, n=cin, prec=wp))
                dx_slots.append(s)
        for s, dx in zip(dx_slots, ops.conv_gemm_multi(dx_calls) if dx_calls else []):       # one grouped launch
            dxs[s] = dx
        return (None, None, None, *dxs)


class AugTailFn(Function):
    @staticmethod
    def forward(ctx, h, anchor, c):
        ctx.save_for_backward(h)
        ctx.c = c
        return ops.aug_classifier_tail(h, c.d_bn1.weight, c.d_bn1.bias, c.d_fc2.weight, c.d_fc2.bias)

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        c = ctx.c
        dh = ops.aug_classifier_tail_bwd(h, c.d_bn1.weight, c.d_bn1.bias, c.d_fc2.weight, c.d_fc2.bias, dout,
                                         G(c.d_bn1.weight), G(c.d_bn1.bias), G(c.d_fc2.weight), G(c.d_fc2.bias))
        return dh, None, None


class LengthRegulateFn(Function):
    @staticmethod
    def forward(ctx, x, csum, T):
        ctx.save_for_backward(csum)
        ctx.S = x.shape[1]
        out = ops.length_regulate(x, csum, T)
        return _r16(out) if (rt.sim_bf16_stream and rt.prec == ops.PREC_BF16) else out

    @staticmethod
    def backward(ctx, dy):
        (csum,) = ctx.saved_tensors
        return ops.length_regulate_bwd(dy, csum, ctx.S), None, None


class BucketEmbedAddFn(Function):
    """out = text + pitch_emb[..] + speaker + energy_emb[..]; out2 = out.detach() + noise (styler.py:55)."""

    @staticmethod
    def forward(ctx, text, speaker, noise, anchor, p_src, p_scale, e_src, e_scale, sm):
        B, T, _ = text.shape
        pid = torch.empty(B, T, device=text.device, dtype=torch.int32)
        eid = torch.empty_like(pid)
        out, out2 = ops.bucket_embed_add(text, speaker, p_src, p_scale, e_src, e_scale, sm.pitch_bins, sm.energy_bins,
                                         sm.pitch_embedding.weight, sm.energy_embedding.weight, noise=noise, p_ids=pid,
                                         e_ids=eid)
        ctx.save_for_backward(pid, eid)
        ctx.sm = sm
        if out2 is None:
            out2 = out.new_zeros(1)
        return out, out2

    @staticmethod
    def backward(ctx, dout, dout2):
        pid, eid = ctx.saved_tensors
        sm = ctx.sm
        if rt.grad_ready_hook is not None:      # both decodes are back-propagated: decoder/PostNet grads are final
            rt.grad_ready_hook()
        dout = dout.contiguous()
        ops.bucket_embed_bwd(dout, pid, eid, G(sm.pitch_embedding.weight), G(sm.energy_embedding.weight))
        dnoise = dout2 if ctx.needs_input_grad[2] else None
        return dout, dout, dnoise, None, None, None, None, None, None


class AddPosFn(Function):
    @staticmethod
    def forward(ctx, x, pe):
        return ops.add_pos(x, pe)

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class PackRowsFn(Function):
    """padded [B, T, C] (+ positional table) -> packed [1, B*T, C] (ops.PackPlan); backward = unpack."""

    @staticmethod
    def forward(ctx, x, pe, plan, out_bf16=False):
        ctx.plan = plan
        return ops.pack_rows(x, plan, add=pe, out_bf16=out_bf16)

    @staticmethod
    def backward(ctx, dy):
        return ops.unpack_rows(dy, ctx.plan), None, None, None


class SplitChannelsFn(Function):
    """x [B, T, k*H] -> its k channel slices (views).  Native slicing makes autograd zero-fill k full-size tensors and add
    them pairwise -- for the LengthRegulator output [B, T, 1280] that is 5 fills + 4 adds of 108 MB per step; here
    backward gathers the k slice gradients into ONE buffer with k strided copies (rt.fused_split)."""

    @staticmethod
    def forward(ctx, x, H):
        ctx.meta = (x.shape, H, x.device)
        return tuple(x[..., i * H:(i + 1) * H] for i in range(x.shape[-1] // H))

    @staticmethod
    def backward(ctx, *gs):
        shape, H, dev = ctx.meta
        if all(g is None for g in gs):
            return None, None
        out = torch.empty(shape, device=dev, dtype=torch.float32)
        ops.copy_rows_multi([(None if g is None else ops._rows_view(g), out[..., i * H:(i + 1) * H])
                             for i, g in enumerate(gs)])
        return out, None


class SplitWidthsFn(Function):
    """x [B, L, sum(widths)] -> channel slices of the given widths (views); backward gathers the slice gradients into one
    buffer (SplitChannelsFn for unequal widths: the four streams of the mel calibrator's output)."""

    @staticmethod
    def forward(ctx, x, widths):
        ctx.meta = (x.shape, widths, x.device)
        outs, off = [], 0
        for w in widths:
            outs.append(x[..., off:off + w])
            off += w
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        shape, widths, dev = ctx.meta
        if all(g is None for g in gs):
            return None, None
        out = torch.empty(shape, device=dev, dtype=torch.float32)
        pairs, off = [], 0
        for g, w in zip(gs, widths):
            pairs.append((None if g is None else ops._rows_view(g), out[..., off:off + w]))
            off += w
        ops.copy_rows_multi(pairs)
        return out, None


class SplitBatchFn(Function):
    """x [2B, ...] -> (x[:B], x[B:]) (views); backward = one concatenation (native slicing: two zero-filled full-size
    tensors, two copies and an add)."""

    @staticmethod
    def forward(ctx, x):
        ctx.meta = (x.shape, x.device)
        ctx.set_materialize_grads(False)               # an unused half arrives as None (handled below), not as a zero-filled tensor
        B = x.shape[0] // 2
        return x[:B], x[B:]

    @staticmethod
    def backward(ctx, ga, gb):
        shape, dev = ctx.meta
        if ga is None and gb is None:
            return None
        B = shape[0] // 2
        out = torch.empty(shape, device=dev, dtype=torch.float32)
        # one kernel launch (two aten copy_ calls were two memcpy NODES of the captured graph, each behind 6-8 us of idle time)
        # each half is one contiguous run: narrow tensors (the classifiers' [2B, 2] log-probabilities) go as one long row
        wide = shape[-1] % 4 == 0
        half = out[:B].numel()
        flat = (lambda t: t.reshape(-1, t.shape[-1])) if wide else (lambda t: t.reshape(1, -1))
        ok = wide or (half % 4 == 0)
        pairs = []
        for dst, g in ((out[:B], ga), (out[B:], gb)):
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() % 16 or not ok):
                dst.copy_(g)                                  # (not the step's case: any layout / dtype)
            elif g is None and not ok:
                dst.zero_()
            else:
                pairs.append((None if g is None else flat(g), flat(dst)))
        if pairs:
            ops.copy_rows_multi(pairs)
        return out


class StackedFanoutFn(Function):
    """x [2B, ...] -> (x, x[:B]) for a stacked tensor with exactly these two consumers (the AudioEncoder's main + DAT encodings:
    the classifiers take all 2B items, the style path the first B).  Backward = ONE launch: the first consumer's gradient with the
    second one's added onto its first half, in place (that tensor is the classifier's fresh dX output and has no other reader) --
    instead of autograd's zero-extended copy of the half + an aten add (round 6)."""

    @staticmethod
    def forward(ctx, x):
        ctx.meta = (x.shape, x.device)
        ctx.set_materialize_grads(False)
        return x.view_as(x), x[: x.shape[0] // 2]

    @staticmethod
    def backward(ctx, g_all, g_main):
        shape, dev = ctx.meta
        B = shape[0] // 2
        if g_all is None and g_main is None:
            return None
        if g_all is None:                               # only the half is used: zero-extended copy (as SplitBatchFn)
            out = torch.empty(shape, device=dev, dtype=torch.float32)
            flat = lambda t: t.reshape(-1, t.shape[-1])
            ops.copy_rows_multi([(flat(g_main.contiguous().float()), flat(out[:B])), (None, flat(out[B:]))])
            return out
        if g_main is None:
            return g_all
        ok = (g_all.dtype == torch.float32 and g_all.is_contiguous() and g_main.dtype == torch.float32 and g_main.is_contiguous()
              and g_all.shape[-1] % 4 == 0)
        if not ok:
            out = g_all.clone()
            out[:B] += g_main
            return out
        ops.add2(g_all[:B], g_main, out=g_all[:B])
        return g_all


class PackPairFn(Function):
    """Two padded [B, T, C] tensors (+ positional table) -> one packed batch of 2B items; backward = unpack per half."""

    @staticmethod
    def forward(ctx, xa, xb, pe, plan, out_bf16=False):
        ctx.plan = plan
        out = ops.pack_rows_pair(xa, xb, plan, add=pe, out_bf16=out_bf16)
        return _r16(out) if (rt.sim_bf16_stream and rt.prec == ops.PREC_BF16 and not out_bf16) else out

    @staticmethod
    def backward(ctx, dy):
        da, db = ops.unpack_rows_pair(dy, ctx.plan)
        return da, db, None, None, None


class UnpackRowsFn(Function):
    """packed -> padded with zero rows at t >= len[b]; backward = pack."""

    @staticmethod
    def forward(ctx, xp, plan):
        ctx.plan, ctx.in16 = plan, xp.dtype == torch.bfloat16
        return ops.unpack_rows(xp, plan)

    @staticmethod
    def backward(ctx, dy):
        return ops.pack_rows(ops._rows_view(dy), ctx.plan, out_bf16=ctx.in16), None


class Add2Fn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add2(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class AddRowvecFn(Function):
    """out[b,t,:] = a[b,t,:] + v[b,:]  (a may be None: pure broadcast)."""

    @staticmethod
    def forward(ctx, a, v, L):
        ctx.has_a = a is not None
        return ops.add_rowvec(a, v, L)

    @staticmethod
    def backward(ctx, dy):
        return (dy if ctx.has_a else None), ops.rowsum(dy), None


class CatFn(Function):
    """Channel concatenation into one [B, L, sum C] buffer (strided copies; backward = slice views)."""

    @staticmethod
    def forward(ctx, *parts):
        B, L = parts[0].shape[:2]
        widths = [p.shape[-1] for p in parts]
        out = torch.empty(B, L, sum(widths), device=parts[0].device, dtype=torch.float32)
        pairs, off = [], 0
        for p, w in zip(parts, widths):
            pairs.append((ops._rows_view(p), out[..., off:off + w]))
            off += w
        ops.copy_rows_multi(pairs)
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, dy):
        outs, off = [], 0
        for w in ctx.widths:
            outs.append(dy[..., off:off + w])
            off += w
        return tuple(outs)


class StyleCatFn(Function):
    """Round 6: the decoder-input concatenation of StyleModeling.forward (modules.py:335-350) and the duration predictor's input as
    ONE tape node / one launch: enc = [text | pitch_up + neck_up | speaker | neck_up + energy_up | residual_up], dp = neck_up +
    duration_up (was three Add2Fn, two AddRowvecFn and the CatFn copy).  Backward: slice views of d_enc, the speaker's row sum, and
    ONE launch for the neck's three contributions (autograd added them with two aten adds)."""

    @staticmethod
    def forward(ctx, te, pu, tnu, spk, eu, ru, du):
        ctx.set_materialize_grads(False)
        ctx.shape = te.shape
        return ops.style_cat(*(t.contiguous() for t in (te, pu, tnu, spk, eu, ru, du)))

    @staticmethod
    def backward(ctx, d_enc, d_dp):
        B, S, _ = ctx.shape
        if d_enc is None and d_dp is None:
            return (None,) * 7
        if d_enc is None:
            return None, None, d_dp, None, None, None, d_dp
        part = lambda i: d_enc[..., 256 * i:256 * (i + 1)]
        if d_dp is None:
            d_tnu = ops.add2(ops._rows_view(part(1)), ops._rows_view(part(3)))
        else:
            d_tnu = ops.add3(part(1), part(3), d_dp)
        return part(0), part(1), d_tnu, ops.rowsum(part(2)), part(3), part(4), d_dp


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy, ctx.p, ctx.seed), None, None


class MaskedErrFn(Function):
    """mean over valid positions of (a-b)^2 (kind 0) or |a-b| (kind 1): masked_select + MSELoss/L1Loss.  The mean is
    taken inside the kernel (no aten div / cast launches)."""

    @staticmethod
    def forward(ctx, a, b, kind, lens):
        a, b = a.contiguous(), b.contiguous()
        mean, acc = ops.masked_err_mean(a, b, kind, lens)
        ctx.save_for_backward(a, b, acc, lens)
        ctx.kind = kind
        return mean.view(())

    @staticmethod
    def backward(ctx, g):
        a, b, acc, lens = ctx.saved_tensors
        return ops.masked_err_bwd(a, b, acc, g.reshape(1), ctx.kind, lens), None, None, None


class MaskedErrMultiFn(Function):
    """Several masked MSE / L1 means (loss.py:16-50) as ONE tape node: one launch forward, one backward.
    apply(kinds, lens_list, a0, b0, a1, b1, ...) -> tuple of scalar means."""

    @staticmethod
    def forward(ctx, kinds, lens_list, *ab):
        a_s = [t.contiguous() for t in ab[0::2]]
        b_s = [t.contiguous() for t in ab[1::2]]
        means, accs = ops.masked_err_mean_multi([(a, b, k, l) for a, b, k, l in zip(a_s, b_s, kinds, lens_list)])
        ctx.save_for_backward(*a_s, *b_s, *accs, *[l for l in lens_list if l is not None])
        ctx.kinds, ctx.has_len, ctx.n = kinds, [l is not None for l in lens_list], len(a_s)
        return tuple(m.view(()) for m in means)

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        t = ctx.saved_tensors
        a_s, b_s, accs, rest = t[0:n], t[n:2 * n], t[2 * n:3 * n], list(t[3 * n:])
        lens = [rest.pop(0) if h else None for h in ctx.has_len]
        live = [i for i in range(n) if gs[i] is not None and ctx.needs_input_grad[2 + 2 * i]]
        das = ops.masked_err_bwd_multi([(a_s[i], b_s[i], accs[i], gs[i].reshape(1), ctx.kinds[i], lens[i]) for i in live]) if live else []
        out = [None, None]
        it = iter(das)
        for i in range(n):
            out += [next(it) if i in live else None, None]
        return tuple(out)


class Nll3Fn(Function):
    """3 x NLLLoss(mean) on [B, 2] log-probabilities, summed (loss.py:46-48 / 60-68): one launch each way."""

    @staticmethod
    def forward(ctx, p0, p1, p2, label):
        lps = [p.contiguous() for p in (p0, p1, p2)]
        if isinstance(label, int):                   # python int label: nothing but tensors goes through save_for_backward
            ctx.save_for_backward(*lps)
            ctx.label = label
        else:
            ctx.save_for_backward(*lps, label)
            ctx.label = None
        return ops.nll3(lps, label).view(())

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        lps, label = list(saved[:3]), (ctx.label if ctx.label is not None else saved[3])
        d3 = ops.nll3(lps, label, gscale=g.reshape(1), want_grad=True)
        return d3[0], d3[1], d3[2], None


class WeightedSumFn(Function):
    """total = sum_i w_i * term_i over scalar tensors (train.py:156-160) in one launch; backward = one launch writing
    g * w_i for every term (instead of a chain of single-element aten add / mul kernels both ways)."""

    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.weights = weights
        return ops.weighted_sum([t.reshape(1) for t in terms], weights).view(())

    @staticmethod
    def backward(ctx, g):
        gw = ops.scale_weights(g.reshape(1), ctx.weights)
        return (None, *[gw[i] for i in range(len(ctx.weights))])


class LossTailFn(Function):
    """Round 6: the two classifier NLL3 terms (main pass, DAT pass) and the weighted total of train.py:156-160 as ONE tape node:
    one launch forward (was nll3 x 2 + weighted_sum), one backward (was scale_weights + nll3 x 2).
    apply(weights [n + 2], label0, label1, mean_0 .. mean_{n-1}, lp_0 .. lp_5) -> (total, cls, cls_dat).  Gradients flow from
    `total` only (cls / cls_dat are returned for logging, as the reference logs them)."""

    @staticmethod
    def forward(ctx, weights, label0, label1, *t):
        n = len(weights) - 2
        means, lps = t[:n], [p.contiguous() for p in t[n:]]
        tens = [l for l in (label0, label1) if not isinstance(l, int)]
        ctx.save_for_backward(*lps, *tens)
        ctx.weights, ctx.n = weights, n
        ctx.labels = tuple(l if isinstance(l, int) else None for l in (label0, label1))
        out = ops.loss_tail([m.reshape(1) for m in means], weights, lps, (label0, label1))
        total, cls, dat = out[0:1].view(()), out[1:2].view(()), out[2:3].view(())
        ctx.mark_non_differentiable(cls, dat)
        ctx.set_materialize_grads(False)               # (no zero fills for the two logging outputs' gradient slots)
        return total, cls, dat

    @staticmethod
    def backward(ctx, g, g_cls, g_dat):
        if g is None:                                   # (the total is not part of the differentiated graph)
            return (None,) * (3 + ctx.n + 6)
        saved = list(ctx.saved_tensors)
        lps, rest = saved[:6], saved[6:]
        labels = tuple(l if l is not None else rest.pop(0) for l in ctx.labels)
        gw, d6 = ops.loss_tail_bwd(g.reshape(1), ctx.weights, lps, labels)
        return (None, None, None, *[gw[i] for i in range(ctx.n)], *[d6[k] for k in range(6)])


class NllFn(Function):
    @staticmethod
    def forward(ctx, logp, label):
        logp = logp.contiguous()
        ctx.save_for_backward(logp, label)
        return ops.nll(logp, label)

    @staticmethod
    def backward(ctx, g):
        logp, label = ctx.saved_tensors
        return ops.nll(logp, label, gscale=g.reshape(1).float().contiguous(), want_grad=True), None


def next_dropout_seed():
    rt.dropout_calls += 1
    return (rt.seed * 1000003 + rt.dropout_calls) & 0x7FFFFFFFFFFFFFFF


def dropout(x, p, training):
    if not training or p <= 0.0 or rt.disable_dropout:
        return x
    seed = next_dropout_seed()
    if torch.is_grad_enabled() and x.requires_grad:
        return DropoutFn.apply(x, p, seed)
    return ops.dropout(x, p, seed)
