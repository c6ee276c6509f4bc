"""GPU parity of the backward / training path: HIP backward kernels (through the autograd Functions of
styler_amd.autograd) vs torch autograd on CPU (fp64 where cheap), the full train step vs the
reference-generated golden fixture (10 loss scalars, global grad norm, sampled gradients) and vs the oracle."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def relerr(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def check(a, b, tol, what):
    e = relerr(a, b)
    assert e <= tol, f"{what}: max err / max|ref| = {e:.3e} > {tol}"


class _Holder(nn.Module):
    pass


@pytest.mark.parametrize("B,L,cin,n,kw,act", [(3, 37, 256, 256, 1, 0), (2, 50, 256, 1024, 9, 1), (2, 41, 80, 512, 5, 2),
                                               (2, 33, 1024, 256, 1, 0), (4, 61, 256, 256, 3, 1), (5, 1, 512, 128, 1, 1)])
def test_conv_gemm_backward(dev, B, L, cin, n, kw, act):
    from styler_amd import autograd as AG
    from styler_amd.runtime import Derived
    g = torch.Generator().manual_seed(kw * 100 + n)
    conv = nn.Conv1d(cin, n, kw, padding=kw // 2).double()
    x = torch.randn(B, L, cin, generator=g, dtype=torch.float64, requires_grad=True)
    res = torch.randn(B, L, n, generator=g, dtype=torch.float64, requires_grad=True)
    z = conv(x.transpose(1, 2)).transpose(1, 2)
    y = (torch.relu(z) if act == 1 else torch.tanh(z) if act == 2 else z) + res
    gy = torch.randn(B, L, n, generator=g, dtype=torch.float64)
    y.backward(gy)

    holder = nn.Conv1d(cin, n, kw, padding=kw // 2).to(dev)
    with torch.no_grad():
        holder.weight.copy_(conv.weight.float()); holder.bias.copy_(conv.bias.float())
    xd = x.detach().float().to(dev).requires_grad_(True)
    rd = res.detach().float().to(dev).requires_grad_(True)
    w_arg = holder.weight if kw > 1 else holder.weight
    lin = holder
    if kw == 1:                                   # nn.Linear-shaped parameter
        lin = nn.Linear(cin, n).to(dev)
        with torch.no_grad():
            lin.weight.copy_(conv.weight.float()[:, :, 0]); lin.bias.copy_(conv.bias.float())
    yd = AG.ConvGemmFn.apply(xd, rd, lin.weight, lin.bias, Derived(), "t", kw, act, False)
    check(yd, y, 1e-5, "fwd")
    yd.backward(gy.float().to(dev))
    check(xd.grad, x.grad, 2e-5, "dx")
    check(rd.grad, res.grad, 1e-6, "dres")
    gw = conv.weight.grad if kw > 1 else conv.weight.grad[:, :, 0]
    check(lin.weight.grad, gw, 2e-5, "dw")
    check(lin.bias.grad, conv.bias.grad, 2e-5, "db")


@pytest.mark.parametrize("B,L,cin,n,kw,pad", [(3, 137, 256, 256, 1, 0), (2, 150, 256, 1024, 9, 4), (2, 141, 80, 512, 5, 2),
                                               (4, 61, 256, 256, 3, 1), (5, 1, 512, 128, 1, 0), (3, 70, 64, 256, 1, 1),
                                               (3, 70, 64, 256, 1, -1), (2, 66, 260, 320, 5, 2), (3, 90, 256, 64, 1, 0),
                                               (2, 80, 48, 64, 3, 1), (2, 100, 320, 64, 5, 2), (2, 130, 192, 320, 1, 0)])
def test_wgrad_bf16(dev, B, L, cin, n, kw, pad):
    """bf16-operand weight gradient (sliding-window gather kernel) vs fp64; also the shifted-Linear form used for
    the LSTM recurrent weights (pad +1 / -1) and the one-hot conv form (cin = 257 padded to 260)."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(B * 7 + kw)
    dz = torch.randn(B, L, n, generator=g, dtype=torch.float64)
    x = torch.randn(B, L, cin, generator=g, dtype=torch.float64)
    cin_eff = 257 if cin == 260 else cin
    ref = torch.zeros(n, cin_eff, kw, dtype=torch.float64)
    for j in range(kw):
        sh = j - pad
        xs = torch.zeros_like(x)
        if sh >= 0:
            xs[:, :L - sh] = x[:, sh:]
        else:
            xs[:, -sh:] = x[:, :L + sh]
        ref[:, :, j] = torch.einsum("btn,btc->nc", dz, xs)[:, :cin_eff]
    for prec, tol in ((ops.PREC_BF16, 2e-2), (ops.PREC_F32, 2e-5)):
        dw = torch.zeros(n, cin_eff, kw, device=dev)
        db = torch.zeros(n, device=dev)
        ops.wgrad(dz.float().to(dev), x.float().to(dev), dw, n, cin_eff, kw=kw, db=db, pad_left=pad,
                  strides=(cin_eff * kw, kw, 1), prec=prec)
        check(dw, ref, tol, f"dw prec={prec}")
        check(db, dz.sum((0, 1)), 1e-4, "db")
    # bf16x3 arithmetic on fp32-typed operands: three calls of the bf16 engine, the low parts staged by the kernel itself
    # (STYLER_IO_X_LO / STYLER_IO_DZ_LO) -- fp32-class, and bit-equal to the same calls on materialised low parts
    dzd, xd = dz.float().to(dev), x.float().to(dev)
    kw_args = dict(kw=kw, pad_left=pad, strides=(cin_eff * kw, kw, 1), prec=ops.PREC_BF16)
    dw3 = torch.zeros(n, cin_eff, kw, device=dev)
    db3 = torch.zeros(n, device=dev)
    ops.wgrad(dzd, xd, dw3, n, cin_eff, kw=kw, db=db3, pad_left=pad, strides=(cin_eff * kw, kw, 1), prec=ops.PREC_BF16X3)
    check(dw3, ref, 1e-4 * max(1.0, float(ref.abs().max())), "dw prec=bf16x3")
    check(db3, dz.sum((0, 1)), 1e-4, "db bf16x3")
    # (shapes with an LDS-DMA kernel run the three parts as ONE launch, STYLER_IO_X3CAT: against three launches)
    prev_cat, ops.x3cat = ops.x3cat, False
    try:
        dw3b, db3b = torch.zeros(n, cin_eff, kw, device=dev), torch.zeros(n, device=dev)
        ops.wgrad(dzd, xd, dw3b, n, cin_eff, kw=kw, db=db3b, pad_left=pad, strides=(cin_eff * kw, kw, 1), prec=ops.PREC_BF16X3)
    finally:
        ops.x3cat = prev_cat
    check(dw3, dw3b, 2e-6 * max(1.0, float(ref.abs().max())), "dw bf16x3: one launch vs three")
    check(db3, db3b, 1e-5 * max(1.0, float(dz.sum((0, 1)).abs().max())), "db bf16x3: one launch vs three")
    if n % 4 == 0:
        a, b_ = torch.zeros_like(dw3), torch.zeros_like(dw3)
        ops.wgrad(dzd, xd, a, n, cin_eff, parts=ops.IO_X_LO, **kw_args)
        ops.wgrad(dzd, xd, a, n, cin_eff, parts=ops.IO_DZ_LO, **kw_args)
        ops.wgrad(dzd, ops.lo_part(xd), b_, n, cin_eff, **kw_args)
        ops.wgrad(ops.lo_part(dzd), xd, b_, n, cin_eff, **kw_args)
        assert torch.equal(a, b_), f"low-part flags differ from materialised low parts by {float((a - b_).abs().max()):.3e}"


@pytest.mark.parametrize("B,L,lens", [(2, 24, [24, 17]), (2, 150, [150, 77]), (1, 200, [131])])
def test_attention_backward(dev, B, L, lens):
    from styler_amd import ops
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, L, 768, generator=g, dtype=torch.float64, requires_grad=True)
    ln = torch.tensor(lens)
    q, k, v = [t.view(B, L, 4, 64).permute(0, 2, 1, 3) for t in qkv.split(256, dim=-1)]
    s = (q @ k.transpose(-1, -2)) / 8.0
    s = s.masked_fill((torch.arange(L)[None, :] >= ln[:, None])[:, None, None, :], float("-inf"))
    out = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, L, 256)
    # gradient only on valid query rows, as in the model (padded rows are zeroed after the LayerNorm)
    gy = torch.randn(B, L, 256, generator=g, dtype=torch.float64) * (torch.arange(L)[None, :, None] < ln[:, None, None])
    out.backward(gy)
    qd = qkv.detach().float().to(dev)
    lse = torch.empty(B, 4, L, device=dev)
    od = ops.attention_fwd(qd, ln.to(dev), lse=lse, prec=ops.PREC_F32)
    dq = ops.attention_bwd(qd, od, gy.float().to(dev), lse, ln.to(dev), prec=ops.PREC_F32)
    check(dq, qkv.grad, 3e-5, "dqkv")
    od16 = ops.attention_fwd(qd, ln.to(dev), lse=lse, prec=ops.PREC_BF16)
    dq16 = ops.attention_bwd(qd, od16, gy.float().to(dev), lse, ln.to(dev), prec=ops.PREC_BF16)
    check(dq16, qkv.grad, 3e-2, "dqkv bf16")
    # bf16x3 arithmetic: fp32-class (rows past an item's length are don't-care, as in the bf16 kernels)
    od3 = ops.attention_fwd(qd, ln.to(dev), lse=lse, prec=ops.PREC_BF16X3)
    dq3 = ops.attention_bwd(qd, od3, gy.float().to(dev), lse, ln.to(dev), prec=ops.PREC_BF16X3)
    vrow = (torch.arange(L)[None, :, None] < ln[:, None, None])
    check(dq3 * vrow.to(dev), qkv.grad * vrow, 2e-4, "dqkv bf16x3")
    lse = torch.empty(B, 4, L, device=dev)
    od16 = ops.attention_fwd(qd, ln.to(dev), lse=lse, prec=ops.PREC_BF16)
    # bf16 storage of dqkv: the fp32 result rounded (valid rows; rows past an item's length are don't-care in both)
    dq16s = ops.attention_bwd(qd, od16, gy.float().to(dev), lse, ln.to(dev), prec=ops.PREC_BF16, out_bf16=True)
    assert dq16s.dtype == torch.bfloat16
    valid = (torch.arange(L)[None, :] < ln[:, None]).to(dev)
    assert torch.equal(dq16s[valid], dq16.to(torch.bfloat16)[valid])
    # qkv STORED as bf16 (round 3, rt.bf16_qkv): k and v are the same bf16 values the fp32-input kernels round to; the
    # 1/sqrt(d_k) scale moves from q into the exponent, so q is rounded once too -- against the fp64 math of the ROUNDED
    # qkv the stored-bf16 kernels must be as close as the fp32-input kernels are (measured: equal to within 1.3x)
    q16 = qd.to(torch.bfloat16)
    qr = q16.double().cpu().requires_grad_(True)
    q_, k_, v_ = [t.view(B, L, 4, 64).permute(0, 2, 1, 3) for t in qr.split(256, dim=-1)]
    s_ = ((q_ @ k_.transpose(-1, -2)) / 8.0).masked_fill((torch.arange(L)[None, :] >= ln[:, None])[:, None, None, :], float("-inf"))
    out_r = (torch.softmax(s_, -1) @ v_).permute(0, 2, 1, 3).reshape(B, L, 256)
    out_r.backward(gy)
    lse_a, lse_b = torch.empty(B, 4, L, device=dev), torch.empty(B, 4, L, device=dev)
    o_a = ops.attention_fwd(q16, ln.to(dev), lse=lse_a, prec=ops.PREC_BF16)
    o_b = ops.attention_fwd(q16.float(), ln.to(dev), lse=lse_b, prec=ops.PREC_BF16)
    vq = valid[..., None].cpu()

    def err(t, ref, mask):
        return float(((t.double().cpu() - ref) * mask).abs().max()) / float((ref * mask).abs().max())
    ea, eb = err(o_a, out_r.detach(), vq), err(o_b, out_r.detach(), vq)
    assert ea <= 2e-2 and ea <= 1.5 * eb + 1e-3, (ea, eb)
    assert float(((lse_a - lse_b).cpu() * valid[:, None, :].cpu()).abs().max()) <= 2e-2
    d_a = ops.attention_bwd(q16, o_a, gy.float().to(dev), lse_a, ln.to(dev), prec=ops.PREC_BF16, out_bf16=True)
    d_b = ops.attention_bwd(q16.float(), o_b, gy.float().to(dev), lse_b, ln.to(dev), prec=ops.PREC_BF16, out_bf16=True)
    assert d_a.dtype == torch.bfloat16 and d_a.shape == d_b.shape
    ea, eb = err(d_a, qr.grad, vq), err(d_b, qr.grad, vq)
    assert ea <= 3e-2 and ea <= 1.5 * eb + 1e-3, (ea, eb)
    # the forward's output and the incoming gradient stored as bf16 too (rt.bf16_att): the output is the RNE of the fp32 one;
    # the backward's MFMA operands are unchanged, only delta = rowsum(dO * O) sees rounded values
    o_c = ops.attention_fwd(q16, ln.to(dev), lse=lse_a, prec=ops.PREC_BF16, out_bf16=True)
    assert o_c.dtype == torch.bfloat16 and torch.equal(o_c[valid], o_a.to(torch.bfloat16)[valid])
    gy16 = gy.float().to(dev).to(torch.bfloat16)
    d_c = ops.attention_bwd(q16, o_c, gy16, lse_a, ln.to(dev), prec=ops.PREC_BF16, out_bf16=True)
    d_d = ops.attention_bwd(q16, o_a, gy16.float(), lse_a, ln.to(dev), prec=ops.PREC_BF16, out_bf16=True)
    ec, ed = err(d_c, qr.grad, vq), err(d_d, qr.grad, vq)
    assert ec <= 3e-2 and ec <= 1.5 * ed + 1e-3, (ec, ed)


def test_layernorm_backward(dev):
    from styler_amd import autograd as AG
    g = torch.Generator().manual_seed(1)
    B, L = 3, 40
    lens = torch.tensor([40, 9, 25])
    valid = (torch.arange(L)[None, :] < lens[:, None])
    x = torch.randn(B, L, 256, generator=g, dtype=torch.float64, requires_grad=True)
    r = torch.randn(B, L, 256, generator=g, dtype=torch.float64, requires_grad=True)
    ln = nn.LayerNorm(256).double()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(256, generator=g)); ln.bias.copy_(torch.randn(256, generator=g))
    y = ln(x + r) * valid[..., None]
    gy = torch.randn(B, L, 256, generator=g, dtype=torch.float64)
    y.backward(gy)
    lnd = nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        lnd.weight.copy_(ln.weight.float()); lnd.bias.copy_(ln.bias.float())
    xd = x.detach().float().to(dev).requires_grad_(True)
    rd = r.detach().float().to(dev).requires_grad_(True)
    yd = AG.LayerNormFn.apply(xd, rd, lnd.weight, lnd, lens.to(dev))
    yd.backward(gy.float().to(dev))
    check(xd.grad, x.grad, 2e-5, "dx"); check(rd.grad, r.grad, 2e-5, "dres")
    check(lnd.weight.grad, ln.weight.grad, 2e-5, "dgamma"); check(lnd.bias.grad, ln.bias.grad, 2e-5, "dbeta")
    # dot tail
    lin = nn.Linear(256, 1).double()
    x2 = torch.randn(B, L, 256, generator=g, dtype=torch.float64, requires_grad=True)
    ln.zero_grad()
    o = lin(ln(x2)).squeeze(-1) * valid
    go = torch.randn(B, L, generator=g, dtype=torch.float64)
    o.backward(go)
    lind = nn.Linear(256, 1).to(dev)
    with torch.no_grad():
        lind.weight.copy_(lin.weight.float()); lind.bias.copy_(lin.bias.float())
    lnd.zero_grad(set_to_none=True)
    x2d = x2.detach().float().to(dev).requires_grad_(True)
    od = AG.LayerNormDotFn.apply(x2d, lind.weight, lnd, lind, lens.to(dev), 0.0, 0)
    check(od, o, 2e-5, "dot fwd")
    od.backward(go.float().to(dev))
    check(x2d.grad, x2.grad, 3e-5, "dot dx"); check(lind.weight.grad, lin.weight.grad, 3e-5, "dot dw")
    check(lind.bias.grad, lin.bias.grad, 3e-5, "dot db"); check(lnd.weight.grad, ln.weight.grad, 3e-5, "dot dgamma")


@pytest.mark.parametrize("C,L", [(256, 53), (320, 53), (320, 441), (256, 512), (320, 700)])
def test_groupnorm_backward(dev, C, L):
    """L <= 512: single-pass backward; longer items: the statistics + apply pair."""
    from styler_amd import autograd as AG
    g = torch.Generator().manual_seed(C + L)
    x = (torch.randn(2, L, C, generator=g, dtype=torch.float64) * 2 + 0.3).requires_grad_(True)
    gn = nn.GroupNorm(C // 16, C).double()
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g)); gn.bias.copy_(torch.randn(C, generator=g))
    y = torch.relu(gn(x.transpose(1, 2))).transpose(1, 2)
    gy = torch.randn(2, L, C, generator=g, dtype=torch.float64)
    y.backward(gy)
    gnd = nn.GroupNorm(C // 16, C).to(dev)
    with torch.no_grad():
        gnd.weight.copy_(gn.weight.float()); gnd.bias.copy_(gn.bias.float())
    xd = x.detach().float().to(dev).requires_grad_(True)
    yd = AG.GroupNormReluFn.apply(xd, gnd.weight, gnd)
    yd.backward(gy.float().to(dev))
    check(xd.grad, x.grad, 3e-5, "dx"); check(gnd.weight.grad, gn.weight.grad, 3e-5, "dgamma")
    check(gnd.bias.grad, gn.bias.grad, 3e-5, "dbeta")


def test_batchnorm_backward(dev):
    from styler_amd import autograd as AG
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(3, 29, 512, generator=g, dtype=torch.float64) * 1.5).requires_grad_(True)
    bn = nn.BatchNorm1d(512).double()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(512, generator=g)); bn.bias.copy_(torch.randn(512, generator=g))
    y = torch.tanh(bn(x.transpose(1, 2))).transpose(1, 2)
    gy = torch.randn(3, 29, 512, generator=g, dtype=torch.float64)
    y.backward(gy)
    bnd = nn.BatchNorm1d(512).to(dev)
    with torch.no_grad():
        bnd.weight.copy_(bn.weight.float()); bnd.bias.copy_(bn.bias.float())
    xd = x.detach().float().to(dev).requires_grad_(True)
    yd = AG.BatchNormActFn.apply(xd, bnd.weight, bnd, 2)
    yd.backward(gy.float().to(dev))
    check(xd.grad, x.grad, 5e-5, "dx"); check(bnd.weight.grad, bn.weight.grad, 5e-5, "dgamma")
    check(bnd.bias.grad, bn.bias.grad, 5e-5, "dbeta")


@pytest.mark.parametrize("H,cin", [(64, 320), (80, 256)])
def test_lstm_backward(dev, H, cin):
    from styler_amd.modules import AudioEncoder
    g = torch.Generator().manual_seed(H)
    enc = AudioEncoder().to(dev)
    s = 1 if H == 64 else 0
    lstm_d = getattr(enc, f"lstm_{s + 1}")
    ref = nn.LSTM(cin, H, 2, batch_first=True, bidirectional=True).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in lstm_d.state_dict().items()})
    x = torch.randn(3, 17, cin, generator=g, dtype=torch.float64, requires_grad=True)
    y = ref(x)[0]
    gy = torch.randn(3, 17, 2 * H, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd = x.detach().float().to(dev).requires_grad_(True)
    yd = enc._lstm(s, xd)
    check(yd, y, 2e-5, "fwd")
    yd.backward(gy.float().to(dev))
    check(xd.grad, x.grad, 5e-5, "dx")
    for name, p in ref.named_parameters():
        check(getattr(lstm_d, name).grad, p.grad, 5e-5, name)


def test_small_op_backwards(dev):
    from oracle import styler_oracle as O
    from styler_amd import autograd as AG
    from styler_amd import ops
    g = torch.Generator().manual_seed(9)
    # length regulator
    x = torch.randn(2, 6, 1280, generator=g, dtype=torch.float64, requires_grad=True)
    d = torch.tensor([[2, 0, 3, 1, 4, 2], [1, 1, 0, 0, 5, 0]])
    out, _ = O.length_regulate(x, d, 14)
    gy = torch.randn(2, 14, 1280, generator=g, dtype=torch.float64)
    out.backward(gy)
    csum, _, _ = ops.duration_scan(2, 6, dev, dur=d.to(dev))
    xd = x.detach().float().to(dev).requires_grad_(True)
    AG.LengthRegulateFn.apply(xd, csum, 14).backward(gy.float().to(dev))
    check(xd.grad, x.grad, 1e-6, "LR dx")
    # mel calibrator
    ml, sl = torch.tensor([20, 5, 7]), torch.tensor([6, 9, 7])
    m = torch.randn(3, 20, 64, generator=g, dtype=torch.float64, requires_grad=True)
    y = O.mel_calibrate(m, ml, sl)
    gy = torch.randn(3, 9, 64, generator=g, dtype=torch.float64)
    y.backward(gy)
    md = m.detach().float().to(dev).requires_grad_(True)
    AG.MelCalibrateFn.apply(md, ml.to(dev), sl.to(dev), 9).backward(gy.float().to(dev))
    check(md.grad, m.grad, 1e-6, "mel_calibrate dx")
    # aug classifier tail (+ GRL sign)
    from styler_amd.modules import AugmentationClassifier
    clf = AugmentationClassifier(160).to(dev)
    P = {"c." + k: v.detach().double().cpu() for k, v in clf.state_dict().items()}
    xin = torch.randn(3, 11, 160, generator=g, dtype=torch.float64, requires_grad=True)
    ref = O.aug_classifier({k: v.requires_grad_(True) for k, v in P.items()}, "c", xin)
    go = torch.randn(3, 2, generator=g, dtype=torch.float64)
    ref.backward(go)
    xind = xin.detach().float().to(dev).requires_grad_(True)
    got = clf(xind)
    check(got, ref, 2e-5, "aug fwd")
    got.backward(go.float().to(dev))
    check(xind.grad, xin.grad, 5e-5, "aug dx (reversed)")
    for k, v in clf.named_parameters():
        check(v.grad, P["c." + k].grad, 5e-5, "aug " + k)
    # masked losses + nll
    a = torch.randn(4, 9, 80, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, 9, 80, generator=g, dtype=torch.float64)
    lens = torch.tensor([5, 1, 9, 3])
    valid = ~O.length_mask(lens, 9)
    lref = O._masked_mean((a - b) ** 2, valid) * 1.7
    lref.backward()
    ad = a.detach().float().to(dev).requires_grad_(True)
    lgot = AG.MaskedErrFn.apply(ad, b.float().to(dev), 0, lens.to(dev)) * 1.7
    lgot.backward()
    assert abs(float(lgot) - float(lref)) < 1e-5
    check(ad.grad, a.grad, 1e-5, "mse grad")
    lp = torch.log_softmax(torch.randn(5, 2, generator=g, dtype=torch.float64), -1).requires_grad_(True)
    lab = torch.tensor([0, 1, 1, 0, 1])
    F.nll_loss(lp, lab).backward()
    lpd = lp.detach().float().to(dev).requires_grad_(True)
    lg = AG.NllFn.apply(lpd, lab.to(dev))
    lg.backward()
    assert abs(float(lg) - float(F.nll_loss(lp, lab))) < 1e-6
    check(lpd.grad, lp.grad, 1e-6, "nll grad")


def test_dropout_stream(dev):
    from styler_amd import ops
    x = torch.ones(64, 100, 256, device=dev)
    y = ops.dropout(x, 0.2, 1234)
    keep = float((y != 0).float().mean())
    assert abs(keep - 0.8) < 5e-3, keep
    assert torch.equal(y, ops.dropout(x, 0.2, 1234)) and not torch.equal(y, ops.dropout(x, 0.2, 1235))
    assert abs(float(y.max()) - 1.25) < 1e-6


def test_clip_adam_matches_torch(dev):
    from styler_amd import ops
    g = torch.Generator().manual_seed(4)
    n = 100003
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) * s for s in (0.001, 3.0, 0.5)]
    pr = nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ss = torch.zeros(1, dtype=torch.float64, device=dev)
    for step, gr in enumerate(grads, 1):
        pr.grad = gr.clone()
        nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        ss.zero_()
        gd = gr.to(dev)
        ops.sumsq(gd, ss)
        ops.adam_step(p, gd, m, v, ss, 1.0, 1e-3, 0.9, 0.98, 1e-9, step)
        assert relerr(p, pr.data) < 2e-6, (step, relerr(p, pr.data))


# ----------------------------------------------------------------------------- full train step
def _batch(g, dev):
    return {k[3:]: T(g[k]).to(dev) for k in g.files if k.startswith("in_")}


@pytest.fixture(scope="module")
def train_model(dev, ref_state_dict):
    from styler_amd import STYLER, rt
    m = STYLER()
    m.load_state_dict(ref_state_dict)
    m = m.to(dev).train()
    rt.disable_dropout = True
    yield m
    rt.disable_dropout = False


def test_train_step_golden(dev, train_model, golden, ref_state_dict):
    """Ten loss scalars, global grad norm, never-touched parameters and sampled gradients of one train step
    vs the fixture captured from the reference (dropout off, train-mode BatchNorm)."""
    from golden.make_golden import grad_sample
    from styler_amd.training import train_losses
    g = golden("train_step")
    b = _batch(golden("full_teacher"), dev)
    train_model.zero_grad(set_to_none=True)
    losses = train_losses(train_model, b)
    got = torch.stack([l.detach().float().reshape(()) for l in losses]).cpu().numpy()
    assert np.max(np.abs(got - g["losses"])) <= 2e-3 * max(1.0, float(np.max(np.abs(g["losses"])))), (got, g["losses"])
    losses[0].backward()
    named = dict(train_model.named_parameters())
    sq = sum(float((p.grad.double() ** 2).sum()) for p in named.values() if p.grad is not None)
    assert abs(sq ** 0.5 - float(g["grad_norm"])) <= 2e-3 * float(g["grad_norm"]), (sq ** 0.5, float(g["grad_norm"]))
    no_grad = sorted(k for k, p in named.items() if p.requires_grad and (p.grad is None or float(p.grad.abs().max()) == 0.0))
    assert no_grad == sorted(str(k) for k in g["no_grad_keys"]), no_grad
    for k in g.files:
        if k.startswith("g:"):
            ref = g[k]
            gotg = grad_sample(named[k[2:]].grad).cpu().numpy()
            scale = max(1e-6, float(np.max(np.abs(ref))))
            err = float(np.max(np.abs(gotg - ref))) / scale
            # This fixture batch sits on a discontinuity (a gate whose pre-activation is within one ulp of zero): scaling
            # its float inputs by (1 + 2^-23) moves THESE samples by up to 7.3e-3 (decoder.layer_stack.3.pos_ffn.w_1.weight;
            # 1e-3 on the attention projections) in one and the same build, and two builds that differ only in how
            # LayerNorm's wave reduction is scheduled land on either side of it (tools/dbg_grads.py reproduces both).
            # The bound covers the two states; the gradient norm above and the oracle comparison at the benched shape
            # (tests/test_11_oracle_c2c3.py, 1e-4-level) are the tight checks.
            assert err <= 1.5e-2, f"{k}: rel err {err:.3e}"
    train_model.load_state_dict(ref_state_dict)


def test_train_step_vs_oracle_vctk_shape(dev, train_model, ref_state_dict):
    from closed_form import make_batch
    from oracle import styler_oracle as O
    from styler_amd.training import train_losses
    b = make_batch(4, 20, 40, 2, 9, seed=31)
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "position_enc" not in k and "_bins" not in k
             and "running_" not in k else v.clone()) for k, v in ref_state_dict.items()}
    ref = O.train_losses(P, b, training="bn_only")
    ref[0].backward()
    train_model.load_state_dict(ref_state_dict)
    train_model.zero_grad(set_to_none=True)
    losses = train_losses(train_model, {k: v.to(dev) for k, v in b.items()})
    for a, e in zip(losses, ref):
        assert abs(float(a) - float(e)) <= 2e-3 * max(1.0, abs(float(e))), (float(a), float(e))
    losses[0].backward()
    worst = 0.0
    for k, p in train_model.named_parameters():
        if P[k].grad is None:
            continue
        # w_ks.bias has an analytically zero gradient (softmax is shift-invariant over keys): floor the scale
        e = float((p.grad.cpu() - P[k].grad).abs().max()) / max(float(P[k].grad.abs().max()), 1e-4)
        worst = max(worst, e)
        assert e <= 1e-2, f"{k}: rel grad err {e:.3e}"
    train_model.load_state_dict(ref_state_dict)


def test_train_state_steps_and_bf16(dev, ref_state_dict):
    """Flat-buffer optimiser: two steps reduce nothing to NaN, parameters move, derived layouts refresh; bf16 mode
    gradients stay close to fp32 ones."""
    from closed_form import make_batch
    from styler_amd import STYLER, rt
    from styler_amd.training import TrainState, train_step
    m = STYLER()
    m.load_state_dict(ref_state_dict)
    m = m.to(dev).train()
    st = TrainState(m)
    b = {k: v.to(dev) for k, v in make_batch(4, 20, 40, 2, 9, seed=32).items()}
    w0 = m.decoder.layer_stack[0].pos_ffn.w_1.weight.detach().clone()
    l1, lr1 = train_step(m, st, b)
    l2, lr2 = train_step(m, st, b)
    assert abs(lr1 - 256 ** -0.5 * 4000 ** -1.5) < 1e-12 and lr2 > lr1
    assert all(torch.isfinite(x).all() for x in l2) and torch.isfinite(st.flat_p).all()
    assert float((m.decoder.layer_stack[0].pos_ffn.w_1.weight - w0).abs().max()) > 0
    assert m.decoder.layer_stack[0].pos_ffn.w_1.weight.data_ptr() >= st.flat_p.data_ptr()
    # bf16 mode on the same weights and batch: same losses and the same flat gradient within the tolerance stated in
    # tests/test_11_oracle_c2c3.py (which pins both modes to the oracle at the benched shape)
    from styler_amd.training import forward_backward
    rt.disable_dropout = True
    try:
        lf = [float(x) for x in forward_backward(m, st, b)]
        gf = st.flat_g.clone()
        st.zero_grad()
        rt.set_precision("bf16")
        lb = [float(x) for x in forward_backward(m, st, b)]
        gb = st.flat_g.clone()
        st.zero_grad()
        assert max(abs(x - y) / max(1.0, abs(x)) for x, y in zip(lf, lb)) <= 1e-2, (lf, lb)
        assert float((gf - gb).norm() / gf.norm()) <= 5e-2
        l3, _ = train_step(m, st, b)
        assert all(torch.isfinite(x).all() for x in l3) and st.adam_steps == 3 and st.n_current_steps == 3
    finally:
        rt.set_precision("fp32")
        rt.disable_dropout = False
        st.close()


def test_train_state_checkpoint_round_trip(dev, ref_state_dict, tmp_path):
    """Reference-format checkpoint ({'model': module.-prefixed 328 keys, 'optimizer': torch.optim.Adam state},
    train.py:221-224) written after two steps; a fresh model + TrainState restored from it (train.py:61-66) takes the same
    third step as the original run (same parameters, same Noam rate, same Adam bias correction), and the optimizer half
    loads into a stock torch.optim.Adam."""
    from closed_form import make_batch
    from styler_amd import STYLER, hparams as hp, rt
    from styler_amd.checkpoint import load_checkpoint, save_checkpoint
    from styler_amd.training import TrainState, train_step
    b = {k: v.to(dev) for k, v in make_batch(4, 20, 40, 2, 9, seed=36).items()}
    rt.disable_dropout = True
    try:
        m = STYLER()
        m.load_state_dict(ref_state_dict)
        m = m.to(dev).train()
        st = TrainState(m)
        for _ in range(2):
            train_step(m, st, b)
        path = str(tmp_path / "checkpoint_2.pth.tar")
        save_checkpoint(path, m, st)
        _, lr3 = train_step(m, st, b)
        p3 = {k: v.detach().clone() for k, v in m.state_dict().items()}
        st.close()

        ckpt = torch.load(path, map_location="cpu")
        assert len(ckpt["model"]) == 328 and all(k.startswith("module.") for k in ckpt["model"])
        m2 = STYLER().to(dev).train()
        st2 = TrainState(m2)
        load_checkpoint(path, m2, st2)
        assert st2.n_current_steps == 2 and st2.adam_steps == 2
        _, lr3b = train_step(m2, st2, b)
        assert lr3b == lr3
        for k, v in m2.state_dict().items():
            assert float((v.float() - p3[k].float()).abs().max()) <= 1e-6 * max(1.0, float(p3[k].float().abs().max())), k
        opt = torch.optim.Adam(m2.parameters(), betas=hp.betas, eps=hp.eps, weight_decay=hp.weight_decay)
        opt.load_state_dict(ckpt["optimizer"])
        some = next(iter(opt.state.values()))
        assert int(some["step"]) == 2 and some["exp_avg"].is_cuda
        # resuming WITHOUT optimizer state: the schedule continues at restore_step, Adam's bias correction restarts
        st3 = TrainState(STYLER().to(dev).train(), restore_step=1000)
        assert st3.n_current_steps == 1000 and st3.adam_steps == 0
        st2.close(); st3.close()
    finally:
        rt.disable_dropout = False


def test_acc_steps_gate(dev, ref_state_dict, monkeypatch):
    """train.py:175-185 with acc_steps = 2: the loss is halved, the first micro-batch only accumulates (no update, no
    zero_grad), the second one updates with the sum of both gradients."""
    from closed_form import make_batch
    from styler_amd import STYLER, hparams as hp, rt
    from styler_amd.training import TrainState, forward_backward, train_step
    b1 = {k: v.to(dev) for k, v in make_batch(3, 20, 40, 2, 9, seed=37).items()}
    b2 = {k: v.to(dev) for k, v in make_batch(3, 20, 40, 2, 9, seed=38).items()}
    rt.disable_dropout = True
    try:
        m = STYLER()
        m.load_state_dict(ref_state_dict)
        m = m.to(dev).train()
        st = TrainState(m)
        forward_backward(m, st, b1)
        g1 = st.flat_g.clone()
        st.zero_grad()
        forward_backward(m, st, b2)
        g2 = st.flat_g.clone()
        st.zero_grad()
        p0 = st.flat_p.clone()
        monkeypatch.setattr(hp, "acc_steps", 2)
        _, lr = train_step(m, st, b1)
        assert lr is None and st.n_current_steps == 0 and torch.equal(st.flat_p, p0)
        assert float((st.flat_g - 0.5 * g1).abs().max()) <= 1e-6 * float(g1.abs().max())
        _, lr = train_step(m, st, b2)
        assert lr is not None and st.n_current_steps == 1 and not torch.equal(st.flat_p, p0)
        want = 0.5 * (g1 + g2)
        assert float((st.flat_g - want).abs().max()) <= 2e-5 * float(want.abs().max())
        st.close()
    finally:
        rt.disable_dropout = False


def test_bucketed_batches_oracle_on_same_rectangle_and_graph_cache(dev, ref_state_dict):
    """A batch padded up to a shape bucket (data.to_device(bucket=...)): (1) the ten losses and sampled gradients equal the
    oracle run on the SAME padded rectangle (T padding is expressible in the reference: its collate pads, train.py:132
    passes the extent); (2) two batches with different exact shapes that fall into one bucket replay ONE captured graph
    (GraphedStepCache) and walk the same trajectory as eager steps on the same padded tensors."""
    from closed_form import make_batch
    from oracle import styler_oracle as O
    from styler_amd import STYLER, rt
    from styler_amd.training import GraphedStepCache, TrainState, train_losses, train_step

    def pad_t(b, T):
        out = dict(b)
        for k in ("mel_target", "mel_aug", "f0", "f0_norm", "f0_norm_aug", "energy", "energy_input", "energy_input_aug"):
            v = b[k]
            out[k] = torch.cat([v, v.new_zeros(v.shape[0], T - v.shape[1], *v.shape[2:])], dim=1)
        return out

    def pad_s(b, S):
        out = dict(b)
        for k in ("text", "D", "log_D"):
            v = b[k]
            out[k] = torch.cat([v, v.new_zeros(v.shape[0], S - v.shape[1])], dim=1)
        return out

    rt.disable_dropout = True
    try:
        b1 = make_batch(4, 16, 30, 2, 9, seed=51)
        T1 = b1["mel_target"].shape[1]
        Tb = -(-T1 // 64) * 64
        assert Tb > T1
        p1 = pad_t(b1, Tb)
        P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "position_enc" not in k and "_bins" not in k
                 and "running_" not in k else v.clone()) for k, v in ref_state_dict.items()}
        ref = O.train_losses(P, p1, training="bn_only", max_mel_len=Tb)
        ref[0].backward()
        m = STYLER()
        m.load_state_dict(ref_state_dict)
        m = m.to(dev).train()
        losses = train_losses(m, {k: v.to(dev) for k, v in p1.items()})
        for a, e in zip(losses, ref):
            assert abs(float(a) - float(e)) <= 2e-3 * max(1.0, abs(float(e))), (float(a), float(e))
        losses[0].backward()
        for k in ("decoder.layer_stack.0.slf_attn.w_qs.weight", "postnet.convolutions.2.0.conv.weight",
                  "style_modeling.style_encoder.audio_encoder.convolutions_2.1.0.conv.weight",
                  "style_modeling.pitch_embedding.weight"):
            g, r = dict(m.named_parameters())[k].grad.cpu(), P[k].grad
            assert float((g - r).abs().max()) <= 1e-2 * float(r.abs().max()), k
        # and the exact-shape batch gives (slightly) different losses: the padded statistics are part of the model
        exact = O.train_losses({k: v.detach() for k, v in P.items()}, b1, training="bn_only")
        assert abs(float(exact[0]) - float(ref[0])) > 0

        # ---- graph cache: two exact shapes, one bucket, one capture ----
        b2 = make_batch(4, 16, 30, 2, 9, seed=52)
        Sb = max(b1["text"].shape[1], b2["text"].shape[1])
        Tb = -(-max(T1, b2["mel_target"].shape[1]) // 64) * 64
        q1 = {k: v.to(dev) for k, v in pad_s(pad_t(b1, Tb), Sb).items()}
        q2 = {k: v.to(dev) for k, v in pad_s(pad_t(b2, Tb), Sb).items()}
        finals = []
        for mode in ("eager", "cache"):
            m = STYLER()
            m.load_state_dict(ref_state_dict)
            m = m.to(dev).train()
            st = TrainState(m)
            if mode == "eager":
                for q in (q1, q2, q1):
                    losses, lr = train_step(m, st, q)
            else:
                cache = GraphedStepCache(m, st, max_graphs=2)
                for q in (q1, q2, q1):
                    losses, lr = cache(q)
                assert (cache.misses, cache.hits, len(cache.steps)) == (1, 2, 1)
            finals.append((torch.stack([x.detach().float().reshape(()) for x in losses]).cpu(), st.flat_p.clone(), lr))
            st.close()
        (l_e, p_e, lr_e), (l_c, p_c, lr_c) = finals
        assert lr_e == lr_c
        assert float((l_e - l_c).abs().max()) <= 2e-4 * max(1.0, float(l_e.abs().max())), (l_e, l_c)
        assert float((p_e - p_c).abs().max()) <= 1e-4
    finally:
        rt.disable_dropout = False
