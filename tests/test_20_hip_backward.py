"""GPU parity of the backward / training path: HIP backward kernels (through the autograd Functions of
styler_amd.autograd) vs torch autograd on CPU (fp64 where cheap).  The full train step / optimiser tests live in tests/test_14_train_step.py."""
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


def seeded_init_(module, g):
    """Re-draw every parameter of `module` from the seeded generator `g` (nn.* constructors draw from torch's GLOBAL RNG;
    the suite seeds that per test as well, tests/conftest.py -- this keeps a test's problem independent of that too)."""
    with torch.no_grad():
        for p in module.parameters():
            bound = 1.0 / max(1.0, float(p[0].numel() if p.dim() > 1 else p.numel())) ** 0.5
            p.copy_((torch.rand(p.shape, generator=g, dtype=torch.float64) * 2 - 1).to(p.dtype) * bound)
    return module


def off_kink(z, rel=1e-4):
    """1 where the pre-activation is safely away from the activation's kink, 0 within rel * max|z| of it: an fp32 kernel
    and an fp64 reference may legitimately take different sides there."""
    return (z.detach().abs() > rel * z.detach().abs().max()).to(z.dtype)


class _Holder(nn.Module):
    pass


@pytest.mark.parametrize("B,L,cin,n,kw,act", [(3, 37, 256, 256, 1, 0), (2, 50, 256, 1024, 9, 1), (2, 41, 80, 512, 5, 2),
                                               (2, 33, 1024, 256, 1, 0), (4, 61, 256, 256, 3, 1), (5, 1, 512, 128, 1, 1)])
def test_conv_gemm_backward(dev, B, L, cin, n, kw, act):
    from styler_amd import autograd as AG
    from styler_amd.runtime import Derived
    g = torch.Generator().manual_seed(kw * 100 + n)
    conv = nn.Conv1d(cin, n, kw, padding=kw // 2).double()
    seeded_init_(conv, g)                         # weights from the seeded generator, not from the global RNG
    x = torch.randn(B, L, cin, generator=g, dtype=torch.float64, requires_grad=True)
    res = torch.randn(B, L, n, generator=g, dtype=torch.float64, requires_grad=True)
    z = conv(x.transpose(1, 2)).transpose(1, 2)
    y = (torch.relu(z) if act == 1 else torch.tanh(z) if act == 2 else z) + res
    gy = torch.randn(B, L, n, generator=g, dtype=torch.float64)
    if act == 1:                                  # ReLU kink: the kernel masks with ITS fp32 pre-activation, the reference
        gy = gy * off_kink(z)                     # with its fp64 one -- no gradient enters where the two may disagree
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
    zn = gn(x.transpose(1, 2))
    y = torch.relu(zn).transpose(1, 2)
    gy = torch.randn(2, L, C, generator=g, dtype=torch.float64) * off_kink(zn).transpose(1, 2)
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
