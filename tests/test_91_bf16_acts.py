"""bf16 storage of the activations / gradients between the convolutions of the AudioEncoder and PostNet stacks (throughput
mode): the claim is that nothing but the storage format changes.  Checked here kernel by kernel: a bf16 output is the
round-to-nearest-even of the fp32 output, a kernel fed the bf16 tensor gives what it gives when fed the same values as fp32,
and the one-node conv + norm Function equals the two-node chain."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def _bn_inputs(dev, rows=2 * 3 * 131, C=512):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(6, rows // 6, C, generator=g).to(dev)
    dy = torch.randn(6, rows // 6, C, generator=g).to(dev)
    w = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    b = (0.1 * torch.randn(C, generator=g)).to(dev)
    return x, dy, w, b


def test_batchnorm_bf16_output_is_rounded_fp32(dev):
    from styler_amd import ops
    x, dy, w, b = _bn_inputs(dev)
    C = x.shape[-1]
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y32, m, r = ops.batchnorm_train(x, w, b, rm.clone(), rv.clone(), ops.ACT_TANH, drop_p=0.5, drop_seed=5, segs=2)
    y16, m2, r2 = ops.batchnorm_train(x, w, b, rm.clone(), rv.clone(), ops.ACT_TANH, drop_p=0.5, drop_seed=5, segs=2, out_bf16=True)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16))
    assert torch.equal(m, m2) and torch.equal(r, r2)
    # backward: bf16 dy in == the same values as fp32 in; bf16 dx out == rounded fp32 dx
    dy16 = dy.to(torch.bfloat16)
    outs = []
    for dyv, o16 in ((dy16.float(), False), (dy16, False), (dy16, True)):
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx = ops.batchnorm_bwd(x, None, dyv, w, m, r, dg, db, ops.ACT_TANH, beta=b, drop_p=0.5, drop_seed=5, segs=2, dx_bf16=o16)
        outs.append((dx, dg, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[2][0], outs[0][0].to(torch.bfloat16))
    for k in (1, 2):                       # parameter gradients: fp64 atomics, order-dependent in the last bits only
        assert torch.allclose(outs[0][k], outs[1][k], rtol=1e-5, atol=1e-5) and torch.allclose(outs[0][k], outs[2][k], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("L", [137, 600])
def test_groupnorm_bf16_output_is_rounded_fp32(dev, L):
    from styler_amd import ops
    g = torch.Generator().manual_seed(12)
    B, C = 6, 320
    x = torch.randn(B, L, C, generator=g).to(dev)
    dy = torch.randn(B, L, C, generator=g).to(dev)
    w = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    b = (0.1 * torch.randn(C, generator=g)).to(dev)
    st = torch.empty(B, C // 16, 2, device=dev)
    y32 = ops.groupnorm_relu(x, w, b, out=torch.empty_like(x), stats=st)
    y16 = ops.groupnorm_relu(x, w, b, out=torch.empty_like(x, dtype=torch.bfloat16), stats=torch.empty_like(st))
    assert torch.equal(y16, y32.to(torch.bfloat16))
    dy16 = dy.to(torch.bfloat16)
    res = []
    for dyv, o16 in ((dy16.float(), False), (dy16, False), (dy16, True)):
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        res.append((ops.groupnorm_relu_bwd(x, dyv, w, b, st, dg, db, dx_bf16=o16), dg, db))
    # dx: bit-equal across the storage variants in both forms (tools/gn_determinism.py: 30 repeats, 0 differing elements)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[2][0], res[0][0].to(torch.bfloat16))
    for k in (1, 2):                       # parameter gradients: fp32 atomics (114 per channel at L = 600), order-dependent
        assert torch.allclose(res[0][k], res[1][k], rtol=1e-4, atol=1e-4) and torch.allclose(res[0][k], res[2][k], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dz16,x16", [(True, True), (True, False), (False, True)])
def test_wgrad_k5_bf16_operands_equal_fp32_operands(dev, dz16, x16):
    """The k = 5 weight gradient rounds its operands to bf16 while staging: operands that already ARE bf16 give the same dW."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(13)
    B, L, n, cin = 6, 211, 320, 320
    dz = torch.randn(B, L, n, generator=g).to(dev).to(torch.bfloat16)
    x = torch.randn(B, L, cin, generator=g).to(dev).to(torch.bfloat16)
    ref_dw, ref_db = torch.zeros(n, cin, 5, device=dev), torch.zeros(n, device=dev)
    ops.wgrad(dz.float(), x.float(), ref_dw, n, cin, kw=5, db=ref_db, prec=ops.PREC_BF16)
    # mode 1 of the LDS-DMA ring (both operands bf16) keeps the register-staged kernel's split plan: bit-equal partial tiles;
    # mode 2 (default: two K groups per block, half the partial tiles) adds the same products in another grouping
    for mode, exact in ((1, True), (2, False)):
        prev = ops.lib.styler_wgrad_dma_config(mode, 0)
        try:
            dw, db = torch.zeros(n, cin, 5, device=dev), torch.zeros(n, device=dev)
            ops.wgrad(dz if dz16 else dz.float(), x if x16 else x.float(), dw, n, cin, kw=5, db=db, prec=ops.PREC_BF16)
        finally:
            ops.lib.styler_wgrad_dma_config(prev & 3, 0)
        if exact or not (dz16 and x16):
            assert torch.equal(dw, ref_dw)
        else:
            assert float((dw - ref_dw).abs().max()) <= 2e-6 * float(ref_dw.abs().max())
        assert torch.allclose(db, ref_db, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("kind", ["bn", "gn"])
def test_conv_norm_node_equals_two_nodes(dev, kind):
    """ConvNormFn (one tape node) == ConvGemmFn followed by the norm Function, fp32 mode (same kernels, same order)."""
    import torch.nn as nn
    from styler_amd import autograd as AG, ops
    from styler_amd.runtime import Derived, rt
    assert rt.prec == ops.PREC_F32
    g = torch.Generator().manual_seed(14)
    B, L, cin, n = 4, 97, 80, 320
    conv = nn.Conv1d(cin, n, 5, padding=2).to(dev)
    norm = (nn.BatchNorm1d(n) if kind == "bn" else nn.GroupNorm(n // 16, n)).to(dev)
    x0 = torch.randn(B, L, cin, generator=g).to(dev)
    dy = torch.randn(B, L, n, generator=g).to(dev)
    saved = rt.disable_dropout
    rt.disable_dropout = True
    try:
        outs = []
        for fused in (True, False):
            for p in list(conv.parameters()) + list(norm.parameters()):
                p.grad = None
            if kind == "bn":
                norm.running_mean.zero_(); norm.running_var.fill_(1.0)
            x = x0.clone().requires_grad_(True)
            cache = Derived()
            if fused:
                y = AG.ConvNormFn.apply(x, conv.weight, conv.bias, cache, "c", 5, norm, kind, ops.ACT_TANH, 0.5, 2, False)
            else:
                z = AG.ConvGemmFn.apply(x, None, conv.weight, conv.bias, cache, "c", 5, ops.ACT_NONE, False, None)
                y = (AG.BatchNormActFn.apply(z, norm.weight, norm, ops.ACT_TANH, 0.5, 2) if kind == "bn"
                     else AG.GroupNormReluFn.apply(z, norm.weight, norm))
            y.backward(dy)
            outs.append([y.detach(), x.grad] + [p.grad.clone() for p in list(conv.parameters()) + list(norm.parameters())])
    finally:
        rt.disable_dropout = saved
    for i, (a, b) in enumerate(zip(*outs)):            # y, dx exact up to nothing; parameter gradients leave as fp32 atomics
        assert torch.allclose(a, b, rtol=1e-5 if i < 2 else 2e-4, atol=1e-6 if i < 2 else 1e-4), i


def test_wgrad_linear_bf16_dz_equals_fp32_dz(dev):
    """Linear weight gradient (128 x 128 tile) with the gradient operand resident as bf16 -- the attention's dqkv -- against
    the same values handed over as fp32: the kernel rounds to bf16 while staging, so dW and the bias gradient are the same."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(14)
    B, L = 3, 517
    dqkv = torch.randn(B, L, 768, generator=g).to(dev).to(torch.bfloat16)
    x = torch.randn(B, L, 256, generator=g).to(dev)
    for i in range(3):
        dz = dqkv[..., i * 256:(i + 1) * 256]
        ref_dw, ref_db = torch.zeros(256, 256, device=dev), torch.zeros(256, device=dev)
        ops.wgrad(dz.float(), x, ref_dw, 256, 256, db=ref_db, prec=ops.PREC_BF16)
        dw, db = torch.zeros(256, 256, device=dev), torch.zeros(256, device=dev)
        ops.wgrad(dz, x, dw, 256, 256, db=db, prec=ops.PREC_BF16)
        assert torch.allclose(dw, ref_dw, rtol=1e-5, atol=1e-4) and torch.allclose(db, ref_db, rtol=1e-5, atol=1e-4)


def test_bf16_stream_kernels_equal_fp32_kernels_on_the_same_values(dev):
    """The decoder's bf16 residual stream (round 3): every kernel that reads or writes the stream in bf16 computes exactly what
    its fp32 form computes on the same (bf16-representable) values -- outputs are the round-to-nearest-even of the fp32
    outputs, parameter gradients identical sums."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(31)
    bf = torch.bfloat16
    B, L = 3, 77
    lens = torch.tensor([77, 40, 5]).to(dev)
    o = torch.randn(B, L, 256, generator=g).to(dev)                                  # GEMM output (fp32 either way)
    x16 = torch.randn(B, L, 256, generator=g).to(dev).to(bf)                          # the stream
    ga, be = (1 + 0.1 * torch.randn(256, generator=g)).to(dev), (0.1 * torch.randn(256, generator=g)).to(dev)
    # forward: y16, s16 from the bf16 kernel; the fp32 kernel on the same residual gives the unrounded sum
    s16 = torch.empty(B, L, 256, device=dev, dtype=bf)
    y16 = ops.add_layernorm(o, ga, be, res=x16, lens=lens, sum_out=s16, in_drop_p=0.2, in_drop_seed=9)
    s32 = torch.empty(B, L, 256, device=dev)
    ops.add_layernorm(o, ga, be, res=x16.float(), lens=lens, sum_out=s32, in_drop_p=0.2, in_drop_seed=9)
    valid = torch.arange(L, device=dev)[None, :] < lens[:, None]                      # (the sum is defined on unmasked rows only)
    assert y16.dtype == bf and torch.equal(s16[valid], s32[valid].to(bf))
    # ... and y is the LayerNorm of the ROUNDED sum (what the backward recomputes its statistics from)
    s_clean = torch.where(valid[..., None], s16.float(), torch.zeros((), device=dev))
    y_ref = ops.add_layernorm(s_clean, ga, be, lens=lens)
    assert torch.equal(y16, y_ref.to(bf))
    s16 = s_clean.to(bf)                                                              # defined everywhere for the backward calls below
    # backward: all four streams bf16 vs the fp32 kernel on the same values
    dy16 = torch.randn(B, L, 256, generator=g).to(dev).to(bf)
    dg1, db1, dg2, db2 = (torch.zeros(256, device=dev) for _ in range(4))
    dx16, dxd16 = ops.layernorm_bwd(s16, dy16, ga, be, dg1, db1, lens=lens, in_drop_p=0.2, in_drop_seed=9)
    dx32, dxd32 = ops.layernorm_bwd(s16.float(), dy16.float(), ga, be, dg2, db2, lens=lens, in_drop_p=0.2, in_drop_seed=9)
    assert dx16.dtype == bf and dxd16.dtype == bf
    assert torch.equal(dx16, dx32.to(bf)) and torch.equal(dxd16, dxd32.to(bf))
    # (round 6: the bf16 stream runs the sixteen-lanes-per-row kernel, the fp32 tensors the wave-per-row one -- the same
    # products and sums in a different order: fp32 rounding, not bit equality, on the two parameter gradients)
    for a_, b_ in ((dg1, dg2), (db1, db2)):
        assert float((a_ - b_).abs().max()) <= 1e-5 * float(b_.abs().max())
    # pack / unpack of a bf16 stream
    T = L
    plan = ops.PackPlan(lens, B, T)
    xp16 = ops.pack_rows(o, plan, out_bf16=True)
    xp32 = ops.pack_rows(o, plan)
    nv = int(lens.sum())
    assert xp16.dtype == bf and torch.equal(xp16[:, :nv], xp32[:, :nv].to(bf))
    assert torch.equal(ops.unpack_rows(xp16, plan), ops.unpack_rows(xp16.float(), plan))
    # GEMM epilogue: bf16 residual + bf16 output
    w = (torch.randn(256, 256, generator=g) / 16).to(dev).to(bf)
    a16 = torch.randn(B, L, 256, generator=g).to(dev).to(bf)
    y_a = ops.conv_gemm(a16, w, None, prec=ops.PREC_BF16, res=x16, out_bf16=True)
    y_b = ops.conv_gemm(a16, w, None, prec=ops.PREC_BF16, res=x16.float())
    assert y_a.dtype == bf and torch.equal(y_a, y_b.to(bf))
    # weight gradients with both operands bf16 (k = 1 and the FFN's k = 9) vs fp32 operands holding the same values
    for kw, n, cin in ((1, 256, 256), (9, 128, 256)):
        dz = torch.randn(2, 300, n, generator=g).to(dev).to(bf)
        xx = torch.randn(2, 300, cin, generator=g).to(dev).to(bf)
        shape = (n, cin) if kw == 1 else (n, cin, kw)
        dw_a, dw_b = torch.zeros(shape, device=dev), torch.zeros(shape, device=dev)
        b_a, b_b = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        prev = ops.lib.styler_wgrad_dma_config(1, 0)       # the DMA ring with the register-staged kernel's split plan
        try:
            ops.wgrad(dz, xx, dw_a, n, cin, kw=kw, db=b_a, prec=ops.PREC_BF16)
        finally:
            ops.lib.styler_wgrad_dma_config(prev & 3, 0)
        ops.wgrad(dz, xx.float(), dw_b, n, cin, kw=kw, db=b_b, prec=ops.PREC_BF16)
        assert torch.equal(dw_a, dw_b), kw
        assert float((b_a - b_b).abs().max()) <= 1e-4 * float(b_b.abs().max())
        dw_c, b_c = torch.zeros(shape, device=dev), torch.zeros(n, device=dev)
        ops.wgrad(dz, xx, dw_c, n, cin, kw=kw, db=b_c, prec=ops.PREC_BF16)       # default mode: two K groups per block
        assert float((dw_c - dw_b).abs().max()) <= 2e-6 * float(dw_b.abs().max()), kw
        assert float((b_c - b_b).abs().max()) <= 1e-4 * float(b_b.abs().max())


@pytest.mark.parametrize("nvalid", [1, 255, 256, 700, 1024])
def test_layernorm_bwd_packed_bf16_rows_vs_fp64_math(dev, nvalid):
    """The sixteen-lanes-per-row LayerNorm backward (round 6) on the decoder's packed bf16 layout -- [1, capacity, 256], the
    valid row count on the device -- against fp64 math on the same bf16 values: dx to bf16 rounding, the two parameter
    gradients to fp32 summation order; rows of the last 256-row tile behind the data are zeros (rows of tiles wholly behind
    it belong to nobody: the GEMMs that consume dx never read them); dx through the dropout mask is 0 or dx / (1 - p)."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(100 + nvalid)
    cap = 1024
    bf = torch.bfloat16
    x = torch.randn(1, cap, 256, generator=g).to(dev).to(bf)
    dy = torch.randn(1, cap, 256, generator=g).to(dev).to(bf)
    ga, be = (1 + 0.1 * torch.randn(256, generator=g)).to(dev), (0.1 * torch.randn(256, generator=g)).to(dev)
    lens = torch.tensor([nvalid], device=dev)
    dg, db = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    dx = ops.layernorm_bwd(x, dy, ga, be, dg, db, lens=lens)
    xv, dv, gv = x[0, :nvalid].double(), dy[0, :nvalid].double(), ga.double()
    mean = xv.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((xv - mean) ** 2).mean(1, keepdim=True) + 1e-5)
    h = (xv - mean) * rstd
    ex = dv * gv
    ref = rstd * (ex - ex.mean(1, keepdim=True) - h * (ex * h).mean(1, keepdim=True))
    assert dx.dtype == bf
    err = (dx[0, :nvalid].double() - ref).abs().max() / ref.abs().max()
    assert float(err) <= 2 ** -8, float(err)
    zend = min(cap, (nvalid + 255) // 256 * 256)
    assert float(dx[0, nvalid:zend].float().abs().max()) == 0.0 if zend > nvalid else True
    assert float((dg.double() - (dv * h).sum(0)).abs().max()) <= 1e-5 * float((dv * h).sum(0).abs().max())
    assert float((db.double() - dv.sum(0)).abs().max()) <= 1e-5 * float(dv.sum(0).abs().max())
    dg2, db2 = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    dx2, dxd = ops.layernorm_bwd(x, dy, ga, be, dg2, db2, lens=lens, in_drop_p=0.25, in_drop_seed=3)
    assert torch.equal(dx2[0, :zend], dx[0, :zend]) and torch.equal(dg2, dg) and torch.equal(db2, db)
    a, b_ = dxd[0, :nvalid].float(), dx[0, :nvalid].float() / 0.75
    kept = a != 0
    assert float((a - b_)[kept].abs().max()) <= 2 ** -7 * float(b_.abs().max())
    if nvalid >= 255:
        assert 0.70 <= float(kept.float().mean()) <= 0.80


def test_bf16_z_norm_kernels_equal_fp32_kernels_on_the_same_values(dev):
    """Round 3, rt.bf16_z: GroupNorm / BatchNorm forward and backward reading the convolution output as bf16
    (STYLER_IO_Z_BF16) compute exactly what they compute on an fp32 tensor holding the same (bf16-representable) values --
    same statistics, same outputs, same gradients.  GroupNorm: both single-pass item lengths (<= 512 and <= 1024 rows
    forward); BatchNorm: two segments, tanh + dropout, and the plain last layer."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(47)
    bf = torch.bfloat16
    assert ops.groupnorm_z_bf16_ok(512) and not ops.groupnorm_z_bf16_ok(513)
    for B, L, C in ((3, 441, 320), (2, 77, 256), (2, 900, 64)):
        z16 = (2.0 * torch.randn(B, L, C, generator=g) + 0.5).to(dev).to(bf)
        ga, be = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
        st_a = torch.empty(B, C // 16, 2, device=dev)
        st_b = torch.empty_like(st_a)
        y_a = ops.groupnorm_relu(z16, ga, be, out=torch.empty(B, L, C, device=dev, dtype=bf), stats=st_a)
        y_b = ops.groupnorm_relu(z16.float(), ga, be, out=torch.empty(B, L, C, device=dev, dtype=bf), stats=st_b)
        assert torch.equal(st_a, st_b) and torch.equal(y_a, y_b), (B, L, C)
        if not ops.groupnorm_z_bf16_ok(L):
            continue
        dy = torch.randn(B, L, C, generator=g).to(dev).to(bf)
        dg_a, db_a, dg_b, db_b = (torch.zeros(C, device=dev) for _ in range(4))
        dx_a = ops.groupnorm_relu_bwd(z16, dy, ga, be, st_a, dg_a, db_a, dx_bf16=True)
        dx_b = ops.groupnorm_relu_bwd(z16.float(), dy, ga, be, st_b, dg_b, db_b, dx_bf16=True)
        assert torch.equal(dx_a, dx_b), (B, L, C)
        # (the parameter gradients leave the blocks as fp32 atomics: the order, hence the last bits, varies between launches)
        for a, b in ((dg_a, dg_b), (db_a, db_b)):
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), (B, L, C)
    for rows_b, L, C, act, p, segs in ((4, 150, 512, ops.ACT_TANH, 0.5, 2), (2, 333, 80, ops.ACT_NONE, 0.0, 1)):
        z16 = (1.5 * torch.randn(rows_b, L, C, generator=g) - 0.3).to(dev).to(bf)
        ga, be = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
        y_a, m_a, r_a = ops.batchnorm_train(z16, ga, be, None, None, act, drop_p=p, drop_seed=11, segs=segs, out_bf16=True)
        y_b, m_b, r_b = ops.batchnorm_train(z16.float(), ga, be, None, None, act, drop_p=p, drop_seed=11, segs=segs, out_bf16=True)
        # column sums are fp64 atomics: mean / rstd agree to fp32 rounding of sums whose order varies
        assert torch.allclose(m_a, m_b, rtol=0, atol=1e-6) and torch.allclose(r_a, r_b, rtol=1e-6, atol=0), (C, act)
        assert float((y_a.float() - y_b.float()).abs().max()) <= 2 ** -7 * float(y_b.float().abs().max()), (C, act)
        dy = torch.randn(rows_b, L, C, generator=g).to(dev).to(bf)
        dg_a, db_a, dg_b, db_b = (torch.zeros(C, device=dev) for _ in range(4))
        dx_a = ops.batchnorm_bwd(z16, None, dy, ga, m_a, r_a, dg_a, db_a, act, beta=be, drop_p=p, drop_seed=11, segs=segs, dx_bf16=True)
        dx_b = ops.batchnorm_bwd(z16.float(), None, dy, ga, m_a, r_a, dg_b, db_b, act, beta=be, drop_p=p, drop_seed=11, segs=segs,
                                 dx_bf16=True)
        assert float((dx_a.float() - dx_b.float()).abs().max()) <= 2 ** -7 * float(dx_b.float().abs().max()), (C, act)
        for a, b in ((dg_a, dg_b), (db_a, db_b)):
            assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-3), (C, act)


def test_mel_calibrator_bf16_storage(dev):
    """rt.bf16_cat: the mel calibrator on a bf16 input gives exactly what it gives on an fp32 tensor holding the same values
    (the means are taken in fp32 either way); its backward with a bf16 output is the round-to-nearest-even of the fp32 one.
    Compression (mel_len > src_len), expansion (mel_len < src_len), equal lengths, an empty item."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(5)
    bf = torch.bfloat16
    B, T, S, C = 4, 50, 12, 1152
    mel_len = torch.tensor([50, 7, 12, 0]).to(dev)
    src_len = torch.tensor([12, 10, 12, 3]).to(dev)
    x16 = torch.randn(B, T, C, generator=g).to(dev).to(bf)
    y_a = ops.mel_calibrate(x16, mel_len, src_len, S)
    y_b = ops.mel_calibrate(x16.float(), mel_len, src_len, S)
    assert y_a.dtype == torch.float32 and torch.equal(y_a, y_b)
    dy = torch.randn(B, S, C, generator=g).to(dev)
    d_a = ops.mel_calibrate_bwd(dy, mel_len, src_len, T, out_bf16=True)
    d_b = ops.mel_calibrate_bwd(dy, mel_len, src_len, T)
    assert d_a.dtype == bf and torch.equal(d_a, d_b.to(bf))
