"""bf16x3 arithmetic, round 5: producers write the operand split themselves (styler_set_x3_out).  Every producer -- the GEMM
epilogue of each engine (64 x 64, 128 x 128, 256 x 256 LDS-DMA, both split-K combine passes, packed rows, length masks),
LayerNorm, GroupNorm / BatchNorm forward and backward -- must file a split that is BIT-IDENTICAL to what the stand-alone pass
(styler_split3_bf16) makes of the fp32 output it wrote next to it, in both storage forms ([hi | lo] and [hi | lo | hi]); and the
bf16x3 training step with the producers on must give the gradients of the step with every split as a separate pass."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class _Cache:
    """ops.x3_cache as a training step sets it (a dict), for the duration of a block."""

    def __enter__(self):
        from styler_amd import ops
        self.prev, ops.x3_cache = ops.x3_cache, {}
        return ops.x3_cache

    def __exit__(self, *a):
        from styler_amd import ops
        ops.x3_cache = self.prev


def _filed(cache, y, plan=None):
    key = (y.data_ptr(), tuple(y.shape), y.stride(), plan.counts.data_ptr() if plan is not None else 0)
    assert key in cache, "the producer did not file a split for its output"
    return cache[key][2]


def _standalone(y, plan=None):
    from styler_amd import ops
    prev, ops.x3_cache = ops.x3_cache, None
    try:
        return ops.split3(y, plan)
    finally:
        ops.x3_cache = prev


def _same(got, y, what, plan=None, rows=None):
    ref = _standalone(y, plan)
    assert got.shape == ref.shape and got.dtype == ref.dtype, (what, got.shape, ref.shape)
    a, b = got.view(torch.int16), ref.view(torch.int16)
    if rows is not None:                               # packed rows: only the valid prefix is defined
        a, b = a.reshape(-1, a.shape[-1])[:rows], b.reshape(-1, b.shape[-1])[:rows]
    assert torch.equal(a, b), f"{what}: {int((a != b).sum())} of {a.numel()} bf16 words differ from styler_split3_bf16's"


CASES = {  # B, L, cin, n, kw, act, x16, lens, residual
    "tile64_linear": (3, 50, 256, 256, 1, 0, False, None, False),
    "tile64_triple_80": (2, 70, 256, 80, 1, 0, False, None, False),            # 80 % 64 != 0: [hi | lo | hi]
    "tile128_k9_relu": (2, 700, 256, 1024, 9, 1, False, None, False),
    "tile128_lens_zero_tiles": (3, 400, 256, 256, 3, 0, False, [400, 130, 7], True),
    "gemm256_k9_relu_x16": (100, 256, 256, 1024, 9, 1, True, None, False),      # 400 tiles of 256 x 256
    "gemm256_split_k_dx": (1, 27060, 1024, 256, 9, 0, True, None, True),        # 106 tiles, K = 9216: split-K = 2 + combine
    "small_m_split_k": (48, 60, 1024, 256, 9, 0, False, None, False),           # 64 x 64 tile split-K + combine
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_gemm_epilogue_files_the_split(dev, case):
    from styler_amd import ops
    B, L, cin, n, kw, act, x16, lens, residual = CASES[case]
    g = torch.Generator().manual_seed(len(case) * 7 + n)
    x = torch.randn(B, L, cin, generator=g).to(dev)
    w = (torch.randn(n, kw * cin, generator=g) / np.sqrt(kw * cin)).to(torch.bfloat16).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    res = torch.randn(B, L, n, generator=g).to(dev) if residual else None
    ln = torch.tensor(lens).to(dev) if lens is not None else None
    xa = x.to(torch.bfloat16) if x16 else x
    plain = ops.conv_gemm(xa, w, b, kw=kw, n=n, act=act, prec=ops.PREC_BF16, res=res, lens=ln)
    with _Cache() as cache:
        y = ops.conv_gemm(xa, w, b, kw=kw, n=n, act=act, prec=ops.PREC_BF16, res=res, lens=ln, x3_out=True)
        got = _filed(cache, y)
    assert torch.equal(y, plain), f"{case}: the fp32 output changed when the split was asked for"
    assert got.shape[-1] == (2 if n % 64 == 0 else 3) * n
    _same(got, y, case)


def test_gemm_epilogue_files_the_split_packed_rows(dev):
    from styler_amd import ops
    g = torch.Generator().manual_seed(5)
    Bi, T, cin, n, kw = 4, 300, 256, 1024, 9
    lens = torch.tensor([300, 211, 40, 1])
    xs = torch.randn(Bi, T, cin, generator=g) * (torch.arange(T)[None, :, None] < lens[:, None, None])
    plan = ops.PackPlan(lens.to(dev), Bi, T)
    xp = ops.pack_rows(xs.to(dev), plan)
    w = (torch.randn(n, kw * cin, generator=g) / np.sqrt(kw * cin)).to(torch.bfloat16).to(dev)
    with _Cache() as cache:
        y = ops.conv_gemm(xp, w, None, kw=kw, n=n, act=1, prec=ops.PREC_BF16, plan=plan, x3_out=True)
        got = _filed(cache, y, plan)
    _same(got, y, "packed rows", plan=plan, rows=int(lens.sum()))


def test_norm_producers_file_the_split(dev):
    from styler_amd import ops
    g = torch.Generator().manual_seed(11)
    # LayerNorm(dropout(o) + x) with a length mask (the FFT sublayers' tail)
    B, L = 3, 77
    lens = torch.tensor([77, 30, 1]).to(dev)
    o, x = torch.randn(B, L, 256, generator=g).to(dev), torch.randn(B, L, 256, generator=g).to(dev)
    ga, be = torch.randn(256, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
    with _Cache() as cache:
        y = ops.add_layernorm(o, ga, be, res=x, lens=lens, sum_out=torch.empty_like(x), x3=True)
        _same(_filed(cache, y), y, "add_layernorm")
    # ... and its backward: the gradient that feeds the sublayer's GEMMs (dx_drop with input dropout, else dx)
    dyl = torch.randn(B, L, 256, generator=g).to(dev)
    for p_in in (0.0, 0.2):
        with _Cache() as cache:
            dg, db = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
            r = ops.layernorm_bwd(o + x, dyl, ga, be, dg, db, lens=lens, in_drop_p=p_in, in_drop_seed=5, x3=True)
            d_o = r[1] if p_in > 0 else r
            _same(_filed(cache, d_o), d_o, f"layernorm_bwd in_drop_p={p_in}")
    # GroupNorm + ReLU forward / backward: single-pass kernels (L <= 512) and the two-kernel forms (L = 700)
    for C, Lg in ((256, 441), (320, 441), (256, 700)):
        xg = (torch.randn(2, Lg, C, generator=g) * 2 + 0.3).to(dev)
        gg, bg = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
        st = torch.empty(2, C // 16, 2, device=dev)
        dy = torch.randn(2, Lg, C, generator=g).to(dev)
        with _Cache() as cache:
            yg = ops.groupnorm_relu(xg, gg, bg, out=torch.empty_like(xg), stats=st, x3=True)
            _same(_filed(cache, yg), yg, f"groupnorm_relu C={C} L={Lg}")
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            dx = ops.groupnorm_relu_bwd(xg, dy, gg, bg, st, dg, db, x3=True)
            _same(_filed(cache, dx), dx, f"groupnorm_relu_bwd C={C} L={Lg}")
    # BatchNorm (train) + tanh + dropout forward / backward, two segments (the paired PostNet), 80 channels -> triple form
    for C in (512, 80):
        xb = (torch.randn(4, 61, C, generator=g) * 1.5).to(dev)
        gb, bb = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
        dyb = torch.randn(4, 61, C, generator=g).to(dev)
        with _Cache() as cache:
            yb, mean, rstd = ops.batchnorm_train(xb, gb, bb, None, None, ops.ACT_TANH, drop_p=0.1, drop_seed=77, segs=2, x3=True)
            _same(_filed(cache, yb), yb, f"batchnorm_train C={C}")
            assert _filed(cache, yb).shape[-1] == (2 if C % 64 == 0 else 3) * C
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            dxb = ops.batchnorm_bwd(xb, None, dyb, gb, mean, rstd, dg, db, ops.ACT_TANH, beta=bb, drop_p=0.1, drop_seed=77,
                                    segs=2, x3=True)
            _same(_filed(cache, dxb), dxb, f"batchnorm_bwd C={C}")


@pytest.mark.parametrize("packed", [False, True])
def test_attention_x3_kernels_file_the_splits(dev, packed):
    """The x3 attention kernels: the output's split (for the out-projection GEMM) and dqkv's (QKV dX GEMM + weight gradients),
    padded rectangle (padded query rows are computed, all-padding key blocks zeroed) and packed rows."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(23)
    B, L = 3, 300
    lens = torch.tensor([300, 140, 9])
    valid = (torch.arange(L)[None, :, None] < lens[:, None, None])
    qkv = (torch.randn(B, L, 768, generator=g) * valid).to(dev)
    dout = (torch.randn(B, L, 256, generator=g) * valid).to(dev)
    plan, rows = None, None
    if packed:
        plan = ops.PackPlan(lens.to(dev), B, L)
        qkv, dout = ops.pack_rows(qkv, plan), ops.pack_rows(dout, plan)
        rows = int(lens.sum())
    lse = torch.empty(B, 4, L, device=dev)
    with _Cache() as cache:
        att = ops.attention_fwd(qkv, lens.to(dev), lse=lse, prec=ops.PREC_BF16X3, plan=plan, x3=True)
        _same(_filed(cache, att, plan), att, "attention_fwd_x3", plan=plan, rows=rows)
        dqkv = ops.attention_bwd(qkv, att, dout, lse, lens.to(dev), prec=ops.PREC_BF16X3, plan=plan, x3=True)
        _same(_filed(cache, dqkv, plan), dqkv, "attention_bwd_x3", plan=plan, rows=rows)
    plain = ops.attention_fwd(qkv, lens.to(dev), lse=torch.empty_like(lse), prec=ops.PREC_BF16X3, plan=plan)
    assert torch.equal(plain if rows is None else plain[:, :rows], att if rows is None else att[:, :rows])


def test_registration_is_consumed_and_rejected_where_it_cannot_be_honoured(dev):
    from styler_amd import ops
    lib = ops.lib
    x = torch.randn(2, 64, 256, device=dev)
    ga, be = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    y3 = torch.empty(2, 64, 512, device=dev, dtype=torch.bfloat16)
    # a bf16 output cannot carry a split: loud failure, and the registration is gone afterwards
    assert lib.styler_set_x3_out(y3.data_ptr(), 2) == 0
    with pytest.raises(ops.StylerHipError):
        ops.add_layernorm(x, ga, be, out_bf16=True)
    y3.zero_()
    ops.add_layernorm(x, ga, be)                       # would fill y3 if the registration had survived
    torch.cuda.synchronize()
    assert float(y3.float().abs().sum()) == 0.0
    assert lib.styler_set_x3_out(y3.data_ptr(), 4) != 0 and lib.styler_set_x3_out(None, 0) == 0


def test_bf16x3_train_step_producers_equal_separate_passes(dev, ref_state_dict):
    """The bf16x3 training step with the splits written by their producers = the step with every split as its own pass:
    the splits are bit-identical, so losses and every parameter gradient must be bit-identical too."""
    from closed_form import make_batch
    from styler_amd import STYLER, ops, rt
    from styler_amd.training import TrainState, forward_backward
    b = {k: v.to(dev) for k, v in make_batch(6, 20, 40, 2, 9, seed=3).items()}
    rt.set_precision("bf16x3")
    prev_drop, rt.disable_dropout = rt.disable_dropout, True
    outs = []
    try:
        for on in (True, False):
            ops.x3_producers = on
            torch.manual_seed(0)
            m = STYLER()
            m.load_state_dict(ref_state_dict)
            m = m.to(dev).train()
            st = TrainState(m)
            for _ in range(2):                          # (the first pass sizes the arena)
                st.zero_grad()
                losses = forward_backward(m, st, b)
            outs.append((torch.stack([l.detach().float().reshape(()) for l in losses]).cpu(), st.flat_g.clone()))
    finally:
        ops.x3_producers = True
        rt.disable_dropout = prev_drop
        rt.set_precision("fp32")
    assert torch.equal(outs[0][0], outs[1][0]), (outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]), f"{int((outs[0][1] != outs[1][1]).sum())} gradient entries differ"


def test_bf16x3_train_step_triple_form(dev, ref_state_dict):
    """STYLER_X3_COMPACT=0 (every split stored as [hi | lo | hi]): the x3 attention kernels and layernorm_bwd can only write the
    compact form, so they must NOT register as producers there (round-5 advisor: the step used to die with STYLER_EINVAL on
    the first attention sublayer) -- the consumers make those splits as passes.  The two storage forms hold the same hi / lo
    values, so the step's losses and gradients agree to accumulation-order noise."""
    from closed_form import make_batch
    from styler_amd import STYLER, ops, rt
    from styler_amd.training import TrainState, forward_backward
    b = {k: v.to(dev) for k, v in make_batch(6, 20, 40, 2, 9, seed=3).items()}
    rt.set_precision("bf16x3")
    prev_drop, rt.disable_dropout = rt.disable_dropout, True
    prev_compact = ops.x3_compact
    outs = []
    try:
        for compact in (True, False):
            ops.x3_compact = compact
            torch.manual_seed(0)
            m = STYLER()
            m.load_state_dict(ref_state_dict)
            m = m.to(dev).train()
            st = TrainState(m)
            for _ in range(2):
                st.zero_grad()
                losses = forward_backward(m, st, b)
            outs.append((torch.stack([l.detach().float().reshape(()) for l in losses]).cpu(), st.flat_g.clone()))
    finally:
        ops.x3_compact = prev_compact
        rt.disable_dropout = prev_drop
        rt.set_precision("fp32")
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6), (outs[0][0], outs[1][0])
    d = float((outs[0][1] - outs[1][1]).abs().max()) / float(outs[0][1].abs().max())
    assert d <= 1e-5, f"compact vs triple storage: gradients differ by {d:.3e} of the largest entry"
