"""styler_linear_ln (csrc/linear_ln.hip): the Linear in front of a sublayer's LayerNorm + dropout + residual + LayerNorm +
pad mask as one launch (transformer/SubLayers.py:55-61,86-89; Layers.py:29,32).  Checked against fp64 math, against the
two-launch path it replaces (styler_conv_gemm + styler_add_layernorm: same dropout stream, same saved sum), and -- model
level -- the train step with the fusion on and off."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _case(B, L, K, seed, dev):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(B, L, K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(256, K, generator=g) / np.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(256, generator=g)
    res = torch.randn(B, L, 256, generator=g)
    ga, be = torch.randn(256, generator=g), torch.randn(256, generator=g)
    return [t.to(dev) for t in (a, w, bias, res, ga, be)]


@pytest.mark.parametrize("B,L,K,lens", [
    (3, 50, 256, [50, 7, 31]),            # encoder-like: ragged items, one partial 128-row tile
    (2, 130, 1024, [130, 1]),             # two tiles, K = 16 steps
    (1, 700, 512, [333]),                 # packed rows: capacity 700, 333 valid -- tiles 3..5 lie behind the data
    (1, 128, 64, [128]),                  # a single K step
    (5, 64, 128, None),                   # no mask
])
def test_linear_ln_vs_fp64(dev, B, L, K, lens):
    from styler_amd import ops
    a, w, bias, res, ga, be = _case(B, L, K, B * 1000 + L + K, dev)
    ln = torch.tensor(lens, device=dev) if lens is not None else None
    assert ops.linear_ln_ok(a, 256)
    s = torch.full((B, L, 256), 7.0, device=dev)
    y = ops.linear_ln(a, w, bias, res, ga, be, lens=ln, sum_out=s)
    o = a.double() @ w.double().T + bias.double() + res.double()
    ref = F.layer_norm(o, (256,), ga.double(), be.double())
    valid = torch.ones(B, L, dtype=torch.bool, device=dev)
    if lens is not None:
        valid = torch.arange(L, device=dev)[None, :] < ln[:, None]
    ref = ref * valid[..., None]
    assert float((y.double() - ref).abs().max()) <= 3e-5
    assert float(((s.double() - o) * valid[..., None]).abs().max()) <= 2e-5
    assert bool((s[~valid] == 7.0).all()), "the sum of a masked row is not written"


@pytest.mark.parametrize("io", ["fp32", "bf16"])
@pytest.mark.parametrize("B,L,K,lens", [(3, 60, 256, [60, 20, 41]), (1, 900, 1024, [513]), (1, 300, 256, [300])])
def test_linear_ln_matches_two_launches(dev, B, L, K, lens, io):
    """Same dropout draws, same statistics as styler_conv_gemm + styler_add_layernorm; in the bf16 storage format (the packed
    decoder's stream) values may differ by the rounding of a last-bit difference of the fp32 projection."""
    from styler_amd import ops
    a, w, bias, res, ga, be = _case(B, L, K, 17 * B + L + K, dev)
    ln = torch.tensor(lens, device=dev)
    if io == "bf16":
        res = res.to(torch.bfloat16)
    seed, p = 1234567, 0.1
    o = ops.conv_gemm(a, w, bias, n=256, prec=ops.PREC_BF16)
    s0 = torch.zeros_like(res)
    y0 = ops.add_layernorm(o, ga, be, res=res, lens=ln, in_drop_p=p, in_drop_seed=seed, sum_out=s0)
    s1 = torch.zeros_like(res)
    y16 = torch.full((B, L, 256), 3.0, device=dev, dtype=torch.bfloat16) if io == "fp32" else None
    y1 = ops.linear_ln(a, w, bias, res, ga, be, lens=ln, drop_p=p, drop_seed=seed, sum_out=s1, out16=y16)
    assert y1.dtype == y0.dtype and s1.dtype == s0.dtype
    valid = (torch.arange(L, device=dev)[None, :] < ln[:, None])[..., None]
    # the dropout masks agree element for element: a dropped element's sum is exactly the residual
    dropped0 = (s0.float() == res.float()) & valid
    dropped1 = (s1.float() == res.float()) & valid
    assert bool((dropped0 == dropped1).all())
    frac = float(dropped1.float().sum() / (valid.float().sum() * 256))
    assert 0.07 < frac < 0.13, frac
    if io == "fp32":
        assert float(((s1 - s0) * valid).abs().max()) <= 2e-5
        assert float((y1 - y0).abs().max()) <= 5e-5
        assert float((y16.float() - y1.to(torch.bfloat16).float()).abs().max()) == 0.0
    else:
        ds = ((s1.float() - s0.float()) * valid).abs()
        assert float((ds / (s0.float().abs() + 1.0)).max()) <= 2.0 ** -7        # one bf16 ulp where a rounding flipped
        assert float(ds.mean()) <= 1e-4
        dy = (y1.float() - y0.float()).abs()
        assert float((dy / (y0.float().abs() + 1.0)).max()) <= 4e-2             # a flipped sum element moves its row's statistics
        assert float(dy.mean()) <= 2e-4
    assert bool((y1.float()[~valid.expand_as(y1)] == 0).all())


def test_linear_ln_rejects_what_it_cannot_run(dev):
    from styler_amd import ops
    a = torch.zeros(2, 10, 96, device=dev, dtype=torch.bfloat16)            # K % 64 != 0
    assert not ops.linear_ln_ok(a, 256)
    assert not ops.linear_ln_ok(torch.zeros(2, 10, 256, device=dev, dtype=torch.bfloat16), 320)
    assert not ops.linear_ln_ok(torch.zeros(2, 10, 256, device=dev), 256)   # fp32 rows
    w = torch.zeros(256, 96, device=dev, dtype=torch.bfloat16)
    g = torch.zeros(256, device=dev)
    with pytest.raises(ops.StylerHipError):
        ops.linear_ln(a, w, None, None, g, g)


@pytest.mark.parametrize("packed", [True, False])
def test_train_step_fused_vs_two_launches(dev, ref_state_dict, packed):
    """The bf16 train step (dropout ON: the fused launch must draw the masks the backward regenerates) with rt.linear_ln on
    and off: losses and every parameter gradient."""
    from closed_form import make_batch
    from styler_amd import STYLER, rt
    from styler_amd.training import train_losses
    from test_92_model_equivalences import grads_close, BF16_FLOOR
    b = make_batch(5, 8, 30, 1, 9, seed=78)
    bd = {k: v.to(dev) for k, v in b.items()}
    m = STYLER()
    m.load_state_dict(ref_state_dict)
    m = m.to(dev).train()
    rt.set_precision("bf16")
    keep = (rt.linear_ln, rt.pack_decoder)
    try:
        rt.pack_decoder = packed
        res = []
        for fused in (False, True):
            rt.linear_ln = fused
            rt.dropout_calls = 1000                    # both forms draw the same seed sequence (autograd.next_dropout_seed)
            m.zero_grad(set_to_none=True)
            torch.manual_seed(5)
            out = train_losses(m, bd)
            out[0].backward()
            torch.cuda.synchronize()
            res.append(([float(v) for v in out[:1]], {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
        (l0, g0), (l1, g1) = res
        assert abs(l0[0] - l1[0]) <= 2e-2 * max(1.0, abs(l0[0])), (l0, l1)
        grads_close(g0, g1, 1e-1, floor=BF16_FLOOR)
    finally:
        rt.linear_ln, rt.pack_decoder = keep
        rt.set_precision("fp32")
