"""Round 6: the BiLSTM recurrences on the matrix cores (csrc/lstm_mfma.hip; modules.py:100-101,179-182 forward + autograd).

Forward: against torch's nn.LSTM in fp64 (the reference's own operator) and against the fp32 VALU kernels of csrc/lstm.hip;
BPTT: gate gradients against the VALU kernel on the same saved tensors, and the whole layer's gradients (input, W_ih, W_hh,
both biases) against nn.LSTM autograd in fp64.  parts = 1 (bf16 products): bf16-class bounds; parts = 3 (bf16x3): fp32-class
bounds.  B = 22 is not a multiple of the 4-item block (dead rows), H = 80 and H = 64 share a launch (surplus wave)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_FWD = {1: 2e-2, 3: 1e-4}          # abs, outputs in (-1, 1)
TOL_BWD = {1: 3e-2, 3: 2e-4}          # of the largest entry of the reference tensor


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _ref_lstm(H, cin, seed):
    g = torch.Generator().manual_seed(seed)
    m = torch.nn.LSTM(cin, H, 1, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * (0.5 / H ** 0.5))
    return m


def _gx(m, x):
    """x W_ih^T + b_ih + b_hh for both directions, [B, S, 2 * 4H] (what the input GEMM hands to the recurrence)."""
    f = x.double() @ m.weight_ih_l0.T + m.bias_ih_l0 + m.bias_hh_l0
    r = x.double() @ m.weight_ih_l0_reverse.T + m.bias_ih_l0_reverse + m.bias_hh_l0_reverse
    return torch.cat([f, r], -1)


def _whh(m, dev):
    return torch.stack([m.weight_hh_l0, m.weight_hh_l0_reverse]).float().contiguous().to(dev)


@pytest.mark.parametrize("parts", [1, 3])
def test_lstm_mfma_forward_and_bptt(dev, parts):
    from styler_amd import ops
    B, S = 22, 13
    Hs, cins = [80, 64, 80], [48, 32, 24]
    ms = [_ref_lstm(H, c, 11 + i) for i, (H, c) in enumerate(zip(Hs, cins))]
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(B, S, c, generator=g, dtype=torch.float64, requires_grad=True) for c in cins]
    refs = [m(x)[0] for m, x in zip(ms, xs)]
    gxs = [_gx(m, x).detach().float().contiguous().to(dev) for m, x in zip(ms, xs)]
    w_hhs = [_whh(m, dev) for m in ms]
    outs, cells, gates = ops.lstm_bidir_multi(gxs, w_hhs, Hs, save=True, parts=parts)
    outs_v, cells_v, gates_v = ops.lstm_bidir_multi(gxs, w_hhs, Hs, save=True, parts=0)
    for i in range(len(Hs)):
        e = float((outs[i].double().cpu() - refs[i].detach()).abs().max())
        assert e <= TOL_FWD[parts], f"LSTM {i} (H = {Hs[i]}): |out - nn.LSTM fp64| = {e:.3e}"
        for a, b, what in ((outs[i], outs_v[i], "out"), (cells[i], cells_v[i], "cell"), (gates[i], gates_v[i], "gates")):
            e = float((a - b).abs().max())
            assert e <= 2 * TOL_FWD[parts], f"LSTM {i} {what}: MFMA vs VALU kernel {e:.3e}"
    # ---- BPTT on the SAME saved tensors (the VALU forward's): gate gradients MFMA vs VALU ----
    douts = [torch.randn(B, S, 2 * H, generator=g).to(dev) for H in Hs]
    dg = ops.lstm_bidir_bwd_multi(douts, gates_v, cells_v, w_hhs, Hs, parts=parts)
    dg_v = ops.lstm_bidir_bwd_multi(douts, gates_v, cells_v, w_hhs, Hs, parts=0)
    for i in range(len(Hs)):
        e = float((dg[i] - dg_v[i]).abs().max()) / float(dg_v[i].abs().max())
        assert e <= TOL_BWD[parts], f"LSTM {i}: dgates MFMA vs VALU kernel, {e:.3e} of the largest entry"
    # ---- the layer's gradients against nn.LSTM autograd (fp64): dW_hh = dgp^T h_prev, dW_ih = dgp^T x, db = colsum, dx ----
    for i, (m, x, H) in enumerate(zip(ms, xs, Hs)):
        (refs[i] * douts[i].double().cpu()).sum().backward()
        dgp = dg[i].double().cpu()
        out = outs_v[i].double().cpu()
        for d, sfx in enumerate(("", "_reverse")):
            sl = dgp[..., d * 4 * H:(d + 1) * 4 * H]
            hprev = torch.zeros(B, S, H, dtype=torch.float64)
            if d == 0:
                hprev[:, 1:] = out[:, :-1, :H]
            else:
                hprev[:, :-1] = out[:, 1:, H:]
            checks = ((torch.einsum("bsj,bsk->jk", sl, hprev), getattr(m, f"weight_hh_l0{sfx}").grad, "dW_hh"),
                      (torch.einsum("bsj,bsk->jk", sl, x.detach()), getattr(m, f"weight_ih_l0{sfx}").grad, "dW_ih"),
                      (sl.sum((0, 1)), getattr(m, f"bias_ih_l0{sfx}").grad, "db"))
            for got, ref, what in checks:
                e = float((got - ref).abs().max()) / float(ref.abs().max())
                assert e <= TOL_BWD[parts], f"LSTM {i}{sfx} {what}: {e:.3e} of the largest entry"
        wih = torch.cat([m.weight_ih_l0, m.weight_ih_l0_reverse]).detach()
        e = float((dgp @ wih - x.grad).abs().max()) / float(x.grad.abs().max())
        assert e <= TOL_BWD[parts], f"LSTM {i} dx: {e:.3e} of the largest entry"


def test_lstm_mfma_repeatable(dev):
    """One barrier per step orders the LDS hand-off of h_t: the same bits on repeated launches (a race would flicker)."""
    from styler_amd import ops
    B, S, H = 96, 60, 80
    gx = torch.randn(B, S, 8 * H, device=dev)
    w = (torch.randn(2, 4 * H, H, device=dev) * 0.1).contiguous()
    first = ops.lstm_bidir_multi([gx] * 4, [w] * 4, [H] * 4, save=True, parts=1)
    for _ in range(3):
        again = ops.lstm_bidir_multi([gx] * 4, [w] * 4, [H] * 4, save=True, parts=1)
        for a, b in zip(first, again):
            assert all(torch.equal(p, q) for p, q in zip(a, b))
    dout = torch.randn(B, S, 2 * H, device=dev)
    d0 = ops.lstm_bidir_bwd_multi([dout] * 4, first[2], first[1], [w] * 4, [H] * 4, parts=1)
    d1 = ops.lstm_bidir_bwd_multi([dout] * 4, first[2], first[1], [w] * 4, [H] * 4, parts=1)
    assert all(torch.equal(p, q) for p, q in zip(d0, d1))
