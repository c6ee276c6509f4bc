"""HIP-vs-HIP equivalences: a fused / deferred / graphed form of the path against its own unfused / immediate / eager form.
These run AFTER every oracle / golden / closed-form parity test (file order = run order under `pytest -x`): a property test
that trips must not hide a parity test.  Sums whose order differs between the two forms (fp32 atomics, split-K partials) are
compared with a bound derived from the data (n * eps * sum|term|), everything else bit for bit."""
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


def order_bound(terms_abs_sum, n_terms, slack=4.0):
    """Bound on |sum_a - sum_b| of two fp32 summation orders of the same n terms: each order is within
    (n - 1) * eps * sum|term| of the exact sum (Higham, recursive summation), `slack` covers the operands' own rounding."""
    eps = 2.0 ** -24
    return slack * n_terms * eps * terms_abs_sum


@pytest.fixture(scope="module")
def train_model(dev, ref_state_dict):
    from styler_amd import STYLER, rt
    m = STYLER()
    m.load_state_dict(ref_state_dict)
    m = m.to(dev).train()
    rt.disable_dropout = True
    yield m
    rt.disable_dropout = False


def test_deferred_wgrad_reduce_matches_immediate(dev, train_model, ref_state_dict):
    """The arena path of train_step (every split-K reduction deferred to one multi-tensor launch) must produce the same
    flat gradient as the immediate per-call reduction."""
    from closed_form import make_batch
    from styler_amd import ops
    from styler_amd.training import train_losses
    b = {k: v.to(dev) for k, v in make_batch(3, 20, 40, 2, 9, seed=33).items()}
    train_model.load_state_dict(ref_state_dict)
    grads = []
    arena = ops.WgradArena()
    for mode in ("immediate", "measure", "arena"):
        train_model.zero_grad(set_to_none=True)
        losses = train_losses(train_model, b)
        if mode != "immediate":
            arena.begin(dev)
            ops.wgrad_arena = arena
        try:
            losses[0].backward()
            if mode != "immediate":
                arena.flush(dev)
        finally:
            ops.wgrad_arena = None
        grads.append({k: p.grad.clone() for k, p in train_model.named_parameters() if p.grad is not None})
    assert arena.buf is not None and arena.used > 0
    # error of a tensor against max(its own largest entry, 1e-3 x the largest gradient entry of the model): a tensor whose
    # exact gradient is zero (a conv bias in front of a BatchNorm) holds only the rounding noise of its fp32-atomics sum,
    # which changes with the order the blocks arrive in (tests/test_92_model_equivalences.py::grads_close)
    gmax = max(float(v.abs().max()) for v in grads[0].values())
    for k in grads[0]:
        for other in grads[1:]:
            e = float((grads[0][k] - other[k]).abs().max()) / max(float(grads[0][k].abs().max()), 1e-3 * gmax, 1e-4)
            assert e <= 1e-4, f"{k}: {e:.3e}"
    train_model.load_state_dict(ref_state_dict)


def test_graphed_train_step_matches_eager(dev, ref_state_dict):
    """forward + losses + backward replayed from one hipGraph (GraphedTrainStep) must walk the same trajectory as the
    eager step: same losses and same parameters after the same number of optimiser steps (dropout off: the two modes
    draw different masks).  With dropout on, two replays must draw DIFFERENT masks (device step counter)."""
    from closed_form import make_batch
    from styler_amd import STYLER, rt
    from styler_amd.training import GraphedTrainStep, TrainState, train_step
    b = {k: v.to(dev) for k, v in make_batch(4, 20, 40, 2, 9, seed=35).items()}
    rt.disable_dropout = True
    try:
        finals = []
        for mode in ("eager", "graph"):
            m = STYLER()
            m.load_state_dict(ref_state_dict)
            m = m.to(dev).train()
            st = TrainState(m)
            if mode == "eager":
                for _ in range(5):
                    losses, lr = train_step(m, st, b)
            else:
                # constructing the graphed step must not train: no optimiser step, BatchNorm running statistics, the
                # dropout step counter and the gradient buffer restored (one instance is built per padded batch shape)
                snap = (st.flat_p.clone(), st.flat_m.clone(), st.flat_v.clone(), st.drop_epoch.clone(),
                        [x.clone() for x in m.buffers()])
                g = GraphedTrainStep(m, st, b, warmup=3)
                assert st.n_current_steps == 0 and st.adam_steps == 0
                assert torch.equal(st.flat_p, snap[0]) and torch.equal(st.flat_m, snap[1]) and torch.equal(st.flat_v, snap[2])
                assert torch.equal(st.drop_epoch, snap[3]) and float(st.flat_g.abs().max()) == 0.0
                assert all(torch.equal(x, y) for x, y in zip(m.buffers(), snap[4]))
                for _ in range(5):
                    losses, lr = g(b)
            finals.append((torch.stack([x.detach().float().reshape(()) for x in losses]).cpu(), st.flat_p.clone(), lr,
                           st.n_current_steps))
        (l_e, p_e, lr_e, n_e), (l_g, p_g, lr_g, n_g) = finals
        assert n_e == n_g == 5 and lr_e == lr_g
        assert float((l_e - l_g).abs().max()) <= 2e-4 * max(1.0, float(l_e.abs().max())), (l_e, l_g)
        assert float((p_e - p_g).abs().max()) <= 1e-4
    finally:
        rt.disable_dropout = False
    m = STYLER()
    m.load_state_dict(ref_state_dict)
    m = m.to(dev).train()
    st = TrainState(m)
    g = GraphedTrainStep(m, st, b, warmup=2)
    p0, m0, v0, n0 = st.flat_p.clone(), st.flat_m.clone(), st.flat_v.clone(), st.n_current_steps
    la = torch.stack([x.detach().float().reshape(()) for x in g(b)[0]]).cpu()
    st.flat_p.copy_(p0); st.flat_m.copy_(m0); st.flat_v.copy_(v0); st.n_current_steps = n0      # same weights again
    lb = torch.stack([x.detach().float().reshape(()) for x in g(b)[0]]).cpu()
    assert torch.isfinite(la).all() and torch.isfinite(lb).all()
    assert float((la - lb).abs().max()) > 0, "two replays drew the same dropout masks"


def test_fused_dropout_add_layernorm(dev):
    """LayerNormFn with drop_p > 0 (dropout + residual + LayerNorm + mask in one kernel, mask regenerated in backward) vs
    the unfused chain styler_dropout -> add -> LayerNorm with the SAME seed (the fused kernel draws the stream
    styler_dropout would draw on the [rows, 256] tensor)."""
    from styler_amd import autograd as AG, ops, rt
    g = torch.Generator().manual_seed(5)
    B, L, p = 3, 37, 0.2
    lens = torch.tensor([37, 11, 30]).to(dev)
    x = torch.randn(B, L, 256, generator=g).to(dev)
    r = torch.randn(B, L, 256, generator=g).to(dev)
    gy = torch.randn(B, L, 256, generator=g).to(dev)
    ln = nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(256, generator=g)); ln.bias.copy_(torch.randn(256, generator=g))
    calls0 = rt.dropout_calls
    xa, ra = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    ya = AG.LayerNormFn.apply(xa, ra, ln.weight, ln, lens, p)
    ya.backward(gy)
    ga, gb = ln.weight.grad.clone(), ln.bias.grad.clone()
    seed = (rt.seed * 1000003 + calls0 + 1) & 0x7FFFFFFFFFFFFFFF          # what next_dropout_seed() handed out
    ln.weight.grad = None; ln.bias.grad = None
    xb, rb = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    yb = AG.LayerNormFn.apply(AG.DropoutFn.apply(xb, p, seed), rb, ln.weight, ln, lens, 0.0)
    yb.backward(gy)
    kept = float((ops.dropout(torch.ones_like(x), p, seed) > 0).float().mean())
    assert 0.7 < kept < 0.9
    check(ya, yb, 1e-6, "fwd"); check(xa.grad, xb.grad, 1e-6, "dx (through the mask)"); check(ra.grad, rb.grad, 1e-6, "dres")
    check(ga, ln.weight.grad, 1e-4, "dgamma"); check(gb, ln.bias.grad, 1e-4, "dbeta")     # sums over 111 rows, two orders


def test_split_graph_step_matches_single_graph(dev, ref_state_dict):
    """GraphedTrainStep(split=True) -- two graphs cut where the decoder-side gradients are final, so that their all-reduce
    can run between the replays -- must walk exactly the trajectory of the single-graph step (dropout off)."""
    from closed_form import make_batch
    from styler_amd import STYLER, rt
    from styler_amd.training import GraphedTrainStep, TrainState
    b = {k: v.to(dev) for k, v in make_batch(4, 20, 40, 2, 9, seed=36).items()}
    rt.disable_dropout = True
    try:
        finals = []
        for split in (False, True):
            m = STYLER()
            m.load_state_dict(ref_state_dict)
            m = m.to(dev).train()
            st = TrainState(m)
            g = GraphedTrainStep(m, st, b, warmup=3, split=split)
            assert len(g.graphs) == (2 if split else 1)
            for _ in range(3):
                losses, lr = g(b)
            finals.append((torch.stack([x.detach().float().reshape(()) for x in losses]).cpu(), st.flat_p.clone(), lr))
        (l1, p1, lr1), (l2, p2, lr2) = finals
        assert lr1 == lr2
        assert float((l1 - l2).abs().max()) <= 2e-4 * max(1.0, float(l1.abs().max())), (l1, l2)
        assert float((p1 - p2).abs().max()) <= 1e-4
    finally:
        rt.disable_dropout = False


def test_fused_batchnorm_tanh_dropout(dev):
    """BatchNormActFn with drop_p > 0 (BatchNorm batch statistics + tanh + dropout in one pass; backward regenerates the
    mask and recomputes tanh from x) vs the unfused chain BatchNormActFn(drop 0) -> DropoutFn with the SAME seed."""
    from styler_amd import autograd as AG, rt
    g = torch.Generator().manual_seed(9)
    B, L, C, p = 3, 29, 512, 0.5
    x = (torch.randn(B, L, C, generator=g) * 1.5 + 0.2).to(dev)
    gy = torch.randn(B, L, C, generator=g).to(dev)
    res = {}
    for fused in (True, False):
        bn = nn.BatchNorm1d(C).to(dev)
        with torch.no_grad():
            bn.weight.copy_(torch.randn(C, generator=torch.Generator().manual_seed(1)))
            bn.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(2)))
        xa = x.clone().requires_grad_(True)
        calls0 = rt.dropout_calls
        if fused:
            y = AG.BatchNormActFn.apply(xa, bn.weight, bn, AG.TANH, p)
            seed = (rt.seed * 1000003 + calls0 + 1) & 0x7FFFFFFFFFFFFFFF
        else:
            y = AG.DropoutFn.apply(AG.BatchNormActFn.apply(xa, bn.weight, bn, AG.TANH, 0.0), p, res["seed"])
        y.backward(gy)
        res[fused] = (y.detach(), xa.grad, bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_var.clone())
        if fused:
            res["seed"] = seed
    for a, c, what in zip(res[True], res[False], ("y", "dx", "dgamma", "dbeta", "running_var")):
        check(a, c, 2e-5, what)
    assert 0.4 < float((res[True][0] != 0).float().mean()) < 0.6


def test_predictor_stage_kernels_equal_their_parts(dev):
    """The fused StylePredictor stage: LayerNorm with dropout on its output == dropout(LayerNorm), and ONE LayerNorm-backward
    kernel (dropout mask regenerated, ReLU mask of its input applied) == dropout backward -> LayerNorm backward -> act_bwd."""
    from styler_amd import ops
    g = torch.Generator().manual_seed(21)
    B, L, p, seed = 5, 173, 0.5, 77
    h = torch.relu(torch.randn(B, L, 256, generator=g)).to(dev)             # a ReLU output, as the conv epilogue leaves it
    dy = torch.randn(B, L, 256, generator=g).to(dev)
    ga, be = (1 + 0.1 * torch.randn(256, generator=g)).to(dev), (0.1 * torch.randn(256, generator=g)).to(dev)
    y1 = ops.add_layernorm(h, ga, be, drop_p=p, drop_seed=seed)
    y2 = ops.dropout(ops.add_layernorm(h, ga, be), p, seed)
    assert torch.equal(y1, y2)
    assert 0.4 < float((y1 == 0).float().mean()) < 0.6
    dg1, db1, dg2, db2 = (torch.zeros(256, device=dev) for _ in range(4))
    dz1 = ops.layernorm_bwd(h, dy, ga, be, dg1, db1, drop_p=p, drop_seed=seed, relu_input=True)
    keep = (y2 != 0) | (ops.add_layernorm(h, ga, be) == 0)                  # the mask of the same stream
    d_ln = ops.layernorm_bwd(h, dy * keep / (1 - p), ga, be, dg2, db2)
    dz2 = ops.act_bwd(d_ln, h, ops.ACT_RELU)
    assert torch.equal(dz1, dz2)
    # Parameter gradients: sums over the B * L rows.  Both calls take the per-block-slot + fixed-order-fold path
    # (ops.layernorm_bwd never adds fp32 atomics into one vector), so a repeated call gives the same bits; the two FORMS are
    # compared with the bound of two summation orders of the same terms, n * eps * sum|term| per channel, derived from the data.
    dg1b, db1b = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    ops.layernorm_bwd(h, dy, ga, be, dg1b, db1b, drop_p=p, drop_seed=seed, relu_input=True)
    assert torch.equal(dg1, dg1b) and torch.equal(db1, db1b), "stand-alone LayerNorm backward is not deterministic"
    dyk = (dy * keep / (1 - p)).double()
    xh = F.layer_norm(h.double(), (256,))
    n = B * L
    bound_g = order_bound((dyk * xh).abs().sum((0, 1)), n) + 1e-6
    bound_b = order_bound(dyk.abs().sum((0, 1)), n) + 1e-6
    assert bool(((dg1 - dg2).abs().double() <= bound_g).all()), float(((dg1 - dg2).abs().double() / bound_g).max())
    assert bool(((db1 - db2).abs().double() <= bound_b).all()), float(((db1 - db2).abs().double() / bound_b).max())
    assert float(bound_g.max()) < 0.2 and float(bound_b.max()) < 0.2          # the bound is far below one row's term


def test_fused_predictor_equals_separate_nodes(dev, ref_state_dict):
    """StylePredictor under autograd: the two-node-per-predictor tape (rt.fused_predictor) gives the outputs and gradients
    of the node-per-op tape."""
    from styler_amd import rt
    from styler_amd.modules import StylePredictor
    g = torch.Generator().manual_seed(22)
    x0 = torch.randn(4, 61, 256, generator=g)
    lens = torch.tensor([61, 40, 17, 55])
    go = torch.randn(4, 61, generator=g).to(dev)
    sd = {k[len("style_modeling.pitch_predictor."):]: v for k, v in ref_state_dict.items()
          if k.startswith("style_modeling.pitch_predictor.")}
    res = []
    keep = rt.fused_predictor, rt.disable_dropout
    try:
        rt.disable_dropout = True
        for fused in (True, False):
            rt.fused_predictor = fused
            m = StylePredictor()
            m.load_state_dict(sd)
            m = m.to(dev).train()
            x = x0.to(dev).requires_grad_(True)
            out = m(x, lens.to(dev))
            out.backward(go)
            res.append((out.detach(), x.grad, {k: v.grad.clone() for k, v in m.named_parameters()}))
    finally:
        rt.fused_predictor, rt.disable_dropout = keep
    assert torch.equal(res[0][0], res[1][0])
    check(res[0][1], res[1][1], 1e-5, "dx")                   # elementwise chain, same kernels: max err / max|ref|
    for k in res[0][2]:
        # sums over the 4 * 61 rows in two orders: bound relative to the tensor's largest entry (the terms of every entry
        # of one tensor have the same scale), 244 rows * 2^-24 * ~10 for the cancellation within a sum
        check(res[0][2][k], res[1][2][k], 2e-4, k)


def test_graph_cache_two_buckets_small_then_large(dev, ref_state_dict):
    """Round-2 advisor finding: several captured steps on ONE TrainState.  The first graph is captured for a small batch
    shape; the second, larger shape's warm-up needs a bigger split-K arena / zero slab -- if the graphs shared those
    buffers the first graph would afterwards memset, write partials and read descriptor tables in freed memory.  Every
    graphed step owns its arena and slab: the two graphs, replayed alternately, must walk the trajectory of eager steps on
    the same batch sequence (dropout off), and the eager steps in between (shared arena) must not disturb them."""
    from closed_form import make_batch
    from styler_amd import STYLER, rt
    from styler_amd.training import GraphedStepCache, TrainState, train_step
    small = {k: v.to(dev) for k, v in make_batch(3, 10, 20, 2, 6, seed=41).items()}
    large = {k: v.to(dev) for k, v in make_batch(6, 30, 50, 4, 12, seed=42).items()}
    seq = [small, large, small, large, large, small]
    rt.disable_dropout = True
    try:
        finals = []
        for mode in ("eager", "cache"):
            m = STYLER()
            m.load_state_dict(ref_state_dict)
            m = m.to(dev).train()
            st = TrainState(m)
            cache = GraphedStepCache(m, st, warmup=2) if mode == "cache" else None
            traj = []
            for i, b in enumerate(seq):
                losses, lr = cache(b) if cache is not None else train_step(m, st, b)
                traj.append(torch.stack([x.detach().float().reshape(()) for x in losses]).cpu())
                if cache is not None and i == 1:
                    # both graphs exist now; their scratch must be distinct allocations, none of them the shared state's
                    a, c = (s_.arena for s_ in cache.steps.values())
                    assert a is not c and a is not st.arena and a.buf.data_ptr() != c.buf.data_ptr()
                    assert a.frozen and c.frozen and st.overlap_allreduce
            if cache is not None:
                assert (cache.misses, cache.hits) == (2, 4)
            torch.cuda.synchronize()
            finals.append((torch.stack(traj), st.flat_p.clone(), st.n_current_steps))
        (l_e, p_e, n_e), (l_c, p_c, n_c) = finals
        assert n_e == n_c == len(seq)
        assert torch.isfinite(l_c).all()
        assert float((l_e - l_c).abs().max()) <= 2e-4 * max(1.0, float(l_e.abs().max())), (l_e, l_c)
        assert float((p_e - p_c).abs().max()) <= 1e-4
    finally:
        rt.disable_dropout = False


def test_graph_cache_prefetch_has_no_side_effects(dev, ref_state_dict):
    """GraphedStepCache.prefetch captures shapes on synthetic batches before the loop (what several ranks do together so
    that no rank stalls its peers later): nothing trains, and a real batch of a prefetched shape is a cache hit whose step
    equals the eager step."""
    from closed_form import make_batch
    from styler_amd import STYLER, rt
    from styler_amd.training import GraphedStepCache, TrainState, synthetic_batch, train_step
    b = {k: v.to(dev) for k, v in make_batch(3, 10, 20, 2, 6, seed=51).items()}
    key = GraphedStepCache.key(b)
    sb = synthetic_batch(*key)
    assert {k: (tuple(v.shape), v.dtype) for k, v in sb.items()} == {k: (tuple(v.shape), v.dtype) for k, v in b.items()}
    rt.disable_dropout = True
    try:
        outs = []
        for mode in ("eager", "prefetched"):
            m = STYLER()
            m.load_state_dict(ref_state_dict)
            m = m.to(dev).train()
            st = TrainState(m)
            if mode == "prefetched":
                cache = GraphedStepCache(m, st, warmup=2)
                snap = (st.flat_p.clone(), [x.clone() for x in m.buffers()])
                cache.prefetch([key, (4, 24, 96)])
                assert cache.prefetched == 2 and st.n_current_steps == 0 and torch.equal(st.flat_p, snap[0])
                assert all(torch.equal(x, y) for x, y in zip(m.buffers(), snap[1]))
                losses, lr = cache(b)
                assert (cache.hits, cache.misses) == (1, 0)
            else:
                losses, lr = train_step(m, st, b)
            torch.cuda.synchronize()
            outs.append((torch.stack([x.detach().float().reshape(()) for x in losses]).cpu(), st.flat_p.clone()))
        assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-4 * max(1.0, float(outs[0][0].abs().max()))
        assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-5
    finally:
        rt.disable_dropout = False
