#!/usr/bin/env python
"""Run-to-run reproducibility of the GroupNorm kernels (both forms): max |difference| of repeated calls on one input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(12)
for L in (137, 600, 1100):
    B, C = 6, 320
    x = torch.randn(B, L, C, generator=g).to(dev)
    dy = torch.randn(B, L, C, generator=g).to(dev)
    w = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev); b = (0.1 * torch.randn(C, generator=g)).to(dev)
    st = torch.empty(B, C // 16, 2, device=dev)
    y0 = ops.groupnorm_relu(x, w, b, out=torch.empty_like(x), stats=st).clone()
    st0 = st.clone()
    fwd_bad = bwd_bad = 0; fwd_max = bwd_max = 0.0
    dx0 = None
    for i in range(30):
        st2 = torch.empty_like(st)
        y = ops.groupnorm_relu(x, w, b, out=torch.empty_like(x), stats=st2)
        fwd_bad += int((y != y0).sum()) + int((st2 != st0).sum()); fwd_max = max(fwd_max, float((y - y0).abs().max()))
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx = ops.groupnorm_relu_bwd(x, dy, w, b, st0, dg, db)
        if dx0 is None: dx0 = dx.clone()
        bwd_bad += int((dx != dx0).sum()); bwd_max = max(bwd_max, float((dx - dx0).abs().max()))
    print(f"L={L}: fwd mismatching elements {fwd_bad} (max diff {fwd_max:.3e}); bwd mismatching {bwd_bad} (max diff {bwd_max:.3e}; |dx| max {float(dx0.abs().max()):.3f})")
# the three storage variants of the backward against each other (what tests/test_91_bf16_acts.py asserts), repeated
for L in (137, 600):
    B, C = 6, 320
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, L, C, generator=g).to(dev)
    dy = torch.randn(B, L, C, generator=g).to(dev)
    w = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev); b = (0.1 * torch.randn(C, generator=g)).to(dev)
    st = torch.empty(B, C // 16, 2, device=dev)
    ops.groupnorm_relu(x, w, b, out=torch.empty_like(x), stats=st)
    dy16 = dy.to(torch.bfloat16)
    bad01 = bad02 = 0; m01 = 0.0
    for i in range(30):
        res = []
        for dyv, o16 in ((dy16.float(), False), (dy16, False), (dy16, True)):
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            res.append(ops.groupnorm_relu_bwd(x, dyv, w, b, st, dg, db, dx_bf16=o16))
        bad01 += int((res[0] != res[1]).sum()); m01 = max(m01, float((res[0] - res[1]).abs().max()))
        bad02 += int((res[2] != res[0].to(torch.bfloat16)).sum())
    print(f"L={L}: dy fp32 vs dy bf16 -> {bad01} differing elements (max {m01:.3e}); bf16 dx vs rounded fp32 dx -> {bad02}")
