#!/usr/bin/env python
"""Timing of the BatchNorm / LayerNorm / GroupNorm kernels at the step's shapes (HIP events around single ops; run under
rocprofv3 --kernel-trace for per-kernel numbers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops

dev = torch.device("cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


rows, C = 96 * 441, 512
x = torch.randn(rows // 441, 441, C, device=dev)
dy = torch.randn_like(x)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
y, mean, rstd = ops.batchnorm_train(x, g, b, rm, rv, ops.ACT_TANH, drop_p=0.5, drop_seed=7, segs=2)
mb = x.numel() * 4 / 1e6
print(f"BN train  [{rows}, {C}] segs=2 tanh drop .5: {t(lambda: ops.batchnorm_train(x, g, b, rm, rv, ops.ACT_TANH, drop_p=0.5, drop_seed=7, segs=2)):7.1f} us  (x = {mb:.1f} MB; stats + fold + finalize + apply)")
print(f"BN bwd    same                              : {t(lambda: ops.batchnorm_bwd(x, None, dy, g, mean, rstd, dg, db, ops.ACT_TANH, beta=b, drop_p=0.5, drop_seed=7, segs=2)):7.1f} us  (stats + fold + apply)")
xs = torch.randn(1, 27060, 256, device=dev); dys = torch.randn_like(xs)
g2, b2 = torch.ones(256, device=dev), torch.zeros(256, device=dev)
dg2, db2 = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
print(f"LN bwd    [27060, 256]                      : {t(lambda: ops.layernorm_bwd(xs, dys, g2, b2, dg2, db2)):7.1f} us  ({xs.numel() * 12 / 1e6:.1f} MB)")
# GroupNorm + ReLU at the AudioEncoder shape (96 items x 441 frames x 320 channels; bf16 output / gradient as in the step)
Cg = 320
xg = torch.randn(96, 441, Cg, device=dev)
gg, bg = torch.ones(Cg, device=dev), torch.zeros(Cg, device=dev)
st = torch.empty(96, Cg // 16, 2, device=dev)
yg = torch.empty_like(xg, dtype=torch.bfloat16)
dyg = torch.randn_like(xg).to(torch.bfloat16)
dgg, dbg = torch.zeros(Cg, device=dev), torch.zeros(Cg, device=dev)
mode = "two kernels" if os.environ.get("STYLER_GN_FUSED") == "0" else "single pass"
print(f"GN fwd    [96, 441, {Cg}] bf16 out ({mode}): {t(lambda: ops.groupnorm_relu(xg, gg, bg, out=yg, stats=st)):7.1f} us  (x = {xg.numel() * 4 / 1e6:.1f} MB)")
print(f"GN bwd    same, bf16 dy / dx ({mode})      : {t(lambda: ops.groupnorm_relu_bwd(xg, dyg, gg, bg, st, dgg, dbg, dx_bf16=True)):7.1f} us")
