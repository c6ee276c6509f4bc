#!/usr/bin/env python
"""PostNet BatchNorm (train mode, tanh, dropout 0.5, two segments) forward / backward at the step's shape, bf16 storage, cache-cold."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
dev = torch.device("cuda")
rows, C = 42336, 512
NS = 8
bf = torch.bfloat16
xs = [torch.randn(96, 441, C, device=dev).to(bf) for _ in range(NS)]
dys = [torch.randn(96, 441, C, device=dev).to(bf) for _ in range(NS)]
w = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)


def timeit(fn, n=40):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i % NS)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


outs = [None] * NS
def fwd(i):
    outs[i] = ops.batchnorm_train(xs[i], w, b, rm, rv, ops.ACT_TANH, drop_p=0.5, drop_seed=5, segs=2, out_bf16=True)
us = timeit(fwd)
print(f"batchnorm_train (colstats + finalize + apply)  {us:6.1f} us")
for i in range(NS):
    fwd(i)
def bwd(i):
    y, m, r = outs[i]
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.batchnorm_bwd(xs[i], None, dys[i], w, m, r, dg, db, ops.ACT_TANH, beta=b, drop_p=0.5, drop_seed=5, segs=2, dx_bf16=True)
try:
    us = timeit(bwd)
    print(f"batchnorm_bwd (colstats + fold + apply, incl. two torch.zeros)  {us:6.1f} us")
except Exception as e:
    print("bwd:", e)
