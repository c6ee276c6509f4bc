#!/usr/bin/env python
"""Micro-benchmark of the bf16 attention kernels at the C2 decoder shape (B=48, L=441, VCTK-like lengths)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
dev = torch.device("cuda")
B, L = 48, 441
g = torch.Generator().manual_seed(0)
lens = torch.randint(150, L + 1, (B,), generator=g).to(dev)
qkv = torch.randn(B, L, 768, device=dev)
lse = torch.empty(B, 4, L, device=dev)
dout = torch.randn(B, L, 256, device=dev)
out = ops.attention_fwd(qkv, lens, lse=lse, prec=ops.PREC_BF16)
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("mean len", float(lens.float().mean()))
print(f"fwd {t(lambda: ops.attention_fwd(qkv, lens, lse=lse, prec=ops.PREC_BF16)):7.1f} us")
print(f"bwd {t(lambda: ops.attention_bwd(qkv, out, dout, lse, lens, prec=ops.PREC_BF16)):7.1f} us (dq + dkv)")
