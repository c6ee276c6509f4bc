#!/usr/bin/env python
"""The config-4 decoder attention forward (B=256, T=2000, bf16 qkv / output), a few launches: the target of PMC passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
B, L = 256, 2000
lens = torch.randint(1000, L + 1, (B,), generator=g).to(dev)
q16 = torch.randn(B, L, 768, device=dev).to(torch.bfloat16)
lse = torch.empty(B, 4, L, device=dev)
for _ in range(4):
    ops.attention_fwd(q16, lens, lse=lse, prec=ops.PREC_BF16, out_bf16=True)
torch.cuda.synchronize()
