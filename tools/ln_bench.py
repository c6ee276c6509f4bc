#!/usr/bin/env python
"""Micro-benchmark of styler_layernorm_bwd / styler_add_layernorm on the decoder shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
dev = torch.device("cuda")
rows = 21168
x = torch.randn(1, rows, 256, device=dev); dy = torch.randn_like(x)
g = torch.randn(256, device=dev); b = torch.randn(256, device=dev)
dg = torch.zeros(256, device=dev); db = torch.zeros(256, device=dev)
for valid in (rows, 13530):
    lens = torch.tensor([valid], device=dev)
    for name, fn in (("ln_bwd", lambda: ops.layernorm_bwd(x, dy, g, b, dg, db, lens=lens)),
                     ("ln_bwd_drop", lambda: ops.layernorm_bwd(x, dy, g, b, dg, db, lens=lens, in_drop_p=0.2, in_drop_seed=7)),
                     ("ln_fwd", lambda: ops.add_layernorm(x, g, b, res=dy, lens=lens))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:12s} valid={valid:6d} {e0.elapsed_time(e1) * 20:8.1f} us")
