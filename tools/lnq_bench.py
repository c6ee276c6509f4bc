#!/usr/bin/env python
"""LayerNorm backward on the decoder's packed bf16 stream (x, dy -> dx, dx through the dropout mask), the kernel alone (direct
C-ABI calls, per-block slots preallocated): the sixteen-lanes-per-row kernel (STYLER_LNBWD_Q16, default) against the wave-per-row
kernel, cache-cold (rotating operand sets)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
lib = ops.lib
dev = torch.device("cuda")
rows, valid = 42336, 27060
NS = 12
bf = torch.bfloat16
xs = [torch.randn(1, rows, 256, device=dev).to(bf) for _ in range(NS)]
dys = [torch.randn(1, rows, 256, device=dev).to(bf) for _ in range(NS)]
dxs = [torch.empty(1, rows, 256, device=dev, dtype=bf) for _ in range(NS)]
dxds = [torch.empty(1, rows, 256, device=dev, dtype=bf) for _ in range(NS)]
g = torch.randn(256, device=dev); b = torch.randn(256, device=dev)
slots = torch.zeros(2, 256, 256, device=dev)
lens = torch.tensor([valid], device=dev)
for name, drop in (("ln_bwd bf16 (dx)", 0.0), ("ln_bwd bf16 (dx, dx_drop)", 0.1)):
    def call(i):
        io = 2 | 4 | 8 | (16 if drop else 0)
        ops._chk(lib.styler_layernorm_bwd(xs[i].data_ptr(), 256, dys[i].data_ptr(), 256, g.data_ptr(), b.data_ptr(), dxs[i].data_ptr(),
                                          256, slots[0].data_ptr(), slots[1].data_ptr(), None, None, None, None, 1, rows, 256,
                                          lens.data_ptr(), 0.0, 0, drop, 7, dxds[i].data_ptr() if drop else None, 256, 256, io,
                                          ops._stream()), "ln_bwd")
    for i in range(3):
        call(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(48):
        call(i % NS)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 48
    nb = valid * 512 * (4 if drop else 3)
    print(f"{name:28s} {us:7.1f} us  {nb / us / 1e6:5.2f} TB/s on the valid rows ({100 * nb / us / 8e6:4.1f} % of 8 TB/s)")
