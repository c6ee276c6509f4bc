#!/usr/bin/env python
"""Probe of styler_layernorm_bwd alone (no slab clear, no fold): time per launch at the decoder's row count for the block cap
given by STYLER_LNBWD_BLOCKS (read once per process), one gradient slot per block.  Usage: STYLER_LNBWD_BLOCKS=1024 python tools/ln_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
from styler_amd._lib import lib

dev = torch.device("cuda")
rows = 27060
blocks = int(os.environ.get("STYLER_LNBWD_BLOCKS", "256"))
gam, bet = torch.randn(256, device=dev), torch.randn(256, device=dev)
slots = torch.zeros(2, max(blocks, 256), 256, device=dev)


def t(fn, n=200):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for dt in (torch.float32, torch.bfloat16):
    s = torch.randn(1, rows, 256, device=dev).to(dt)
    dy = torch.randn(1, rows, 256, device=dev).to(dt)
    dx, dxd = torch.empty_like(s), torch.empty_like(s)
    io = 0 if dt == torch.float32 else (2 | 4 | 8 | 16)
    for p in (0.0, 0.2):
        def call():
            rc = lib.styler_layernorm_bwd(s.data_ptr(), 256, dy.data_ptr(), 256, gam.data_ptr(), bet.data_ptr(), dx.data_ptr(), 256,
                                          slots[0].data_ptr(), slots[1].data_ptr(), None, None, None, None, 1, rows, 256, None,
                                          0.0, 0, float(p), 5, dxd.data_ptr() if p > 0 else None, 256, max(blocks, 256), io,
                                          torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        print(f"blocks<={blocks} dtype={str(dt)[6:]} in_drop_p={p}: {t(call):.1f} us", flush=True)

# the packed decoder launch: row capacity 42 336 (2 B T), 27 060 valid rows, lengths given (one item)
cap = 42336
lens = torch.tensor([rows], device=dev, dtype=torch.int64)
s = torch.randn(1, cap, 256, device=dev).bfloat16()
dy = torch.randn(1, cap, 256, device=dev).bfloat16()
dx, dxd = torch.empty_like(s), torch.empty_like(s)


def call_packed():
    rc = lib.styler_layernorm_bwd(s.data_ptr(), 256, dy.data_ptr(), 256, gam.data_ptr(), bet.data_ptr(), dx.data_ptr(), 256,
                                  slots[0].data_ptr(), slots[1].data_ptr(), None, None, None, None, 1, cap, 256, lens.data_ptr(),
                                  0.0, 0, 0.2, 5, dxd.data_ptr(), 256, max(blocks, 256), 2 | 4 | 8 | 16,
                                  torch.cuda.current_stream().cuda_stream)
    assert rc == 0


print(f"packed capacity {cap} rows, {rows} valid, bf16, in_drop_p=0.2: {t(call_packed):.1f} us", flush=True)
