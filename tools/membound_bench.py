#!/usr/bin/env python
"""Achieved HBM-side rate of the memory-bound kernels north_star names (LengthRegulator, LayerNorm, bucketise+embed) at the
C2 / C3 shape and at C4: algorithmic bytes / HIP-event time per launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops

dev = torch.device("cuda")


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def report(name, nbytes, us):
    print(f"{name:58s} {nbytes / 1e6:9.1f} MB {us:8.1f} us {nbytes / us / 1e6:7.2f} TB/s  {100 * nbytes / us / 8e6:5.1f} % of 8 TB/s", flush=True)


for tag, B, S, T in (("C2/C3", 48, 60, 441), ("C4", 128, 300, 2000)):
    g = torch.Generator().manual_seed(0)
    d = torch.randint(2, 14, (B, S), generator=g)
    if tag == "C4":
        d = torch.randint(5, 9, (B, S), generator=g)
    csum, mel_len, _ = ops.duration_scan(B, S, dev, dur=d.to(dev))
    x = torch.randn(B, S, 1280, device=dev)
    us = timeit(lambda: ops.length_regulate(x, csum, T))
    report(f"length_regulate {tag} [B={B},S={S}->T={T},1280]", B * T * 5120 + B * S * 5120, us)
    dy = torch.randn(B, T, 1280, device=dev)
    us = timeit(lambda: ops.length_regulate_bwd(dy, csum, S))
    report(f"length_regulate_bwd {tag}", B * T * 5120 + B * S * 5120, us)
    rows = B * T
    a = torch.randn(1, rows, 256, device=dev)
    r = torch.randn_like(a)
    gam, bet = torch.randn(256, device=dev), torch.randn(256, device=dev)
    s_out = torch.empty_like(a)
    us = timeit(lambda: ops.add_layernorm(a, gam, bet, res=r, sum_out=s_out))
    report(f"add_layernorm (x + res -> y, sum) {tag} rows={rows}", rows * 4096, us)
    us = timeit(lambda: ops.add_layernorm(a, gam, bet, res=r))
    report(f"add_layernorm (x + res -> y) {tag} rows={rows}", rows * 3072, us)
    dg, db = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    us = timeit(lambda: ops.layernorm_bwd(a, r, gam, bet, dg, db))
    report(f"layernorm_bwd (s, dy -> dx) {tag} rows={rows}", rows * 3072, us)
    text, spk = torch.randn(B, T, 256, device=dev), torch.randn(B, T, 256, device=dev)
    p, e = torch.rand(B, T, device=dev) * 300 + 80, torch.rand(B, T, device=dev) * 100
    pb = torch.exp(torch.linspace(4.26, 6.68, 255)).to(dev)
    eb = torch.linspace(0.1, 525.43, 255).to(dev)
    pe, ee = torch.randn(256, 256, device=dev), torch.randn(256, 256, device=dev)
    us = timeit(lambda: ops.bucket_embed_add(text, spk, p, 1.0, e, 1.0, pb, eb, pe, ee))
    report(f"bucket_embed_add {tag} frames={rows}", rows * (2048 + 2048 + 1024 + 8), us)
