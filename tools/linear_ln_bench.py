#!/usr/bin/env python
"""styler_linear_ln (one launch) vs styler_conv_gemm + styler_add_layernorm (two) at the decoder / encoder shapes of the
benched step; cache-cold-ish (rotating operand sets)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
dev = torch.device("cuda")
NSET = 6


def bench(fn, n=30):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, B, L, K, valid, s16 in (("dec_fc", 1, 42336, 256, 27060, True), ("dec_w2", 1, 42336, 1024, 27060, True),
                                   ("enc_fc", 48, 60, 256, None, False), ("enc_w2", 48, 60, 1024, None, False),
                                   ("c4_fc", 1, 256000, 256, 192000, True), ("c4_w2", 1, 256000, 1024, 192000, True)):
    sets = []
    for _ in range(NSET):
        a = torch.randn(B, L, K, device=dev).to(torch.bfloat16)
        res = torch.randn(B, L, 256, device=dev)
        if s16:
            res = res.to(torch.bfloat16)
        sets.append((a, res, torch.empty_like(res), torch.empty_like(res)))
    w = (torch.randn(256, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias, ga, be = torch.randn(256, device=dev), torch.randn(256, device=dev), torch.randn(256, device=dev)
    lens = torch.tensor([valid], device=dev) if valid else torch.randint(20, 61, (B,), device=dev)
    o = torch.empty(B, L, 256, device=dev)

    def two(i):
        a, res, s, y = sets[i % NSET]
        ops.conv_gemm(a, w, bias, n=256, prec=ops.PREC_BF16, out=o)
        ops.add_layernorm(o, ga, be, res=res, lens=lens, in_drop_p=0.1, in_drop_seed=5, sum_out=s, out=y)

    def gemm_only(i):
        a, res, s, y = sets[i % NSET]
        ops.conv_gemm(a, w, bias, n=256, prec=ops.PREC_BF16, out=o)

    def fused(i):
        a, res, s, y = sets[i % NSET]
        ops.linear_ln(a, w, bias, res, ga, be, lens=lens, drop_p=0.1, drop_seed=5, sum_out=s, out=y)

    rows = valid if valid else int(lens.sum())
    byts = rows * (K * 2 + 256 * (2 if s16 else 4) * 3)
    t2, tg, t1 = bench(two), bench(gemm_only), bench(fused)
    print(f"{name:8s} rows={rows:7d} K={K:5d}: gemm {tg:7.1f} us, gemm + add_layernorm {t2:7.1f} us, linear_ln {t1:7.1f} us "
          f"({byts / t1 * 1e-6:5.2f} TB/s algorithmic, {2.0 * rows * 256 * K / t1 * 1e-6:6.1f} TFLOP/s)")
