#!/usr/bin/env python
"""A/B of the 256 x 256 LDS-DMA GEMM engine (csrc/gemm256.hip) against the 128 x 128 engine on the shapes of the step and of
config 4, interleaved rounds in ONE process (bf16 activations in, bf16 or fp32 out, uniform random operands).

    python tools/gemm256_bench.py [rounds]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops

SHAPES = [  # name, B, L, cin, n, kw, y_bf16
    ("p_ffn_w1_k9 (C3 decoder, 424 tiles)", 1, 27060, 256, 1024, 9, True),
    ("s_postnet_512_k5 (C3, 332 tiles)", 96, 441, 512, 512, 5, False),
    ("s_aenc_256_k5 (C3, 166 tiles)", 96, 441, 256, 256, 5, False),
    ("s_postnet_512_k5 bf16 out", 96, 441, 512, 512, 5, True),
    ("s_aenc_256_k5 bf16 out", 96, 441, 256, 256, 5, True),
    ("p_dx_w1_k9 (C3, 106 tiles, K=9216)", 1, 27060, 1024, 256, 9, False),
    ("p_ffn_w2_k1 (C3, 106 tiles, K=1024)", 1, 27060, 1024, 256, 1, False),
    ("q_ffn_2x (848 tiles)", 1, 54120, 256, 1024, 9, True),
    ("c4_ffn_w1_k9 (8000 tiles)", 1, 512000, 256, 1024, 9, True),
    ("c4_postnet_512_k5 (4000 tiles)", 256, 2000, 512, 512, 5, False),
    ("c4_aenc_256_k5 (1000 tiles)", 128, 2000, 256, 256, 5, False),
    ("overhead probe: 1 K step, 8000 tiles", 1, 512000, 64, 1024, 1, True),
    ("square 4096 (256 tiles)", 1, 4096, 4096, 4096, 1, False),
    ("square 8192 x 4096 (512 tiles)", 1, 8192, 4096, 4096, 1, False),
]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda")
    for name, B, L, cin, n, kw, y16 in SHAPES:
        x = (torch.rand(B, L, cin, device=dev) * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand(n, kw * cin, device=dev) * 2 - 1) / (kw * cin) ** 0.5).to(torch.bfloat16)
        b = torch.randn(n, device=dev)
        y = torch.empty(B, L, n, device=dev, dtype=torch.bfloat16 if y16 else torch.float32)
        fl = 2.0 * B * L * n * kw * cin
        act = ops.ACT_NONE if "dx" in name else ops.ACT_RELU     # dX launches have a plain epilogue (and may run split-K)
        res = {0: [], 1: [], 2: []}
        prev = ops.gemm256_config(-1, -1)
        prev_h = ops.gemm256_height(-1)
        try:
            for r in range(rounds):
                for eng in (0, 1, 2):                   # 128 x 128 | 256 x 256 | 192 x 256 (round 6)
                    ops.gemm256_config(min(eng, 1), -1, split=1, take_all=1)
                    ops.gemm256_height(3 if eng == 2 else 4)
                    for _ in range(2):
                        ops.conv_gemm(x, w, b, kw=kw, act=act, prec=ops.PREC_BF16, out=y)
                    iters = max(3, min(20, int(4e-3 / (fl / 0.9e15))))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        ops.conv_gemm(x, w, b, kw=kw, act=act, prec=ops.PREC_BF16, out=y)
                    e1.record()
                    torch.cuda.synchronize()
                    res[eng].append(e0.elapsed_time(e1) * 1e3 / iters)
        finally:
            ops.gemm256_config(*prev)
            ops.gemm256_height(*prev_h)
        m = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print(f"{name:42s} M={B * L:7d} N={n:5d} K={kw * cin:5d}  128^2: {m[0]:9.1f} us {fl / m[0] / 1e6:7.1f} TF/s   "
              f"256^2: {m[1]:9.1f} us {fl / m[1] / 1e6:7.1f} TF/s x{m[0] / m[1]:.3f}   192x256: {m[2]:9.1f} us "
              f"{fl / m[2] / 1e6:7.1f} TF/s x{m[0] / m[2]:.3f}", flush=True)
        del x, w, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
