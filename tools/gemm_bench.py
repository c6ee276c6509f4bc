#!/usr/bin/env python
"""Micro-benchmark of styler_conv_gemm on the shapes of the C2 forward (per-shape TFLOP/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops

SHAPES = [  # name, B, L, cin, n, kw
    ("ffn_w1_k9", 48, 441, 256, 1024, 9), ("ffn_w2_k1", 48, 441, 1024, 256, 1), ("qkv", 48, 441, 256, 768, 1),
    ("attn_fc", 48, 441, 256, 256, 1), ("postnet_512_k5", 48, 441, 512, 512, 5), ("postnet_in", 48, 441, 80, 512, 5),
    ("postnet_out", 48, 441, 512, 80, 5), ("pred_k3", 48, 441, 256, 256, 3), ("aenc_320_k5", 48, 441, 320, 320, 5),
    ("enc_w1_k9", 48, 60, 256, 1024, 9), ("mel_linear", 48, 441, 256, 80, 1), ("big_w1", 128, 2000, 256, 1024, 9),
    # the paired decode's row count (2 x 13 530 valid frames)
    ("p_ffn_w1_k9", 1, 27060, 256, 1024, 9), ("p_ffn_w2_k1", 1, 27060, 1024, 256, 1), ("p_qkv", 1, 27060, 256, 768, 1),
    ("p_attn_fc", 1, 27060, 256, 256, 1), ("p_dx_w1_k9", 1, 27060, 1024, 256, 9), ("p_dx_qkv", 1, 27060, 768, 256, 1),
    # the stacked AudioEncoder / PostNet passes (2 x 48 items)
    ("s_aenc_320_k5", 96, 441, 320, 320, 5), ("s_aenc_256_k5", 96, 441, 256, 256, 5), ("s_postnet_512_k5", 96, 441, 512, 512, 5),
    ("s_postnet_out", 96, 441, 512, 80, 5),
    # round quantisation probes: 768 / 993 / 1536 tiles of 128 x 128 on the AudioEncoder shape, 1536 / 1696 / 2304 on the FFN's
    ("q_aenc_768", 1, 32768, 320, 320, 5), ("q_aenc_993", 1, 42336, 320, 320, 5), ("q_aenc_1536", 1, 65536, 320, 320, 5),
    ("q_ffn_1536", 1, 24576, 256, 1024, 9), ("q_ffn_1696", 1, 27136, 256, 1024, 9), ("q_ffn_2304", 1, 36864, 256, 1024, 9),
]

WGRAD_SHAPES = [  # name, B, L, cin, n, kw  (dw[n, cin, kw] += dz^T x)
    ("w_ffn_w1_k9", 48, 441, 256, 1024, 9), ("w_ffn_w2_k1", 48, 441, 1024, 256, 1), ("w_qkv", 48, 441, 256, 768, 1),
    ("w_attn_fc", 48, 441, 256, 256, 1), ("w_postnet_k5", 48, 441, 512, 512, 5), ("w_pred_k3", 48, 441, 256, 256, 3),
    ("w_aenc_320_k5", 48, 441, 320, 320, 5), ("w_lstm_ih", 48, 60, 320, 640, 1), ("w_enc_qkv", 48, 60, 256, 768, 1),
    ("w_mel_linear", 48, 441, 256, 80, 1),
]


def wgrad_main():
    dev = torch.device("cuda")
    for name, B, L, cin, n, kw in WGRAD_SHAPES:
        dz = torch.randn(B, L, n, device=dev)
        x = torch.randn(B, L, cin, device=dev)
        dw = torch.zeros(n, cin, kw, device=dev) if kw > 1 else torch.zeros(n, cin, device=dev)
        db = torch.zeros(n, device=dev)
        for prec in ("bf16",):
            p = ops.PREC_BF16
            for _ in range(3):
                ops.wgrad(dz, x, dw, n, cin, kw=kw, db=db, prec=p)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 20
            e0.record()
            for _ in range(iters):
                ops.wgrad(dz, x, dw, n, cin, kw=kw, db=db, prec=p)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            fl = 2.0 * B * L * n * kw * cin
            sp = ops.lib.styler_wgrad_splits(B, L, n, cin, kw, kw // 2, p)
            print(f"{name:16s} {prec:5s} K={B*L:6d} n={n:5d} cin={cin:5d} kw={kw} splits={sp:4d} {us:9.1f} us  {fl/us/1e6:8.1f} TFLOP/s (incl. reduce)", flush=True)


def main():
    if "wgrad" in sys.argv[1:]:
        return wgrad_main()
    dev = torch.device("cuda")
    precs = [a for a in sys.argv[1:] if a in ("bf16", "fp32")] or ["bf16", "fp32"]
    only = [a for a in sys.argv[1:] if a not in ("bf16", "fp32", "io")]
    for name, B, L, cin, n, kw in SHAPES:
        if only and name not in only:
            continue
        x = torch.randn(B, L, cin, device=dev)
        w = torch.randn(n, kw * cin, device=dev) / (kw * cin) ** 0.5
        b = torch.randn(n, device=dev)
        for prec in precs:
            wk = ops.cast_bf16(w) if prec == "bf16" else w
            p = ops.PREC_BF16 if prec == "bf16" else ops.PREC_F32
            # storage variants of the activation operand / the output (bf16 mode only): fp32|bf16 x fp32|bf16
            ios = [(False, False)] + ([(True, False), (False, True), (True, True)] if prec == "bf16" and "io" in sys.argv else [])
            for x16, y16 in ios:
                xx = x.to(torch.bfloat16) if x16 else x
                y = torch.empty(B, L, n, device=dev, dtype=torch.bfloat16 if y16 else torch.float32)
                for _ in range(3):
                    ops.conv_gemm(xx, wk, b, kw=kw, prec=p, out=y)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = 20
                e0.record()
                for _ in range(iters):
                    ops.conv_gemm(xx, wk, b, kw=kw, prec=p, out=y)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / iters
                fl = 2.0 * B * L * n * kw * cin
                by = B * L * (cin * (2 if x16 else 4) + n * (2 if y16 else 4)) + n * kw * cin * 2
                tag = f"x{'16' if x16 else '32'}y{'16' if y16 else '32'}"
                print(f"{name:16s} {prec:5s} {tag} M={B*L:6d} N={n:5d} K={kw*cin:5d}  {us:9.1f} us  {fl/us/1e6:8.1f} TFLOP/s  "
                      f"{by/us/1e6:6.2f} TB/s", flush=True)

if __name__ == "__main__":
    main()
