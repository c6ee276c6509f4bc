#!/usr/bin/env python
"""Weight-gradient engine A/B on the training step's shapes, both operands resident as bf16: the register-staged kernel
(wgrad_tr_kernel) against the LDS-DMA ring (wgrad_dma_kernel), interleaved rounds in one process.  Every shape is first
checked: the two kernels' split-K partial tiles must be bit-equal and the reduced gradient must match an fp64 matmul.
usage: wgrad_bench.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
from styler_amd._lib import lib

SHAPES = [  # name, B, L, cin, n, kw   (dw[n, cin, kw] += dz^T x; rows = B * L)
    ("dec_ffn_w1_k9", 61, 441, 256, 1024, 9), ("postnet_512_k5", 96, 441, 512, 512, 5), ("aenc_256_k5", 96, 441, 256, 256, 5),
    ("aenc_320_k5", 96, 441, 320, 320, 5), ("postnet_out_k5", 96, 441, 512, 80, 5), ("dec_ffn_w2_k1", 1, 27060, 1024, 256, 1),
    ("dec_q_k1", 1, 27060, 256, 256, 1), ("enc_ffn_w1_k9", 48, 60, 256, 1024, 9), ("ragged_k9", 7, 333, 256, 128, 9),
]


def run(dz, x, n, cin, kw, ws, db, mode, nst128=2):
    lib.styler_wgrad_dma_config(mode, nst128)
    B, L = dz.shape[:2]
    strides = (cin * kw, kw, 1) if kw > 1 else (cin, 1, 0)
    ops._chk(lib.styler_wgrad(dz.data_ptr(), dz.stride(1), x.data_ptr(), x.stride(1), ws.data_ptr(), db.data_ptr(), None,
                              *strides, B, L, n, cin, kw, kw // 2, ops.PREC_BF16, ws.data_ptr(), 1, 3,
                              torch.cuda.current_stream().cuda_stream), "styler_wgrad")


def plan(B, L, n, cin, kw, mode):
    lib.styler_wgrad_dma_config(mode, 0)
    nb = int(lib.styler_wgrad_workspace_bytes_io(B, L, n, cin, kw, kw // 2, ops.PREC_BF16, 3))
    sp = int(lib.styler_wgrad_splits_io(B, L, n, cin, kw, kw // 2, ops.PREC_BF16, 3))
    return nb, sp


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(4)
    print(f"{'shape':18s} {'rows':>6s} | reg: splits us TF/s | dma (4 waves): us TF/s | dma2 (2 K groups): splits us TF/s | reg/dma2"
          f"      (median of {rounds} rounds x 10 launches)")
    for name, B, L, cin, n, kw in SHAPES:
        dz = (torch.randn(B, L, n, generator=g)).to(dev).to(torch.bfloat16)
        x = (torch.randn(B, L, cin, generator=g)).to(dev).to(torch.bfloat16)
        variants = [("reg", 0), ("dma", 1), ("dma2", 2)]
        outs, plans = {}, {}
        for tag, mode in variants:
            nb, sp = plans[tag] = plan(B, L, n, cin, kw, mode)
            ws = torch.full((nb // 4,), float("nan"), device=dev)
            db = torch.zeros(n, device=dev)
            run(dz, x, n, cin, kw, ws, db, mode)
            torch.cuda.synchronize()
            outs[tag] = (ws, db)
        same = torch.equal(outs["dma"][0], outs["reg"][0])
        if not same:
            print(f"{name}: dma (mode 1) MISMATCH: partial tiles differ from the register-staged kernel's in "
                  f"{int((outs['dma'][0] != outs['reg'][0]).sum())} floats")
        red = {t: outs[t][0].view(plans[t][1], n, kw, cin).double().sum(0) for t in outs}
        e2 = float((red["dma2"] - red["reg"]).abs().max() / red["reg"].abs().max())
        b2 = float((outs["dma2"][1] - outs["reg"][1]).abs().max() / (outs["reg"][1].abs().max() + 1e-9))
        b1 = float((outs["dma"][1] - outs["reg"][1]).abs().max() / (outs["reg"][1].abs().max() + 1e-9))
        if e2 > 2e-6 or b2 > 1e-4 or b1 > 1e-4:
            print(f"{name}: dma2 reduced gradient differs from reg by {e2:.2e} (rel), bias {b2:.2e} / {b1:.2e}")
        if B * L <= 4096:                  # reduced gradient vs fp64 (small shapes only: the CPU matmul)
            dzc, xc = dz.double().cpu(), x.double().cpu()
            xp = torch.nn.functional.pad(xc, (0, 0, kw // 2, kw // 2))
            ref = torch.stack([torch.einsum("btn,btc->nc", dzc, xp[:, j:j + L]) for j in range(kw)], 1)
            for t in ("dma", "dma2"):
                err = float((red[t].cpu() - ref).abs().max() / ref.abs().max())
                dberr = float((outs[t][1].double().cpu() - dzc.sum((0, 1))).abs().max() / dzc.sum((0, 1)).abs().max())
                print(f"{name}: {t} vs fp64 rel err dw {err:.2e} db {dberr:.2e}")
        times = {tag: [] for tag, _ in variants}
        wss = {t: torch.empty(plans[t][0] // 4, device=dev) for t in plans}
        db = torch.zeros(n, device=dev)
        for _ in range(rounds):
            for tag, mode in variants:
                run(dz, x, n, cin, kw, wss[tag], db, mode)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(dz, x, n, cin, kw, wss[tag], db, mode)
                e1.record()
                torch.cuda.synchronize()
                times[tag].append(e0.elapsed_time(e1) * 100.0)
        fl = 2.0 * B * L * n * kw * cin
        med = {t: sorted(v)[len(v) // 2] for t, v in times.items()}
        print(f"{name:18s} {B * L:6d} | {plans['reg'][1]:4d} {med['reg']:8.1f} {fl / med['reg'] / 1e6:6.0f} | {med['dma']:8.1f} "
              f"{fl / med['dma'] / 1e6:6.0f} | {plans['dma2'][1]:4d} {med['dma2']:8.1f} {fl / med['dma2'] / 1e6:6.0f} | "
              f"{med['reg'] / med['dma2']:.3f}", flush=True)
    lib.styler_wgrad_dma_config(2, 2)


if __name__ == "__main__":
    main()
