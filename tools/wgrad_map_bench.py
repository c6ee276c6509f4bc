#!/usr/bin/env python
"""Round 5 A/B of the bf16 weight-gradient launches on the step's shapes (both operands bf16-resident, LDS-DMA ring, mode 2):
  map0 / map1 : the former tile-major block map vs the XCD box map (styler_wgrad_tune knob 0) -- partial tiles must be bit-equal;
  tall        : k = 5 gradients on the 128 x 64 x 5 tile (knob 1) -- reduced gradient vs the 64 x 64 tile's, <= 2e-6 relative.
Interleaved rounds in one process; median of `rounds` x 10 launches.   usage: wgrad_map_bench.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
from styler_amd._lib import lib

SHAPES = [  # name, B, L, cin, n, kw
    ("dec_ffn_w1_k9", 61, 441, 256, 1024, 9), ("postnet_512_k5", 96, 441, 512, 512, 5), ("aenc_256_k5", 96, 441, 256, 256, 5),
    ("aenc_320_k5", 96, 441, 320, 320, 5), ("enc_ffn_w1_k9", 48, 60, 256, 1024, 9), ("dec_ffn_w2_k1", 1, 27060, 1024, 256, 1),
    ("ragged_k5_128", 7, 333, 72, 128, 5),
]


def run(dz, x, n, cin, kw, ws, db):
    B, L = dz.shape[:2]
    strides = (cin * kw, kw, 1) if kw > 1 else (cin, 1, 0)
    ops._chk(lib.styler_wgrad(dz.data_ptr(), dz.stride(1), x.data_ptr(), x.stride(1), ws.data_ptr(), db.data_ptr(), None,
                              *strides, B, L, n, cin, kw, kw // 2, ops.PREC_BF16, ws.data_ptr(), 1, 3,
                              torch.cuda.current_stream().cuda_stream), "styler_wgrad")


def setup(cfg):
    lib.styler_wgrad_tune(0, cfg[0]); lib.styler_wgrad_tune(1, cfg[1]); lib.styler_wgrad_tune(2, cfg[2] if len(cfg) > 2 else 0)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(4)
    lib.styler_wgrad_dma_config(2, 2)
    cfgs = {"map0": (0, 0, 0), "map1": (0, 0, 1), "tall": (0, 1, 0)}     # round-5 second series: "map1" column = ring of 4 stages (knob 2)
    print(f"{'shape':16s} {'rows':>6s} | map0: splits us TF/s | map1: us TF/s x | tall: splits us TF/s x   (median of {rounds} x 10 launches)")
    for name, B, L, cin, n, kw in SHAPES:
        dz = torch.randn(B, L, n, generator=g).to(dev).to(torch.bfloat16)
        x = torch.randn(B, L, cin, generator=g).to(dev).to(torch.bfloat16)
        outs, plans, wss = {}, {}, {}
        for tag, cfg in cfgs.items():
            setup(cfg)
            nb = int(lib.styler_wgrad_workspace_bytes_io(B, L, n, cin, kw, kw // 2, ops.PREC_BF16, 3))
            sp = int(lib.styler_wgrad_splits_io(B, L, n, cin, kw, kw // 2, ops.PREC_BF16, 3))
            plans[tag] = (nb, sp)
            ws = torch.full((nb // 4,), float("nan"), device=dev)
            db = torch.zeros(n, device=dev)
            run(dz, x, n, cin, kw, ws, db)
            torch.cuda.synchronize()
            outs[tag] = (ws.view(sp, n, kw, cin), db)
            wss[tag] = torch.empty(nb // 4, device=dev)
        if not torch.equal(outs["map0"][0], outs["map1"][0]):
            print(f"{name}: MISMATCH map0 vs map1 partial tiles ({int((outs['map0'][0] != outs['map1'][0]).sum())} floats)")
        r0 = outs["map0"][0].double().sum(0)
        rt = outs["tall"][0].double().sum(0)
        e = float((rt - r0).abs().max() / r0.abs().max())
        eb = float((outs["tall"][1] - outs["map0"][1]).abs().max() / (outs["map0"][1].abs().max() + 1e-9))
        if e > 2e-6 or eb > 1e-4:
            print(f"{name}: tall vs 64x64 reduced gradient rel {e:.2e}, bias {eb:.2e}")
        if B * L <= 4096:
            dzc, xc = dz.double().cpu(), x.double().cpu()
            xp = torch.nn.functional.pad(xc, (0, 0, kw // 2, kw // 2))
            ref = torch.stack([torch.einsum("btn,btc->nc", dzc, xp[:, j:j + L]) for j in range(kw)], 1)
            for t in cfgs:
                err = float((outs[t][0].double().sum(0).cpu() - ref).abs().max() / ref.abs().max())
                print(f"{name}: {t} vs fp64 rel err dw {err:.2e}")
        times = {t: [] for t in cfgs}
        db = torch.zeros(n, device=dev)
        for _ in range(rounds):
            for tag, cfg in cfgs.items():
                setup(cfg)
                run(dz, x, n, cin, kw, wss[tag], db)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(dz, x, n, cin, kw, wss[tag], db)
                e1.record()
                torch.cuda.synchronize()
                times[tag].append(e0.elapsed_time(e1) * 100.0)
        fl = 2.0 * B * L * n * kw * cin
        med = {t: sorted(v)[len(v) // 2] for t, v in times.items()}
        print(f"{name:16s} {B * L:6d} | {plans['map0'][1]:4d} {med['map0']:7.1f} {fl / med['map0'] / 1e6:5.0f} | {med['map1']:7.1f} "
              f"{fl / med['map1'] / 1e6:5.0f} {med['map0'] / med['map1']:.3f} | {plans['tall'][1]:4d} {med['tall']:7.1f} "
              f"{fl / med['tall'] / 1e6:5.0f} {med['map0'] / med['tall']:.3f}", flush=True)
    lib.styler_wgrad_tune(0, 0); lib.styler_wgrad_tune(1, 1); lib.styler_wgrad_tune(2, 0)


if __name__ == "__main__":
    main()
