#!/usr/bin/env python
"""Recurrent kernels of the AudioEncoder's BiLSTMs at the step's shape (B = 96 stacked items, S = 60): all four LSTMs in one
launch (lstm_bidir_multi, the step's form) vs one launch per LSTM -- is the batched launch bound by latency or by VALU issue?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda")
    B, S = 96, 60
    Hs = [80, 64, 64, 64]          # necks of the d / f0 / e / r streams (hparams.py:30)
    g = torch.Generator().manual_seed(0)
    gxs = [torch.randn(B, S, 8 * H, generator=g).to(dev) for H in Hs]
    whs = [(torch.randn(2, 4 * H, H, generator=g) / H ** 0.5).to(dev) for H in Hs]
    t_multi = timeit(lambda: ops.lstm_bidir_multi(gxs, whs, Hs, save=True))
    print(f"forward, four LSTMs in one launch: {t_multi:.1f} us")
    for parts in (1, 3):                     # round 6: the matrix-core recurrence (csrc/lstm_mfma.hip)
        t = timeit(lambda: ops.lstm_bidir_multi(gxs, whs, Hs, save=True, parts=parts))
        print(f"forward, four in one launch, MFMA parts = {parts}: {t:.1f} us")
    tot = 0.0
    for i, H in enumerate(Hs):
        gates = torch.empty(B, S, 8 * H, device=dev); cell = torch.empty(B, S, 2 * H, device=dev)
        t = timeit(lambda: ops.lstm_bidir(gxs[i], whs[i], H, cell_out=cell, gates_out=gates))
        tot += t
        print(f"  forward, LSTM {i} alone (H = {H}, {2 * B} blocks): {t:.1f} us")
    print(f"  sum of the four: {tot:.1f} us")
    outs, cells, gates = ops.lstm_bidir_multi(gxs, whs, Hs, save=True)
    douts = [torch.randn_like(o) for o in outs]
    t_bm = timeit(lambda: ops.lstm_bidir_bwd_multi(douts, gates, cells, whs, Hs))
    print(f"backward, four in one launch: {t_bm:.1f} us")
    for parts in (1, 3):
        t = timeit(lambda: ops.lstm_bidir_bwd_multi(douts, gates, cells, whs, Hs, parts=parts))
        print(f"backward, four in one launch, MFMA parts = {parts}: {t:.1f} us")
    tot = 0.0
    for i, H in enumerate(Hs):
        t = timeit(lambda: ops.lstm_bidir_bwd(douts[i], gates[i], cells[i], whs[i], H))
        tot += t
        print(f"  backward, LSTM {i} alone: {t:.1f} us")
    print(f"  sum of the four: {tot:.1f} us")


if __name__ == "__main__":
    main()
