#!/usr/bin/env python
"""Where a weight-gradient ring kernel spends its cycles (trace build: tools/wgrad_trace.sh; run with
STYLER_LIB=styler_amd/libstyler_hip_trace.so).  Per wave of the first blocks: cycle sums of the loop's phases
[loop top | DMA wait (s_waitcnt vmcnt) | barrier 1 | DMA issue | MFMA half 1 | barrier 2 | MFMA half 2]."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
from styler_amd._lib import lib

SHAPES = [("dec_ffn_w1_k9", 61, 441, 256, 1024, 9), ("postnet_512_k5", 96, 441, 512, 512, 5), ("dec_ffn_w2_k1", 1, 27060, 1024, 256, 1)]
NAMES = ["dma wait", "barrier 1", "dma instrs", "mfma half 1", "barrier 2", "mfma half 2", "dma descr", "loop top"]


def main():
    dev = torch.device("cuda")
    cl = ctypes.CDLL(os.environ["STYLER_LIB"])
    cl.styler_wgrad_trace_ptr.argtypes = [ctypes.c_void_p]
    g = torch.Generator().manual_seed(4)
    for name, B, L, cin, n, kw in SHAPES:
        dz = torch.randn(B, L, n, generator=g).to(dev).to(torch.bfloat16)
        x = torch.randn(B, L, cin, generator=g).to(dev).to(torch.bfloat16)
        nb = int(lib.styler_wgrad_workspace_bytes_io(B, L, n, cin, kw, kw // 2, ops.PREC_BF16, 3))
        ws = torch.empty(nb // 4, device=dev)
        db = torch.zeros(n, device=dev)
        strides = (cin * kw, kw, 1) if kw > 1 else (cin, 1, 0)
        tr = torch.zeros(16 * 8 * 10, dtype=torch.int64, device=dev)

        def run():
            ops._chk(lib.styler_wgrad(dz.data_ptr(), dz.stride(1), x.data_ptr(), x.stride(1), ws.data_ptr(), db.data_ptr(), None,
                                      *strides, B, L, n, cin, kw, kw // 2, ops.PREC_BF16, ws.data_ptr(), 1, 3,
                                      torch.cuda.current_stream().cuda_stream), "styler_wgrad")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        assert cl.styler_wgrad_trace_ptr(ctypes.c_void_p(tr.data_ptr())) == 0
        run()
        torch.cuda.synchronize()
        cl.styler_wgrad_trace_ptr(None)
        t = tr.view(16, 8, 10).cpu()
        print(f"== {name}: rows {B * L}, n {n}, cin {cin}, kw {kw}")
        for b in (0, 9):
            for w in range(8):
                r = t[b, w]
                tot = int(r[9])
                if tot == 0:
                    continue
                parts = "  ".join(f"{NAMES[k]} {100.0 * int(r[k]) / tot:5.1f}%" for k in (7, 0, 1, 6, 2, 3, 4, 5))
                print(f"block {b:2d} wave {w}: trips {int(r[8]):4d}  total {tot:8d} cyc ({tot / max(1, int(r[8])):7.0f} / trip)  {parts}")


if __name__ == "__main__":
    main()
