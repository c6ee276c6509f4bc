#!/usr/bin/env python
"""Phase timeline of styler_conv_gemm blocks (styler_gemm_set_trace): where a block's life goes, per shape.

For every block: entry -> first tile staged -> main loop done -> stores issued -> stores acknowledged (100 MHz counter).
Prints medians / deciles of each phase in microseconds, the spread of block start times and the launch's span."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from styler_amd import ops

SHAPES = [  # name, B, L, cin, n, kw
    ("p_qkv", 1, 27060, 256, 768, 1), ("p_attn_fc", 1, 27060, 256, 256, 1), ("p_ffn_w2_k1", 1, 27060, 1024, 256, 1),
    ("p_ffn_w1_k9", 1, 27060, 256, 1024, 9), ("postnet_512_k5", 96, 441, 512, 512, 5), ("p_dx_w1_k9", 1, 27060, 1024, 256, 9),
]


def main():
    dev = torch.device("cuda")
    only = sys.argv[1:]
    for name, B, L, cin, n, kw in SHAPES:
        if only and name not in only:
            continue
        x = torch.randn(B, L, cin, device=dev)
        w = ops.cast_bf16(torch.randn(n, kw * cin, device=dev) / (kw * cin) ** 0.5)
        b = torch.randn(n, device=dev)
        y = torch.empty(B, L, n, device=dev)
        big = ops.lib.styler_conv_gemm_variant(B, L, cin, n, kw, ops.PREC_BF16) & 1
        tile = 128 if big else 64
        mt, nt = (B * L + tile - 1) // tile, (n + tile - 1) // tile
        grid = ((mt + 7) // 8) * 8 * nt
        buf = torch.zeros(grid, 8, device=dev, dtype=torch.int64)
        for _ in range(3):
            ops.conv_gemm(x, w, b, kw=kw, prec=ops.PREC_BF16, out=y)
        torch.cuda.synchronize()
        ops.lib.styler_gemm_set_trace(buf.data_ptr())
        ops.conv_gemm(x, w, b, kw=kw, prec=ops.PREC_BF16, out=y)
        torch.cuda.synchronize()
        ops.lib.styler_gemm_set_trace(None)
        t = buf.cpu().numpy().astype(np.int64)
        t = t[t[:, 1] != 0]
        st = t[:, 1:6].astype(np.float64) * 0.01                     # microseconds
        t0 = st[:, 0].min()
        ph = np.diff(st, axis=1)
        q = lambda a: "%6.2f %6.2f %6.2f" % tuple(np.percentile(a, [10, 50, 90]))
        print(f"{name}: {len(t)} blocks of {tile}^2, launch span {st[:, 4].max() - t0:.1f} us; per-block life p10/p50/p90 {q(st[:, 4] - st[:, 0])}")
        for i, lab in enumerate(("prologue+first tile", "main loop", "epilogue to stores issued", "store acknowledge")):
            print(f"    {lab:28s} {q(ph[:, i])}")
        starts = np.sort(st[:, 0] - t0)
        print("    block start offsets (us) p10/p50/p90/max: %.2f %.2f %.2f %.2f" % (*np.percentile(starts, [10, 50, 90]), starts[-1]))
        if os.environ.get("STYLER_LIB", "").endswith("_steps.so"):      # diagnosis build: per-phase cycle totals of the main loop
            p0, p1, p2, p3 = t[:, 6] >> 32, t[:, 6] & 0xffffffff, t[:, 7] >> 32, t[:, 7] & 0xffffffff
            nst = ((cin + 63) // 64) * kw
            print("    main-loop cycles per step (median): issue loads %.0f, ds_read + MFMA %.0f, wait loads + ds_write %.0f, barrier %.0f"
                  % tuple(np.median(x) / nst for x in (p0, p1, p2, p3)))
            continue
        hw = t[:, 6]
        cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4)          # cu_id | se_id << 4 (per XCD)
        key = (t[:, 0] % 8) * 1000 + cu
        per = np.bincount(np.unique(key, return_inverse=True)[1])
        print(f"    distinct (xcd, se, cu): {len(per)}, blocks per CU min/median/max {per.min()} {int(np.median(per))} {per.max()}")


if __name__ == "__main__":
    main()
