#!/usr/bin/env python
"""Phase timeline of styler_linear_ln blocks (styler_linear_ln_set_trace): where a block's life goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from styler_amd import ops
dev = torch.device("cuda")
for name, B, L, K, valid in (("dec_fc", 1, 42336, 256, 27060), ("dec_w2", 1, 42336, 1024, 27060), ("c4_fc", 1, 256000, 256, 192000)):
    a = torch.randn(B, L, K, device=dev).to(torch.bfloat16)
    res = torch.randn(B, L, 256, device=dev).to(torch.bfloat16)
    s, y = torch.empty_like(res), torch.empty_like(res)
    w = (torch.randn(256, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias, ga, be = torch.randn(256, device=dev), torch.randn(256, device=dev), torch.randn(256, device=dev)
    lens = torch.tensor([valid], device=dev)
    nb = (L + 127) // 128
    buf = torch.zeros(nb, 8, device=dev, dtype=torch.int64)
    f = lambda: ops.linear_ln(a, w, bias, res, ga, be, lens=lens, drop_p=0.1, drop_seed=5, sum_out=s, out=y)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ops.lib.styler_linear_ln_set_trace(buf.data_ptr())
    f()
    torch.cuda.synchronize()
    ops.lib.styler_linear_ln_set_trace(None)
    t = buf.cpu().numpy().astype(np.int64)
    t = t[t[:, 1] != 0]
    st = t[:, 1:8].astype(np.float64) * 0.01
    t0 = st[:, 0].min()
    ph = np.diff(st, axis=1)
    q = lambda x: "%6.2f %6.2f %6.2f" % tuple(np.percentile(x, [10, 50, 90]))
    print(f"{name}: {len(t)} live blocks, launch span {st[:, 6].max() - t0:.1f} us; block life p10/p50/p90 {q(st[:, 6] - st[:, 0])}")
    for i, lab in enumerate(("entry -> masks known", "-> step 0 landed", "-> K loop done", "-> tile staged in LDS", "-> rows issued", "-> stores acknowledged")):
        print(f"    {lab:26s} {q(ph[:, i])}")
    starts = np.sort(st[:, 0] - t0)
    print("    block start offsets p10/p50/p90/max: %.2f %.2f %.2f %.2f" % (*np.percentile(starts, [10, 50, 90]), starts[-1]))
