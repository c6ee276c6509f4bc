"""Root cause of the round-4 driver failure `test_conv_gemm_backward[2-50-256-1024-9-1]` (fp32 mode, dX 3.098e-3 > 2e-5).

Re-runs the OLD form of that test (nn.Conv1d weights from torch's global RNG, fp64 reference with ITS OWN ReLU mask, no
kink guard) over N global seeds on the GPU and, per seed, reports
  * err_own   = max|dx_hip - dx_ref| / max|dx_ref|, reference masked by its fp64 pre-activation (what the old test asserted),
  * flips     = number of elements where the kernel's fp32 mask (y > 0 of its own forward) differs from the fp64 one,
  * zmax_flip = largest |z_fp64| among those elements (a legitimate flip has |z| at the fp32 rounding level of the sum),
  * err_kmask = the same error with the reference's ReLU derivative taken with the KERNEL's mask.
Verdict per seed: 'kink' if err_own > tol but err_kmask <= tol and zmax_flip < 1e-5; 'DEFECT' if err_kmask > tol.
A kernel defect in the long-K (K = 9216) 64 x 64 fp32 path would show up as err_kmask > tol.

usage: python tools/flake_hunt.py [--seeds 300] [--first 0]   (GPU box; prints one line per seed + a summary)"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(seed, dev, B=2, L=50, cin=256, n=1024, kw=9, tol=2e-5):
    from styler_amd import autograd as AG
    from styler_amd import ops
    from styler_amd.runtime import Derived
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(kw * 100 + n)
    conv = nn.Conv1d(cin, n, kw, padding=kw // 2).double()        # global RNG, as the old test
    x = torch.randn(B, L, cin, generator=g, dtype=torch.float64, requires_grad=True)
    res = torch.randn(B, L, n, generator=g, dtype=torch.float64)
    z = conv(x.transpose(1, 2)).transpose(1, 2)
    gy = torch.randn(B, L, n, generator=g, dtype=torch.float64)
    (torch.relu(z) + res).backward(gy)
    dx_own = x.grad.clone()

    holder = nn.Conv1d(cin, n, kw, padding=kw // 2).to(dev)
    with torch.no_grad():
        holder.weight.copy_(conv.weight.float()); holder.bias.copy_(conv.bias.float())
    xd = x.detach().float().to(dev).requires_grad_(True)
    rd = res.float().to(dev).requires_grad_(True)
    yd = AG.ConvGemmFn.apply(xd, rd, holder.weight, holder.bias, Derived(), "t", kw, 1, False)
    yd.backward(gy.float().to(dev))
    dx_hip = xd.grad.double().cpu()

    # the kernel's own mask: y > 0 of its forward without the residual (what ConvGemmFn saves)
    w, prec = AG.gemm_weight(Derived(), "t", holder.weight, cin)
    yk = ops.conv_gemm(xd.detach(), w, holder.bias, kw=kw, n=n, act=1, prec=prec)
    kmask = (yk > 0).cpu()
    rmask = z.detach() > 0
    flips = kmask != rmask
    nflip = int(flips.sum())
    zmax = float(z.detach().abs()[flips].max()) if nflip else 0.0

    x2 = x.detach().clone().requires_grad_(True)
    z2 = conv(x2.transpose(1, 2)).transpose(1, 2)
    z2.backward(gy * kmask.double())
    dx_k = x2.grad

    den = float(dx_own.abs().max())
    e_own = float((dx_hip - dx_own).abs().max()) / den
    e_k = float((dx_hip - dx_k).abs().max()) / float(dx_k.abs().max())
    verdict = "ok" if e_own <= tol else ("kink" if (e_k <= tol and zmax < 1e-5) else "DEFECT")
    if e_k > tol:
        verdict = "DEFECT"
    return dict(seed=seed, err_own=e_own, flips=nflip, zmax_flip=zmax, err_kmask=e_k, verdict=verdict)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=300)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from styler_amd import rt
    counts = {}
    worst_k = 0.0
    print(f"# flake hunt: conv_gemm backward [B=2 L=50 cin=256 n=1024 kw=9 ReLU], fp32 mode, tol 2e-5, seeds "
          f"{a.first}..{a.first + a.seeds - 1}")
    for s in range(a.first, a.first + a.seeds):
        r = one(s, dev)
        counts[r["verdict"]] = counts.get(r["verdict"], 0) + 1
        worst_k = max(worst_k, r["err_kmask"])
        if r["verdict"] != "ok" or s % 25 == 0:
            print("seed %4d  err_own %.3e  flips %d  zmax_flip %.3e  err_kmask %.3e  %s"
                  % (s, r["err_own"], r["flips"], r["zmax_flip"], r["err_kmask"], r["verdict"]))
    print(f"# summary: {counts}; worst err_kmask over all seeds {worst_k:.3e} (tol 2e-5)")
    sys.exit(1 if counts.get("DEFECT") else 0)


if __name__ == "__main__":
    main()
