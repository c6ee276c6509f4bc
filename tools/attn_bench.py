#!/usr/bin/env python
"""Micro-benchmark of the bf16 attention kernels: the C2 decoder shape (B=48, L=441, VCTK-like lengths) and the config-4
decoder shape (both decode branches: B=256, T=2000), qkv / output stored as bf16 as in the throughput mode.  Also prints the
forward's error against fp64 math on a small case (the kernel's numerics depend on a switch: STYLER_ATTN_LAZY)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from styler_amd import ops
dev = torch.device("cuda")


def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def run(name, B, L, lo, n=20):
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(lo, L + 1, (B,), generator=g).to(dev)
    qkv = torch.randn(B, L, 768, device=dev)
    q16 = qkv.to(torch.bfloat16)
    lse = torch.empty(B, 4, L, device=dev)
    dout = torch.randn(B, L, 256, device=dev)
    fl = 4.0 * 64 * 4 * float((lens.double() ** 2).sum())          # 2 products x 2 flop x d_k x heads x sum(len^2)
    f32 = t(lambda: ops.attention_fwd(qkv, lens, lse=lse, prec=ops.PREC_BF16), n)
    f16 = t(lambda: ops.attention_fwd(q16, lens, lse=lse, prec=ops.PREC_BF16, out_bf16=True), n)
    out = ops.attention_fwd(q16, lens, lse=lse, prec=ops.PREC_BF16, out_bf16=True)
    b16 = t(lambda: ops.attention_bwd(q16, out, dout.to(torch.bfloat16), lse, lens, prec=ops.PREC_BF16, out_bf16=True), n)
    print(f"{name:12s} B={B} L={L} mean len {float(lens.float().mean()):.0f}: fwd fp32-in {f32:8.1f} us  fwd bf16-in/out {f16:8.1f} us "
          f"= {fl / f16 / 1e6:6.0f} TFLOP/s ({fl / f16 / 1e6 / 2500:.3f} of peak)   bwd (dq + dkv) {b16:8.1f} us = "
          f"{2.5 * fl / b16 / 1e6:6.0f} TFLOP/s", flush=True)


def accuracy():
    B, L = 3, 333
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(B, L, 768, generator=g) * 2).to(torch.bfloat16)
    ln = torch.tensor([333, 200, 77])
    q, k, v = [x.view(B, L, 4, 64).permute(0, 2, 1, 3).double() for x in qkv.double().split(256, dim=-1)]
    s = (q @ k.transpose(-1, -2)) / 8.0
    s = s.masked_fill((torch.arange(L)[None, :] >= ln[:, None])[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, L, 256)
    lse = torch.empty(B, 4, L, device=dev)
    out = ops.attention_fwd(qkv.to(dev), ln.to(dev), lse=lse, prec=ops.PREC_BF16)
    vq = (torch.arange(L)[None, :] < ln[:, None])
    e = float(((out.double().cpu() - ref) * vq[..., None]).abs().max())
    el = float(((lse.double().cpu() - torch.logsumexp(s, -1)) * vq[:, None, :]).abs().max())
    print(f"accuracy (bf16 qkv, scores x2): out max abs err {e:.3e}, lse max abs err {el:.3e}")


if __name__ == "__main__":
    print("STYLER_ATTN_LAZY =", os.environ.get("STYLER_ATTN_LAZY", "(default)"))
    accuracy()
    run("c2_decoder", 48, 441, 150, 30)
    run("c4_decoder", 256, 2000, 1000, 5)
