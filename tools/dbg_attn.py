import sys, torch
sys.path.insert(0, "/root/repo")
from styler_amd import ops
dev = torch.device("cuda")
for B, L, lens in [(2, 150, [150, 77]), (1, 200, [131]), (1, 200, [200])]:
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, L, 768, generator=g, dtype=torch.float64, requires_grad=True)
    ln = torch.tensor(lens)
    q, k, v = [t.view(B, L, 4, 64).permute(0, 2, 1, 3) for t in qkv.split(256, dim=-1)]
    s = (q @ k.transpose(-1, -2)) / 8.0
    s = s.masked_fill((torch.arange(L)[None, :] >= ln[:, None])[:, None, None, :], float("-inf"))
    out = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, L, 256)
    gy = torch.randn(B, L, 256, generator=g, dtype=torch.float64) * (torch.arange(L)[None, :, None] < ln[:, None, None])
    out.backward(gy)
    qd = qkv.detach().float().to(dev)
    lse = torch.empty(B, 4, L, device=dev)
    od16 = ops.attention_fwd(qd, ln.to(dev), lse=lse, prec=ops.PREC_BF16)
    dq16 = ops.attention_bwd(qd, od16, gy.float().to(dev), lse, ln.to(dev), prec=ops.PREC_BF16).cpu().double()
    ref = qkv.grad
    for b in range(B):
        for name, sl in (("dq", slice(0, 256)), ("dk", slice(256, 512)), ("dv", slice(512, 768))):
            e = (dq16[b, :, sl] - ref[b, :, sl]).abs()
            rows = e.max(dim=1).values
            bad = (rows > 0.03 * ref[b, :, sl].abs().max()).nonzero().flatten().tolist()
            print(B, L, lens, "item", b, name, "maxerr", float(e.max()), "refmax", float(ref[b, :, sl].abs().max()), "bad rows", bad[:6], "...", bad[-3:], len(bad))
