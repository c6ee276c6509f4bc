import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
if len(sys.argv) > 2 and sys.argv[1] == "dump":
    import conftest
    from styler_amd import STYLER, rt
    from styler_amd.training import train_losses
    g = np.load(os.path.join(root, "tests/golden/full_teacher.npz"))
    dev = torch.device("cuda")
    b = {k[3:]: torch.from_numpy(np.ascontiguousarray(g[k])).to(dev) for k in g.files if k.startswith("in_")}
    if os.environ.get("PERTURB"):                    # one-ulp perturbation of every float input
        b = {k: (v * (1.0 + 2.0 ** -23) if v.dtype == torch.float32 else v) for k, v in b.items()}
    sd = conftest._reference_state_dict()
    m = STYLER()
    m.load_state_dict(sd); m = m.to(dev).train(); rt.disable_dropout = True
    losses = train_losses(m, b); losses[0].backward()
    d = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}
    d["__losses__"] = torch.stack([l.detach().double().reshape(()) for l in losses]).cpu()
    torch.save(d, sys.argv[2])
else:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    rows = []
    for k in a:
        d = float((a[k] - b[k]).abs().max()); s = float(b[k].abs().max()) + 1e-30
        rows.append((d / s, k))
    rows.sort(reverse=True)
    print("losses a", a["__losses__"].tolist()); print("losses b", b["__losses__"].tolist())
    for r in rows[:60]:
        if "w_ks.bias" in r[1] or "conv.bias" in r[1]: continue
        print("%.3e %s" % r)
