import sys; sys.path.insert(0, "/root/repo")
import torch
from styler_amd import ops
dev = torch.device("cuda")
gs = [torch.randn(29482716, device=dev) for _ in range(6)]
out = torch.zeros(1, dtype=torch.float64, device=dev)
for i in range(3): ops.sumsq(gs[i], out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(30): ops.sumsq(gs[i % 6], out)
e1.record(); torch.cuda.synchronize()
print("sumsq", round(e0.elapsed_time(e1) * 1e3 / 30, 1), "us")
out.zero_(); ops.sumsq(gs[0], out); print(float(out), float((gs[0].double() ** 2).sum()))
