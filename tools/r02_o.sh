#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "gemm or conv or linear or ffn or postnet" 2>&1 | tail -3
timeout 200 python tools/gemm_trace.py 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import torch
dev = torch.device("cuda")
for mb in (28, 83, 256):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    for name, fn in (("fill", lambda: y.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("read(sum)", lambda: x.sum())):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        by = n * 4 * (2 if name == "copy" else 1)
        print(f"{name:10s} {mb:4d} MB  {us:8.1f} us  {by / us / 1e6:6.2f} TB/s")
PY
for i in 1 2; do python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['repeat']['ms_per_step_median'])"; done
