#!/bin/bash
cd $GRAFT_REPO_ROOT
STYLER_GEMM_PF2=1 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bf16_parity.py -m gpu -q -x 2>&1 | tail -2
for v in 0 1; do echo "== STYLER_GEMM_PF2=$v"; STYLER_GEMM_PF2=$v timeout 200 python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu.ids; done
bash tools/ab.sh STYLER_GEMM_PF2 0 -1 1 0 -1 1 -- --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 0
