#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "attention or attn" 2>&1 | tail -3
timeout 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_bf16_parity.py tests/test_hip_backward.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['repeat']['ms_per_step_median'])"; done
