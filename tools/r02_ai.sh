#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['repeat']['ms_per_step_median'])"; done
bash tools/quick_trace.sh r02ai
grep -E "gn_|bn_|layernorm" gpurun_out/r02ai_kernel_stats.txt
