#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_bf16_parity.py -m gpu -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
bash tools/ab.sh STYLER_BF16_ACTS 0 1 0 1 -- --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 0
bash tools/quick_trace.sh r02y
head -30 gpurun_out/r02y_kernel_stats.txt
