#!/bin/bash
# re-check of older switches on the final round-6 build (the launch mix changed): one box, two rounds
O=gpurun_out/r06x; mkdir -p $O
ab() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat']['ms_per_step_median'])"; }
for r in 1 2; do
  ab A=default
  ab STYLER_TEXT_STREAM=0
  ab STYLER_PRED_STREAM=0
  ab STYLER_PRED_STREAM_CLS=0
  ab STYLER_GEMM256_MIN_TILES=300
  ab STYLER_GEMM256_MIN_TILES=450
  ab STYLER_GEMM256_MIN_TILES3=200
  ab STYLER_GEMM256_MIN_TILES3=0
  ab STYLER_WGRAD_K5_TALL=0
  ab STYLER_SKIP_DAT_NOISE=0
  ab STYLER_LINEAR_LN=0
done | tee $O/ab.txt
