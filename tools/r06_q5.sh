#!/bin/bash
# fused loss head: tests + same-box A/B of the switch and of the masked-error grid cap
O=gpurun_out/r06m; mkdir -p $O
python -m pytest tests/test_14_train_step.py tests/test_11_oracle_c2c3.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
ab() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat'])"; }
for r in 1 2; do
  ab STYLER_FUSED_LOSS=0
  ab STYLER_FUSED_LOSS=1
  ab STYLER_FUSED_LOSS=1 STYLER_LOSS_BLOCKS=64
  ab STYLER_FUSED_LOSS=1 STYLER_LOSS_BLOCKS=128
  ab STYLER_FUSED_LOSS=1 STYLER_LOSS_BLOCKS=32
done | tee $O/ab.txt
