#!/bin/bash
# split-K = 2 finished in the kernel: tests + same-box A/B of the switch
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python -m pytest tests/test_10_hip_parity.py tests/test_14_train_step.py tests/test_92_model_equivalences.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
ab() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat'])"; }
for r in 1 2 3; do
  ab STYLER_GEMM256_FIXUP=0
  ab STYLER_GEMM256_FIXUP=1
done | tee $O/ab.txt
