#!/bin/bash
# round 6, lease 1: linear_ln parity + micro-benchmark + step A/B
O=gpurun_out/r06i; mkdir -p $O
python -m pytest tests/test_22_linear_ln.py -x -q -m gpu > $O/tests.txt 2>&1; tail -15 $O/tests.txt
python tools/linear_ln_bench.py > $O/linear_ln_bench.txt 2>&1; cat $O/linear_ln_bench.txt
for r in 1 2; do for v in 0 1; do
  STYLER_LINEAR_LN=$v timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LINEAR_LN=$v', d['ms_per_step'], d['repeat'])"
done; done | tee $O/ab.txt
