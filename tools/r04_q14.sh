#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04q14}; mkdir -p $O
for c in 1 0; do
  echo "== STYLER_X3_COMPACT=$c" >> $O/tests.txt
  STYLER_X3_COMPACT=$c timeout 900 python -m pytest tests/test_20_hip_backward.py tests/test_11_oracle_c2c3.py tests/test_92_model_equivalences.py -q -m gpu -k "x3 or wgrad_bf16 or conv_gemm_backward or switches or reproducible" >> $O/tests.txt 2>&1
  tail -3 $O/tests.txt
done
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 --prec bf16x3 --steps 10 --warmup 3"
for c in 0 1 0 1; do echo "== STYLER_X3_COMPACT=$c" >> $O/ab.txt; STYLER_X3_COMPACT=$c timeout 300 python bench.py $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt; done
cat $O/ab.txt
