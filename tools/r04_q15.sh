#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04q15}; mkdir -p $O
timeout 900 python -m pytest tests/test_11_oracle_c2c3.py tests/test_92_model_equivalences.py tests/test_20_hip_backward.py -q -m gpu -k "x3 or switches or reproducible or lstm or mlp or classifier" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 --prec bf16x3 --steps 10 --warmup 3"
for c in 0 1 0 1; do echo "== STYLER_X3_GROUPED=$c" >> $O/ab.txt; STYLER_X3_GROUPED=$c timeout 300 python bench.py $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt; done
cat $O/ab.txt; tail -3 $O/ab.err
