#!/bin/bash
# round 5, lease 3: bf16x3 producer-side splits -- bit-equality tests, the mode's oracle parity tests, step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q2; mkdir -p $O
timeout 900 python -m pytest tests/test_93_x3_producers.py -x -q -m gpu > $O/t93.txt 2>&1; tail -15 $O/t93.txt
timeout 900 python -m pytest tests/test_11_oracle_c2c3.py tests/test_92_model_equivalences.py tests/test_20_hip_backward.py -x -q -m gpu -k "x3 or conv_gemm_backward or groupnorm or batchnorm or layernorm" > $O/tx3.txt 2>&1; tail -3 $O/tx3.txt
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
for kv in STYLER_X3_PRODUCERS=0 STYLER_X3_PRODUCERS=1 STYLER_X3_PRODUCERS=0 STYLER_X3_PRODUCERS=1; do
  echo "== bf16x3 $kv" >> $O/ab.txt
  env $kv timeout 300 python bench.py --prec bf16x3 $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt
done
for kv in STYLER_WGRAD_K5_TALL=0 STYLER_WGRAD_K5_TALL=1; do
  echo "== bf16 $kv" >> $O/ab.txt
  env $kv timeout 300 python bench.py $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt
done
cat $O/ab.txt; tail -5 $O/ab.err
