#!/bin/bash
# round 5, lease 4: bf16x3 producers incl. LayerNorm backward -- tests, kernel trace of the bf16x3 step, A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q3; mkdir -p $O
timeout 900 python -m pytest tests/test_93_x3_producers.py tests/test_11_oracle_c2c3.py tests/test_20_hip_backward.py tests/test_10_hip_parity.py -x -q -m gpu -k "x3 or producers or split or layernorm or attention" > $O/t.txt 2>&1; tail -5 $O/t.txt
bash tools/quick_trace.sh r05x3 --prec bf16x3 --mode train
head -45 gpurun_out/r05x3_kernel_stats.txt
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
for kv in STYLER_X3_PRODUCERS=0 STYLER_X3_PRODUCERS=1; do
  echo "== bf16x3 $kv" >> $O/ab.txt
  env $kv timeout 300 python bench.py --prec bf16x3 $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt
done
cat $O/ab.txt
