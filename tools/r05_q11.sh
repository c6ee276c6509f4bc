#!/bin/bash
# NOTE: STYLER_WGRAD_HOT_MB and STYLER_LSTM_PIPELINE were experiment switches of this lease; both lost (profiles/r05_rejected_ab.txt) and their code was removed.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q11; mkdir -p $O
STYLER_WGRAD_HOT_MB=96 STYLER_LSTM_PIPELINE=1 timeout 900 python -m pytest tests/test_11_oracle_c2c3.py tests/test_14_train_step.py tests/test_15_dist_gpu.py -x -q -m gpu > $O/t.txt 2>&1; tail -3 $O/t.txt
timeout 200 python tools/lstm_bench.py > $O/lstm.txt 2>&1; grep -v amdgpu $O/lstm.txt | tail -12
bash tools/ab_env.sh r05q11 STYLER_WGRAD_HOT_MB=0 STYLER_WGRAD_HOT_MB=64 STYLER_WGRAD_HOT_MB=128 STYLER_WGRAD_HOT_MB=200 STYLER_LSTM_PIPELINE=1 "STYLER_LSTM_PIPELINE=1 STYLER_WGRAD_HOT_MB=128" STYLER_WGRAD_HOT_MB=0
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
for kv in STYLER_GEMM256_MIN_TILES=384 STYLER_GEMM256_MIN_TILES=190 STYLER_GEMM256_MIN_TILES=120 STYLER_GEMM256_MIN_TILES=60 STYLER_WGRAD_HOT_MB=128; do
  echo "== bf16x3 $kv" >> $O/abx3.txt
  env $kv timeout 300 python bench.py --prec bf16x3 $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/abx3.txt
done
cat $O/abx3.txt
