#!/bin/bash
# round 5, lease: after making the split outputs template arguments -- tests, default bench (hbm rooflines), bf16x3 A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q5; mkdir -p $O
timeout 900 python -m pytest tests/test_93_x3_producers.py tests/test_10_hip_parity.py tests/test_20_hip_backward.py tests/test_91_bf16_acts.py tests/test_11_oracle_c2c3.py -x -q -m gpu -k "producers or split or x3 or batchnorm or groupnorm or layernorm or attention or oracle" > $O/t.txt 2>&1; tail -4 $O/t.txt
timeout 600 python bench.py --no-cpu --no-aux > $O/bench.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r05q5/bench.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d.get('repeat'))
        for h in d['roofline']['hbm']: print(h['kernel'][:50], h.get('avg_us'), h.get('frac'))
PY
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
for kv in STYLER_X3_PRODUCERS=0 STYLER_X3_PRODUCERS=1 STYLER_X3_PRODUCERS=0 STYLER_X3_PRODUCERS=1; do
  echo "== bf16x3 $kv" >> $O/ab.txt
  env $kv timeout 300 python bench.py --prec bf16x3 $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt
done
cat $O/ab.txt
