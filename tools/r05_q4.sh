#!/bin/bash
# round 5, lease 5: BatchNorm column statistics as per-block slots (no fp64 atomics) -- tests, default bench line (hbm rooflines)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q4; mkdir -p $O
timeout 900 python -m pytest tests/test_10_hip_parity.py tests/test_20_hip_backward.py tests/test_91_bf16_acts.py tests/test_92_model_equivalences.py tests/test_90_equivalences.py tests/test_14_train_step.py tests/test_11_oracle_c2c3.py -x -q -m gpu -k "batchnorm or postnet or conv_norm or reproducible or golden or oracle or segments or switches" > $O/t.txt 2>&1; tail -5 $O/t.txt
timeout 600 python bench.py --no-cpu --no-aux > $O/bench.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r05q4/bench.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d.get('repeat'))
        for h in d['roofline']['hbm']: print(h['kernel'], h.get('shape'), h.get('avg_us'), h.get('frac'))
PY
