#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q10; mkdir -p $O
timeout 900 python -m pytest tests/test_20_hip_backward.py tests/test_91_bf16_acts.py tests/test_11_oracle_c2c3.py tests/test_92_model_equivalences.py tests/test_14_train_step.py -x -q -m gpu -k "wgrad or c3_train or reproducible or golden or lstm or oracle" > $O/t.txt 2>&1; tail -3 $O/t.txt
for v in 0 1; do echo "== DESCTAB=$v"; STYLER_WGRAD_DESCTAB=$v python tools/wgrad_bench.py 3 2>&1 | grep -v amdgpu | cut -c1-70,110-140; done
bash tools/ab_env.sh r05q10 STYLER_WGRAD_DESCTAB=0 STYLER_WGRAD_DESCTAB=1 STYLER_WGRAD_DESCTAB=0 STYLER_WGRAD_DESCTAB=1
