#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04q12}; mkdir -p $O
timeout 900 python -m pytest tests/test_92_model_equivalences.py tests/test_20_hip_backward.py tests/test_90_equivalences.py tests/test_91_bf16_acts.py tests/test_11_oracle_c2c3.py -q -m gpu > $O/tests.txt 2>&1
tail -8 $O/tests.txt
bash tools/ab_env.sh $1 STYLER_BIAS_SLOTS=0 STYLER_BIAS_SLOTS=1 STYLER_BIAS_SLOTS=0 STYLER_BIAS_SLOTS=1 > /dev/null
cat $O/ab.txt
