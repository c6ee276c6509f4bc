#!/bin/bash
# round 5, lease 2: weight-gradient block map / k5 tall tile -- micro A/B, parity tests of the weight-gradient paths, step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q1; mkdir -p $O
timeout 600 python tools/wgrad_map_bench.py 5 > $O/wgrad_map_bench.txt 2>&1; cat $O/wgrad_map_bench.txt
timeout 900 python -m pytest tests/test_20_hip_backward.py tests/test_91_bf16_acts.py tests/test_14_train_step.py tests/test_11_oracle_c2c3.py -x -q -m gpu -k "wgrad or oracle or golden or train_state or conv_gemm" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
STYLER_WGRAD_K5_TALL=1 timeout 900 python -m pytest tests/test_20_hip_backward.py tests/test_91_bf16_acts.py tests/test_11_oracle_c2c3.py tests/test_92_model_equivalences.py -x -q -m gpu -k "wgrad or c3_train or reproducible" > $O/tests_tall.txt 2>&1; tail -3 $O/tests_tall.txt
bash tools/ab_env.sh r05q1 "STYLER_WGRAD_XCDMAP=0" "STYLER_WGRAD_XCDMAP=1" "STYLER_WGRAD_K5_TALL=1" "STYLER_WGRAD_XCDMAP=0" "STYLER_WGRAD_XCDMAP=1" "STYLER_WGRAD_K5_TALL=1"
