#!/bin/bash
# round 5: the loss-only side stream (STYLER_PRED_STREAM=1) through the parity / equivalence / reproducibility / two-rank tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q8; mkdir -p $O
STYLER_PRED_STREAM=1 timeout 1200 python -m pytest tests/test_11_oracle_c2c3.py tests/test_14_train_step.py tests/test_15_dist_gpu.py tests/test_90_equivalences.py tests/test_92_model_equivalences.py tests/test_93_x3_producers.py -x -q -m gpu > $O/t.txt 2>&1; tail -6 $O/t.txt
