#!/bin/bash
# NOTE: STYLER_TEXT_LATE was an experiment switch of this lease; it lost (profiles/r05_rejected_ab.txt) and its code was removed.
# round 5: rt.text_late (text encoder forked at the BiLSTM section, GateFn) -- tests, A/B, timeline; attention forward variant 7 as the default
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q13; mkdir -p $O
timeout 900 python -m pytest tests/test_92_model_equivalences.py tests/test_11_oracle_c2c3.py tests/test_14_train_step.py tests/test_15_dist_gpu.py -x -q -m gpu > $O/t.txt 2>&1; tail -3 $O/t.txt
bash tools/ab_env.sh r05q13 STYLER_TEXT_LATE=0 STYLER_TEXT_LATE=1 STYLER_TEXT_LATE=0 STYLER_TEXT_LATE=1
