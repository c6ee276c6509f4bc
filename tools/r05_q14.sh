#!/bin/bash
# NOTE: variant 23 (V tile at a row stride of 48 dwords) showed no gain and is not compiled in any more.
# round 5: embedding on the main stream (text-stream join), attention forward variant 23 (V tile stride 48) vs 7
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q14; mkdir -p $O
for v in 7 23 7 23; do
  echo "== STYLER_ATTN_FWD_V=$v" >> $O/attn.txt
  STYLER_ATTN_FWD_V=$v timeout 120 python tools/attn_bench.py 2>&1 | grep -E "accuracy|decoder" | cut -c1-150 >> $O/attn.txt
done
cat $O/attn.txt
STYLER_ATTN_FWD_V=23 timeout 300 python -m pytest tests -x -q -m gpu -k "attention or attn or c4" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_92_model_equivalences.py tests/test_11_oracle_c2c3.py tests/test_14_train_step.py tests/test_15_dist_gpu.py -x -q -m gpu > $O/t.txt 2>&1; tail -3 $O/t.txt
