#!/bin/bash
# round 5: the lazy attention forward's variants (STYLER_ATTN_FWD_V bit set: 1 setprio, 2 permlane swap + deferred row sum, 4 64-key
# softmax step, 8 register prefetch) at the C2 / C4 decoder shapes, same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q12; mkdir -p $O
for v in 0 1 2 3 4 6 7 14 15 0; do
  echo "== STYLER_ATTN_FWD_V=$v" >> $O/attn.txt
  STYLER_ATTN_FWD_V=$v timeout 120 python tools/attn_bench.py 2>&1 | grep -E "accuracy|decoder" | cut -c1-150 >> $O/attn.txt
done
cat $O/attn.txt
for v in 6 7 14; do
  echo "== tests, STYLER_ATTN_FWD_V=$v"; STYLER_ATTN_FWD_V=$v timeout 300 python -m pytest tests -x -q -m gpu -k "attention or attn or c4" 2>&1 | tail -2
done
