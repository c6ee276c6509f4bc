#!/bin/bash
# round 5: PMC passes (separate runs, --kernel-trace only) on the config-4 attention forward launch: the round-4 kernel
# (STYLER_ATTN_FWD_V=0) vs the round-5 default (variant 7), same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 7; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    STYLER_ATTN_FWD_V=$v timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/v${v}_$i -o p -- python $R/tools/attn_c4_once.py > $O/v${v}_$i.log 2>&1
  done
  echo "## STYLER_ATTN_FWD_V=$v" >> $O/pmc.txt
  python $R/tools/pmc_summary.py $(find $O/v${v}_* -name "*counter_collection.csv") 2>&1 | grep -A 40 "attention_fwd" >> $O/pmc.txt
  rm -rf $O/v${v}_[123]
done
cat $O/pmc.txt
