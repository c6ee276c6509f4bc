#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcg
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmcg/avail_sq.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')$RANDOM
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/gpurun_out/pmcg/$tag -o p -- python $R/tools/gemm_bench.py bf16 p_ffn_w1_k9 p_qkv > $R/gpurun_out/pmcg/$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcg/*/*/*counter_collection.csv") + glob.glob("gpurun_out/pmcg/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60] + " g=" + r["Grid_Size"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "conv_gemm" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
