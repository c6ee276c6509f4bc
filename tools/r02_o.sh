#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r02o/t.txt
for v in 0 1; do
  STYLER_GEMM_ASTAT=$v timeout 300 python tools/gemm_bench.py bf16 io qkv attn_fc mel_linear p_qkv p_attn_fc p_dx_qkv ffn_w2_k1 > gpurun_out/r02o/gemm_astat$v.txt 2>&1
done
run() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat']['ms_per_step_median'])"; }
{
run STYLER_GEMM_ASTAT=0
run STYLER_GEMM_ASTAT=1
run STYLER_GEMM_ASTAT=0
run STYLER_GEMM_ASTAT=1
} > gpurun_out/r02o/ab.txt 2>&1
cat gpurun_out/r02o/t.txt gpurun_out/r02o/ab.txt; paste <(grep -v amdgpu gpurun_out/r02o/gemm_astat0.txt | awk '{print $1,$3,$7,$8}') <(grep -v amdgpu gpurun_out/r02o/gemm_astat1.txt | awk '{print $7,$8,$10,$11}')
