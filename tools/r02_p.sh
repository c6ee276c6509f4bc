#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/gemm_bench.py bf16 io qkv attn_fc ffn_w2_k1 p_qkv p_attn_fc p_ffn_w2_k1 p_dx_qkv p_ffn_w1_k9 pred_k3 2>&1 | grep -v amdgpu.ids
bash tools/quick_trace.sh r02p
head -60 gpurun_out/r02p_kernel_stats.txt
