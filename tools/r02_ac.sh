#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in top mask fence0; do echo "== $v"; STYLER_GEMM_PF2=0 STYLER_LIB=$GRAFT_REPO_ROOT/styler_amd/libstyler_hip_$v.so timeout 200 python tools/gemm_bench.py bf16 p_ffn_w1_k9 p_dx_w1_k9 postnet_512_k5 p_qkv p_attn_fc p_ffn_w2_k1 aenc_320_k5 2>&1 | grep -v amdgpu.ids; done
echo "== fence0 pf2 auto"; STYLER_LIB=$GRAFT_REPO_ROOT/styler_amd/libstyler_hip_fence0.so timeout 200 python tools/gemm_bench.py bf16 p_dx_w1_k9 p_attn_fc p_ffn_w2_k1 2>&1 | grep -v amdgpu.ids
STYLER_LIB=$GRAFT_REPO_ROOT/styler_amd/libstyler_hip_fence0.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "gemm or conv" 2>&1 | tail -2
