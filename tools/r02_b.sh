#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r02b/pytest.txt
cp gpurun_out/parity_report.json gpurun_out/r02b/ 2>/dev/null
timeout 300 python tools/gemm_bench.py bf16 io > gpurun_out/r02b/gemm_io.txt 2>&1
STYLER_GEMM_TILE=1 timeout 300 python tools/gemm_bench.py bf16 io p_qkv p_attn_fc p_ffn_w2_k1 p_dx_qkv mel_linear attn_fc qkv > gpurun_out/r02b/gemm_io_tile64.txt 2>&1
timeout 600 python bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
cat gpurun_out/r02b/pytest.txt; tail -3 gpurun_out/r02b/bench.err; cat gpurun_out/r02b/bench.json
