#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r02_prof.sh > gpurun_out/r02prof_run.log 2>&1
python tools/membound_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02prof/r02_membound_bench.txt
python tools/norm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02prof/r02_norm_bench.txt
STYLER_GN_FUSED=0 python tools/norm_bench.py 2>&1 | grep "^GN" >> gpurun_out/r02prof/r02_norm_bench.txt
python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu.ids > gpurun_out/r02prof/r02_gemm_bench.txt
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02prof/r02_attn_bench.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02prof/bench_default.json 2> gpurun_out/r02prof/bench_default.err
tail -1 gpurun_out/r02prof/bench_default.json
