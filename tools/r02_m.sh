#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r02m/t.txt
run() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat']['ms_per_step_median'])"; }
{
run STYLER_TILED_COPY=0
run STYLER_TILED_COPY=1
run STYLER_TILED_COPY=0
run STYLER_TILED_COPY=1
} > gpurun_out/r02m/ab.txt 2>&1
bash tools/quick_trace.sh r02m
cat gpurun_out/r02m/t.txt gpurun_out/r02m/ab.txt; grep -E "strided_copy" gpurun_out/r02m_kernel_stats.txt
