#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
run() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat']['ms_per_step_median'])"; }
{
run STYLER_TEXT_STREAM=0
run STYLER_TEXT_STREAM=1
run STYLER_WGRAD_BLOCKS=768
run STYLER_WGRAD_BLOCKS=1024
run STYLER_WGRAD_GROUP_BLOCKS=256
run STYLER_WGRAD_GROUP_BLOCKS=64
run STYLER_TEXT_STREAM=0
} > gpurun_out/r02e/ab.txt 2>&1
cat gpurun_out/r02e/ab.txt
