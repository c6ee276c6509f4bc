#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
run() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat'])"; }
{
run STYLER_WGRAD_STREAM=0 STYLER_WGRAD_BLOCKS=512
run STYLER_WGRAD_STREAM=1 STYLER_WGRAD_BLOCKS=512
run STYLER_WGRAD_STREAM=0 STYLER_WGRAD_BLOCKS=256
run STYLER_WGRAD_STREAM=1 STYLER_WGRAD_BLOCKS=256
run STYLER_WGRAD_STREAM=1 STYLER_WGRAD_BLOCKS=128
run STYLER_WGRAD_STREAM=1 STYLER_WGRAD_BLOCKS=256 STYLER_WGRAD_GROUP_BLOCKS=64
run STYLER_WGRAD_STREAM=0 STYLER_WGRAD_BLOCKS=512
} > gpurun_out/r02d/ab.txt 2>&1
STYLER_WGRAD_STREAM=1 timeout 300 python -m pytest tests/test_hip_backward.py -q -x -k "graphed or split_graph or deferred or train_step" 2>&1 | tail -5 >> gpurun_out/r02d/ab.txt
cat gpurun_out/r02d/ab.txt
