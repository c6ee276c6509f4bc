#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 900 python -m pytest tests/test_hip_backward.py tests/test_bf16_parity.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r02g/pytest.txt
run() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat']['ms_per_step_median'])"; }
{
run STYLER_WGRAD_GROUP_TOTAL=768
run STYLER_WGRAD_GROUP_TOTAL=512
run STYLER_WGRAD_GROUP_TOTAL=1024
run STYLER_WGRAD_GROUP_TOTAL=1536
run STYLER_WGRAD_GROUP_TOTAL=100000
} > gpurun_out/r02g/ab.txt 2>&1
cat gpurun_out/r02g/pytest.txt gpurun_out/r02g/ab.txt
