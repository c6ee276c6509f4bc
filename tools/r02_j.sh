#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "torch_library or postnet or batchnorm" 2>&1 | tail -5 > gpurun_out/r02j/t.txt
run() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat']['ms_per_step_median'])"; }
{
run STYLER_BN_RPB=32
run STYLER_BN_RPB=64
run STYLER_BN_RPB=128
run STYLER_LNBWD_ROWS=4
run STYLER_LNBWD_BLOCKS=1024
run STYLER_LNBWD_BLOCKS=1024 STYLER_LNBWD_ROWS=4
run STYLER_LNBWD_BLOCKS=2048
run STYLER_BN_RPB=32
} > gpurun_out/r02j/ab.txt 2>&1
cat gpurun_out/r02j/t.txt gpurun_out/r02j/ab.txt
