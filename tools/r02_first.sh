#!/bin/bash
# round-2 first GPU session: gated tests, A/B of the experimental switches, kernel trace of the current launch mix
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
STYLER_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02a/pytest.txt
{
tools/ab.sh STYLER_PAIR_AUDIO 0 1 0 1 -- --steps 60 --warmup 10 --prof-steps 0
tools/ab.sh STYLER_FUSED_SPLIT 0 1 0 1 -- --steps 60 --warmup 10 --prof-steps 0
STYLER_PAIR_AUDIO=1 STYLER_FUSED_SPLIT=1 timeout 300 python bench.py --no-cpu --steps 60 --warmup 10 --prof-steps 0 2>&1 | tail -1
} > gpurun_out/r02a/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r02a/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-graph --no-cpu --steps 10 --warmup 3 --prof-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r02a/trace.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/r02a/trace -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/r02a/train_kernel_stats.txt > /dev/null
rm -rf gpurun_out/r02a/trace
cat gpurun_out/r02a/pytest.txt gpurun_out/r02a/ab.txt
