#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r02c/pytest.txt
cp gpurun_out/parity_report.json gpurun_out/r02c/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r02c/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-aux --steps 20 --warmup 5 --prof-steps 0 --repeat 0 > $GRAFT_REPO_ROOT/gpurun_out/r02c/trace.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/r02c/trace -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/r02c/train_graph_kernel_stats.txt > /dev/null
rm -rf gpurun_out/r02c/trace
cat gpurun_out/r02c/pytest.txt; tail -2 gpurun_out/r02c/trace.log | cut -c1-300; head -30 gpurun_out/r02c/train_graph_kernel_stats.txt
