#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/quick_trace.sh r02y
head -40 gpurun_out/r02y_kernel_stats.txt
