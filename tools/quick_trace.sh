#!/bin/bash
# usage: tools/quick_trace.sh <tag> [bench args]: kernel-trace summary of a short bench run -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/qt_$tag -o t -- python $R/bench.py --no-cpu --no-aux --steps 12 --warmup 3 --prof-steps 0 --repeat 0 "$@" > $R/gpurun_out/qt_$tag.log 2>&1
db=$(find $R/gpurun_out/qt_$tag -name "*.db" | head -1)
python $R/tools/prof_summary.py $db $R/gpurun_out/${tag}_kernel_stats.txt > /dev/null
rm -rf $R/gpurun_out/qt_$tag
