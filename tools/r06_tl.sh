#!/bin/bash
# quick: kernel trace of the default train command -> stats, timeline, ordered listing in gpurun_out/<tag>/
tag=${1:-r06tl}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 0 --mode train --steps 20 --warmup 5 "$@" > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1)
python $R/tools/prof_summary.py $db $O/kernel_stats.txt > /dev/null
python $R/tools/timeline.py $db > $O/timeline.txt 2>&1
python $R/tools/step_listing.py $db > $O/step_listing.txt 2>&1
rm -rf $O/trace
head -8 $O/timeline.txt; tail -3 $O/trace.log
