#!/bin/bash
# kernel trace of the default training command -> per-kernel stats, timeline, ordered listing of one step
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python $R/bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 0 --mode train --steps 20 --warmup 5 > $O/tr.log 2>&1
db=$(find $O/tr -name "*.db" | head -1)
python $R/tools/prof_summary.py $db $O/kernel_stats.txt > /dev/null
python $R/tools/timeline.py $db > $O/timeline.txt 2>&1
python $R/tools/step_listing.py $db > $O/step_listing.txt 2>&1
rm -rf $O/tr
head -30 $O/timeline.txt; tail -2 $O/step_listing.txt
