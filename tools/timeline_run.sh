#!/bin/bash
# kernel trace of the default training command + tools/timeline.py -> gpurun_out/tl/timeline.txt (copy to profiles/rNN_timeline.txt)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python $R/bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 0 --mode train --steps 20 --warmup 5 > $O/tr.log 2>&1
db=$(find $O/tr -name "*.db" | head -1)
python $R/tools/timeline.py $db > $O/timeline.txt 2>&1
rm -rf $O/tr
cat $O/timeline.txt
