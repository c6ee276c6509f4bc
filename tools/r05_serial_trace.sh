cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STYLER_PRED_STREAM=0 STYLER_TEXT_STREAM=0 timeout 600 rocprofv3 --kernel-trace -d $O/trace_serial -o t -- python $R/bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 0 --mode train --steps 20 --warmup 5 > $O/trace_serial.log 2>&1
db=$(find $O/trace_serial -name "*.db" | head -1)
python $R/tools/prof_summary.py $db $O/r05_train_bf16_graph_serial_kernel_stats.txt > /dev/null
python $R/tools/timeline.py $db | head -5
rm -rf $O/trace_serial
tail -1 $O/trace_serial.log | cut -c1-200
