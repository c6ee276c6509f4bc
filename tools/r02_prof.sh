#!/bin/bash
# round-2 profiles of the final launch mix: kernel traces (the default bench command, hipGraph replay) + separate PMC passes
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02prof
cd /tmp && export TMPDIR=/tmp
for mode in train fwd; do
  timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r02prof/trace_$mode -o t -- python $R/bench.py --mode $mode --no-cpu --no-aux --steps 20 --warmup 5 --prof-steps 0 --repeat 0 > $R/gpurun_out/r02prof/trace_$mode.log 2>&1
  db=$(find $R/gpurun_out/r02prof/trace_$mode -name "*.db" | head -1)
  python $R/tools/prof_summary.py $db $R/gpurun_out/r02prof/r02_${mode}_bf16_graph_kernel_stats.txt > /dev/null
  rm -rf $R/gpurun_out/r02prof/trace_$mode
done
cd $R
bash tools/pmc_run.sh r02prof/pmc_train -- python $R/bench.py --mode train --no-cpu --no-aux --steps 4 --warmup 2 --prof-steps 0 --repeat 0 > /dev/null 2>&1
bash tools/pmc_run.sh r02prof/pmc_fwd -- python $R/bench.py --mode fwd --no-cpu --no-aux --steps 4 --warmup 2 --prof-steps 0 --repeat 0 > /dev/null 2>&1
{
python tools/pmc_traffic.py gpurun_out/r02prof/pmc_train "conv_gemm_kernel<2, 2, true" train_conv_gemm_2x2_bf16
python tools/pmc_traffic.py gpurun_out/r02prof/pmc_train "wgrad_tr_kernel" train_wgrad_bf16
python tools/pmc_traffic.py gpurun_out/r02prof/pmc_train "wgrad_reduce_multi" train_wgrad_reduce_multi
python tools/pmc_traffic.py gpurun_out/r02prof/pmc_train "strided_copy_multi" train_strided_copy_multi
python tools/pmc_traffic.py gpurun_out/r02prof/pmc_fwd "conv_gemm_kernel<2, 2, true" fwd_conv_gemm_2x2_bf16
} > gpurun_out/r02prof/r02_pmc_traffic.jsonl 2> gpurun_out/r02prof/pmc_traffic.err
python tools/pmc_summary.py $(find gpurun_out/r02prof/pmc_train -name "*counter_collection.csv") > gpurun_out/r02prof/r02_pmc_train_counters.txt 2>&1
python tools/pmc_summary.py $(find gpurun_out/r02prof/pmc_fwd -name "*counter_collection.csv") > gpurun_out/r02prof/r02_pmc_fwd_counters.txt 2>&1
rm -rf gpurun_out/r02prof/pmc_train gpurun_out/r02prof/pmc_fwd
cat gpurun_out/r02prof/r02_pmc_traffic.jsonl; head -12 gpurun_out/r02prof/r02_train_bf16_graph_kernel_stats.txt
