#!/bin/bash
# round-6 profiles of the current launch mix: kernel traces (default train command = hipGraph replay -> stats, timeline, ordered
# listing; C2 forward; C4 forward; the bf16x3 train step), separate PMC passes, the wgrad A/B, the default bench line.
# Everything lands in gpurun_out/r06prof/ (copy what is to be judged into profiles/).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() {   # trace <name> <bench args...>
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_$name -o t -- python $R/bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 0 "$@" > $O/trace_$name.log 2>&1
  db=$(find $O/trace_$name -name "*.db" | head -1)
  python $R/tools/prof_summary.py $db $O/r06_${name}_kernel_stats.txt > /dev/null
  if [ "$name" = "train_bf16_graph" ]; then
    python $R/tools/timeline.py $db > $O/r06_timeline.txt 2>&1
    python $R/tools/step_listing.py $db > $O/r06_step_listing.txt 2>&1
  fi
  rm -rf $O/trace_$name
}
trace train_bf16_graph --mode train --steps 20 --warmup 5
# the same step with every side stream off (text encoder, loss-only branch): per-kernel durations WITHOUT co-runners -- in the default
# trace above a launch's duration includes the time it shared the chip with a side-stream kernel (15 % of the step)
STYLER_PRED_STREAM=0 STYLER_TEXT_STREAM=0 trace train_bf16_graph_serial --mode train --steps 20 --warmup 5
trace fwd_bf16_graph --mode fwd --steps 20 --warmup 5
trace c4_fwd_dual --mode fwd --shape c4 --batch 128 --dual --steps 5 --warmup 2
trace train_bf16x3_graph --mode train --prec bf16x3 --steps 10 --warmup 3
cd $R
bash tools/pmc_run.sh r06prof/pmc_train -- python $R/bench.py --mode train --no-cpu --no-aux --no-hbm --steps 4 --warmup 2 --prof-steps 0 --repeat 0 > /dev/null 2>&1
{
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "conv_gemm_kernel<2, 2, true|conv_gemm256_kernel" train_conv_gemm_2x2_bf16
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "conv_gemm256_kernel" train_conv_gemm256
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "wgrad_tr_kernel|wgrad_dma_kernel" train_wgrad_bf16
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "wgrad_dma_kernel<9" train_wgrad_dma_k9
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "wgrad_dma_kernel<5" train_wgrad_dma_k5
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "wgrad_reduce_multi" train_wgrad_reduce_multi
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "length_regulate_kernel" train_length_regulate
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "add_layernorm_kernel" train_add_layernorm
python tools/pmc_traffic.py gpurun_out/r06prof/pmc_train "layernorm_bwd_kernel" train_layernorm_bwd
} > $O/r06_pmc_traffic.jsonl 2> $O/pmc_traffic.err
python tools/pmc_summary.py $(find gpurun_out/r06prof/pmc_train -name "*counter_collection.csv") > $O/r06_pmc_train_counters.txt 2>&1
rm -rf gpurun_out/r06prof/pmc_train
timeout 300 python tools/wgrad_bench.py 5 > $O/r06_wgrad_bench.txt 2>&1
timeout 300 python tools/gemm256_bench.py 3 > $O/r06_gemm256_bench.txt 2>&1
timeout 300 python tools/gemm_bench.py bf16 > $O/r06_gemm_bench.txt 2>&1
timeout 300 python tools/find_torch_ops.py > $O/r06_torch_ops.txt 2>&1
timeout 300 python tools/attn_bench.py > $O/r06_attn_bench.txt 2>&1
timeout 300 python tools/lstm_bench.py > $O/r06_lstm_bench_prof.txt 2>&1
timeout 300 python tools/step_gemms.py 3 > $O/r06_step_gemms.txt 2>&1
cp $O/r06_pmc_traffic.jsonl $R/profiles/r06_pmc_traffic.jsonl    # the default line below embeds THIS pass as roofline.traffic
cp $O/r06_train_bf16_graph_kernel_stats.txt $R/profiles/r06_train_bf16_graph_kernel_stats.txt   # ... and frac_rocprof from THIS trace
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err
cat $O/r06_pmc_traffic.jsonl; head -12 $O/r06_timeline.txt; tail -c 400 $O/r06_bench_default.json
