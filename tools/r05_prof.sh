#!/bin/bash
# round-5 profiles of the current launch mix: kernel traces (default train command = hipGraph replay -> stats, timeline, ordered
# listing; C2 forward; C4 forward; the bf16x3 train step), separate PMC passes, the wgrad A/B, the default bench line.
# Everything lands in gpurun_out/r05prof/ (copy what is to be judged into profiles/).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() {   # trace <name> <bench args...>
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_$name -o t -- python $R/bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 0 "$@" > $O/trace_$name.log 2>&1
  db=$(find $O/trace_$name -name "*.db" | head -1)
  python $R/tools/prof_summary.py $db $O/r05_${name}_kernel_stats.txt > /dev/null
  if [ "$name" = "train_bf16_graph" ]; then
    python $R/tools/timeline.py $db > $O/r05_timeline.txt 2>&1
    python $R/tools/step_listing.py $db > $O/r05_step_listing.txt 2>&1
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
bash tools/pmc_run.sh r05prof/pmc_train -- python $R/bench.py --mode train --no-cpu --no-aux --no-hbm --steps 4 --warmup 2 --prof-steps 0 --repeat 0 > /dev/null 2>&1
{
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "conv_gemm_kernel<2, 2, true|conv_gemm256_kernel" train_conv_gemm_2x2_bf16
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "conv_gemm256_kernel" train_conv_gemm256
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "wgrad_tr_kernel|wgrad_dma_kernel" train_wgrad_bf16
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "wgrad_dma_kernel<9" train_wgrad_dma_k9
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "wgrad_dma_kernel<5" train_wgrad_dma_k5
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "wgrad_reduce_multi" train_wgrad_reduce_multi
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "length_regulate_kernel" train_length_regulate
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "add_layernorm_kernel" train_add_layernorm
python tools/pmc_traffic.py gpurun_out/r05prof/pmc_train "layernorm_bwd_kernel" train_layernorm_bwd
} > $O/r05_pmc_traffic.jsonl 2> $O/pmc_traffic.err
python tools/pmc_summary.py $(find gpurun_out/r05prof/pmc_train -name "*counter_collection.csv") > $O/r05_pmc_train_counters.txt 2>&1
rm -rf gpurun_out/r05prof/pmc_train
timeout 300 python tools/wgrad_bench.py 5 > $O/r05_wgrad_bench.txt 2>&1
timeout 300 python tools/wgrad_map_bench.py 5 > $O/r05_wgrad_tune_bench.txt 2>&1
timeout 300 python tools/gemm256_bench.py 3 > $O/r05_gemm256_bench.txt 2>&1
timeout 300 python tools/gemm_bench.py bf16 > $O/r05_gemm_bench.txt 2>&1
timeout 300 python tools/find_torch_ops.py > $O/r05_torch_ops.txt 2>&1
timeout 300 python tools/attn_bench.py > $O/r05_attn_bench.txt 2>&1
cp $O/r05_pmc_traffic.jsonl $R/profiles/r05_pmc_traffic.jsonl    # the default line below embeds THIS pass as roofline.traffic
cp $O/r05_train_bf16_graph_kernel_stats.txt $R/profiles/r05_train_bf16_graph_kernel_stats.txt   # ... and frac_rocprof from THIS trace
timeout 900 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
cat $O/r05_pmc_traffic.jsonl; head -12 $O/r05_timeline.txt; tail -c 400 $O/r05_bench_default.json
# round 5 extras: the weight-gradient phase trace (trace build, built in the container by tools/wgrad_trace.sh) with the
# descriptor table off / on, and the stand-alone kernel A/B of the table
if [ -f $R/styler_amd/libstyler_hip_trace.so ]; then
  { echo "# tools/wgrad_trace.py: per-wave cycle sums of the ring loop's phases (trace build), block 0 and 9"; echo "## descriptor table OFF (STYLER_WGRAD_DESCTAB=0: the round-4 path)";
    STYLER_WGRAD_DESCTAB=0 STYLER_LIB=$R/styler_amd/libstyler_hip_trace.so timeout 300 python tools/wgrad_trace.py 2>&1 | grep -v amdgpu.ids
    echo "## descriptor table ON (default)";
    STYLER_LIB=$R/styler_amd/libstyler_hip_trace.so timeout 300 python tools/wgrad_trace.py 2>&1 | grep -v amdgpu.ids; } > $O/r05_wgrad_trace.txt
fi
{ echo "# tools/wgrad_bench.py 5, same box: STYLER_WGRAD_DESCTAB=0 (per-issue descriptor arithmetic, round 4) then =1 (per-block LDS table, default)"
  for v in 0 1; do echo "## STYLER_WGRAD_DESCTAB=$v"; STYLER_WGRAD_DESCTAB=$v timeout 300 python tools/wgrad_bench.py 5 2>&1 | grep -v amdgpu.ids; done; } > $O/r05_wgrad_bench.txt
