#!/bin/bash
# NOTE: the per-tap-read build lost (profiles/r05_rejected_ab.txt); the code was reverted, libstyler_hip_alt.so is not built any more.
# round 5, last lease: the ring kernels' x operand per tap as two transposed LDS reads (default build) vs the sliding register
# window (libstyler_hip_alt.so = -DSTYLER_WGRAD_TAPREAD=0), same box: stand-alone kernels, tests, step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q16; mkdir -p $O
A=$GRAFT_REPO_ROOT/styler_amd/libstyler_hip_alt.so
{ echo "## sliding window (alt build)"; STYLER_LIB=$A timeout 200 python tools/wgrad_bench.py 3 2>&1 | grep -v amdgpu | cut -c1-150
  echo "## per-tap reads (default build)"; timeout 200 python tools/wgrad_bench.py 3 2>&1 | grep -v amdgpu | cut -c1-150; } > $O/wgrad.txt
cat $O/wgrad.txt
timeout 600 python -m pytest tests/test_20_hip_backward.py tests/test_11_oracle_c2c3.py tests/test_14_train_step.py tests/test_91_bf16_acts.py -x -q -m gpu -k "wgrad or c3_train or golden or oracle or reproducible or conv_gemm_backward or bf16" > $O/t.txt 2>&1; tail -3 $O/t.txt
bash tools/ab_env.sh r05q16 STYLER_LIB=$A STYLER_LIB= STYLER_LIB=$A STYLER_LIB=
