#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 96 80; do echo "== NMIN=$v"; STYLER_GEMM_BIG_NMIN=$v python tools/gemm_bench.py bf16 postnet_out mel_linear 2>&1 | grep -v amdgpu; done
bash tools/ab.sh STYLER_GEMM_BIG_NMIN 96 80 96 80 -- --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 0
