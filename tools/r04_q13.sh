#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04q13}; mkdir -p $O
timeout 600 python -m pytest tests/test_10_hip_parity.py -x -q -m gpu -k "gemm or conv" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
ALT=$PWD/styler_amd/alt/libstyler_nodma.so
STYLER_LIB=$ALT timeout 300 python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu > $O/gb_nodma.txt
timeout 300 python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu > $O/gb_dma.txt
paste <(cut -c1-75 $O/gb_nodma.txt) <(cut -c50-75 $O/gb_dma.txt)
bash tools/ab_env.sh $1 STYLER_LIB=$ALT X=1 STYLER_LIB=$ALT X=1 > /dev/null; cat $O/ab.txt
