#!/bin/bash
# round-4 check of: small-M split-K, x3 attention, low-part flags -> kernel tests, oracle parity, A/B timings
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04q7}; mkdir -p $O
timeout 900 python -m pytest tests/test_10_hip_parity.py tests/test_20_hip_backward.py tests/test_11_oracle_c2c3.py -x -q -m gpu -k "attention or small_split or wgrad or conv_gemm or oracle or c2 or c3" > $O/tests.txt 2>&1
tail -5 $O/tests.txt
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
run() { echo "== $*" >> $O/ab.txt; env "$@" timeout 300 python bench.py $Q $EXTRA 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt; }
EXTRA=""
run STYLER_GEMM_SMALL_SPLITK=0
run STYLER_GEMM_SMALL_SPLITK=1
run STYLER_GEMM_SMALL_SPLITK=0
run STYLER_GEMM_SMALL_SPLITK=1
EXTRA="--prec bf16x3 --steps 10 --warmup 3"
run STYLER_ATTN_X3=0 STYLER_GEMM_SMALL_SPLITK=0
run STYLER_ATTN_X3=1 STYLER_GEMM_SMALL_SPLITK=0
run STYLER_ATTN_X3=1 STYLER_GEMM_SMALL_SPLITK=1
cat $O/ab.txt
