#!/bin/bash
# round-4 check of the bf16x3 changes: kernel tests, oracle parity (all modes), A/B timings of the x3 switches
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04q8}; mkdir -p $O
timeout 900 python -m pytest tests/test_10_hip_parity.py tests/test_20_hip_backward.py tests/test_11_oracle_c2c3.py tests/test_91_bf16_acts.py -x -q -m gpu -k "attention or small_split or wgrad or conv_gemm or oracle or c2 or c3" > $O/tests.txt 2>&1
tail -5 $O/tests.txt
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
run() { echo "== $*" >> $O/ab.txt; env "$@" timeout 300 python bench.py $Q $EXTRA 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt; }
EXTRA="--prec bf16x3 --steps 10 --warmup 3"
run STYLER_X3_CACHE=0 STYLER_WGRAD_X3CAT=0
run STYLER_X3_CACHE=1 STYLER_WGRAD_X3CAT=0
run STYLER_X3_CACHE=1 STYLER_WGRAD_X3CAT=1
EXTRA=""
run X=1
cat $O/ab.txt
