#!/bin/bash
# same-box A/B of two library builds: usage ab_lib.sh <outdir> <other .so> [reps]
O=gpurun_out/$1; mkdir -p $O; R=${3:-3}
one() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat'])"; }
for r in $(seq 1 $R); do one STYLER_LIB=$PWD/$2; one A=default; done | tee $O/ab_lib.txt
