#!/bin/bash
# generic multi-value env sweep of the default train bench: usage r06_q10.sh <outdir> <ENVVAR> <reps> v1 v2 ...
O=gpurun_out/$1; V=$2; R=$3; shift 3; mkdir -p $O
ab() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat'])"; }
for r in $(seq 1 $R); do for v in "$@"; do ab $V=$v; done; done | tee $O/ab.txt
