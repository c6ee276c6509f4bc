#!/bin/bash
# generic env-switch A/B: usage r06_q8.sh <outdir> <ENVVAR> <reps> <pytest args...>
O=gpurun_out/$1; V=$2; R=$3; shift 3; mkdir -p $O
timeout 1200 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
ab() { env "$@" timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['repeat'])"; }
for r in $(seq 1 $R); do ab $V=0; ab $V=1; done | tee $O/ab.txt
