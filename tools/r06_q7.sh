#!/bin/bash
# generic: a few test files + N repeats of the default train bench (no env switch); usage r06_q7.sh <outdir> <reps> <pytest args...>
O=gpurun_out/$1; R=$2; shift 2; mkdir -p $O
timeout 1200 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
for r in $(seq 1 $R); do
  timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['repeat'])"
done | tee $O/bench.txt
