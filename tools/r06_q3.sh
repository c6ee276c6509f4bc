#!/bin/bash
# full GPU suite + step A/B of one env switch: usage r06_q3.sh <outdir> <ENVVAR> [bench args]
O=gpurun_out/$1; V=$2; shift 2; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
for r in 1 2; do for v in 0 1; do
  env $V=$v timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$v', d['ms_per_step'], d['repeat'])"
done; done | tee $O/ab.txt
