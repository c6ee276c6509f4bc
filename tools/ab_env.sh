#!/bin/bash
# usage: tools/ab_env.sh OUTDIR "ENV1=a ENV2=b" "ENV1=c" ... : one short default-bench run per environment on the SAME box
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; shift; mkdir -p $O
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
for kv in "$@"; do
  echo "== $kv" >> $O/ab.txt
  env $kv timeout 300 python bench.py $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt
done
cat $O/ab.txt
