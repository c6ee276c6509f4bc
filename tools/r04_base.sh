#!/bin/bash
# round-4 baseline on today's box: the default line + the existing concurrency switches (same-box A/B)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04base; mkdir -p $O
timeout 400 python bench.py > $O/default.json 2> $O/default.err
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
for kv in STYLER_WGRAD_STREAM=0 STYLER_WGRAD_STREAM=1 STYLER_TEXT_STREAM=0 STYLER_WGRAD_GROUP_ALL=1; do
  echo "== $kv" >> $O/ab.txt
  env $kv timeout 300 python bench.py $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt
done
cat $O/ab.txt; tail -c 1500 $O/default.json
