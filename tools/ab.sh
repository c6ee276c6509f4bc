#!/bin/bash
# usage: tools/ab.sh VAR v1 v2 ... -- <bench args>: runs bench.py once per value of the env var on the SAME box
var=$1; shift
vals=()
while [ "$1" != "--" ]; do vals+=("$1"); shift; done
shift
for v in "${vals[@]}"; do
  env $var=$v timeout 300 python bench.py --no-cpu "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['ms_per_step'], d['value'])"
done
