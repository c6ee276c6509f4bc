#!/bin/bash
# usage: tools/ab_lib.sh <lib_a.so> <lib_b.so> [rounds] -- <bench args>: the default bench line with two builds of libstyler_hip.so
# (STYLER_LIB) on the SAME box, alternating; prints ms_per_step of every run.  Build the "before" library from a commit with
#   git archive <commit> styler_amd/csrc include | tar -x -C /tmp/prev && make -C /tmp/prev/styler_amd/csrc -j8
# and copy it next to the current one (styler_amd/libstyler_hip_prev.so: *.so is git-ignored but travels to the GPU box).
a=$1; b=$2; shift 2
rounds=2
if [ "$1" != "--" ]; then rounds=$1; shift; fi
shift
for r in $(seq $rounds); do
  for lib in $a $b; do
    STYLER_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['repeat']['ms_per_step_median'])"
  done
done
