#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "batchnorm or postnet or bn or train_step or c3" 2>&1 | tail -2
run() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && env "$@" timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/nb_$tag -o t -- python $R/tools/norm_bench.py > $R/gpurun_out/nb_$tag.log 2>&1 ); db=$(find $R/gpurun_out/nb_$tag -name "*.db" | head -1); python tools/prof_summary.py $db gpurun_out/nb_${tag}_stats.txt > /dev/null; echo "== $tag $@"; grep -E "^BN|^LN" gpurun_out/nb_$tag.log; grep -E "bn_|layernorm" gpurun_out/nb_${tag}_stats.txt; rm -rf gpurun_out/nb_$tag; }
run base A=1
run rpb256 STYLER_BN_RPB=256
run rpb64 STYLER_BN_RPB=64
for i in 1 2; do python bench.py --no-cpu --no-aux --steps 40 --warmup 5 --prof-steps 0 --repeat 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['repeat']['ms_per_step_median'])"; done
