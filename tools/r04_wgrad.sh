#!/bin/bash
# round-4: wgrad DMA ring A/B (micro-bench, parity tests, in-step A/B of the three modes)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04wg; mkdir -p $O; rm -f $O/ab.txt
timeout 300 python tools/wgrad_bench.py 5 > $O/wgrad_bench.txt 2>&1
timeout 600 python -m pytest tests/test_20_hip_backward.py tests/test_91_bf16_acts.py tests/test_92_model_equivalences.py -x -q -m gpu > $O/tests.txt 2>&1
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
for kv in STYLER_WGRAD_DMA=0 STYLER_WGRAD_DMA=1 STYLER_WGRAD_DMA=2 STYLER_WGRAD_DMA=1 STYLER_WGRAD_DMA=2; do
  echo "== $kv" >> $O/ab.txt
  env $kv timeout 300 python bench.py $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))" >> $O/ab.txt
done
cat $O/wgrad_bench.txt; tail -5 $O/tests.txt; cat $O/ab.txt
