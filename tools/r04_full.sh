#!/bin/bash
# full GPU suite + the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04full}; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1
tail -6 $O/tests.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
