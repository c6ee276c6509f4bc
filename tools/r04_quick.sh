#!/bin/bash
# quick validation: model-level GPU tests + default-shape bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04q}; mkdir -p $O
timeout 900 python -m pytest tests/test_20_hip_backward.py tests/test_90_equivalences.py tests/test_92_model_equivalences.py tests/test_11_oracle_c2c3.py -x -q -m gpu > $O/tests.txt 2>&1
tail -4 $O/tests.txt
Q="--no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 2"
timeout 300 python bench.py $Q 2>>$O/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('repeat'))"
