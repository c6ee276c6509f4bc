#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
timeout 900 python -m pytest tests/test_deepspeaker.py tests/test_hip_backward.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r02i/t.txt
cat gpurun_out/r02i/t.txt
