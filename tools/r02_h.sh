#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_c5_pipeline.py tests/test_deepspeaker.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r02h/new.txt
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_c5_pipeline.py --deselect tests/test_deepspeaker.py 2>&1 | tail -6 > gpurun_out/r02h/all.txt
cat gpurun_out/r02h/new.txt gpurun_out/r02h/all.txt
