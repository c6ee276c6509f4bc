#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
python -m pytest tests/test_22_linear_ln.py -x -q -m gpu 2>&1 | tail -3
python tools/linear_ln_trace.py 2>&1 | grep -v amdgpu.ids | tee $O/trace.txt
python tools/linear_ln_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/linear_ln_bench.txt
