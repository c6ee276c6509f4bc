#!/bin/bash
# Builds styler_amd/libstyler_hip_trace.so = the product library with gemm_bwd.hip compiled -DSTYLER_WGRAD_TRACE (run HERE, in the
# build container; the .so travels to the GPU box), for tools/wgrad_trace.py.
set -e
cd "$(dirname "$0")/../styler_amd/csrc"
make -s
mkdir -p build/trace
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -munsafe-fp-atomics -DSTYLER_WGRAD_TRACE -c gemm_bwd.hip -o build/trace/gemm_bwd.o
objs=$(ls build/*.o | grep -v "build/gemm_bwd.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libstyler_hip_trace.so $objs build/trace/gemm_bwd.o
ls -la ../libstyler_hip_trace.so
