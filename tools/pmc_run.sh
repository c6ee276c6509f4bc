#!/bin/bash
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> -- <command...>
# Runs the command under rocprofv3 three times (separate PMC passes, kernel-trace only) and dumps CSVs.
set -u
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d $R/gpurun_out/$out/sq -o p -- "$@" > $R/gpurun_out/$out/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/$out/tcc -o p -- "$@" > $R/gpurun_out/$out/tcc.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/$out/fetch -o p -- "$@" > $R/gpurun_out/$out/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/gpurun_out/$out/write -o p -- "$@" > $R/gpurun_out/$out/write.log 2>&1
cd $R
find gpurun_out/$out -name "*.csv" | head -20
