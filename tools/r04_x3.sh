#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04x3; mkdir -p $O
timeout 900 python -m pytest tests/test_11_oracle_c2c3.py tests/test_92_model_equivalences.py -x -q -m gpu > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python $R/bench.py --prec bf16x3 --no-cpu --no-aux --no-hbm --prof-steps 0 --repeat 0 --mode train --steps 10 --warmup 3 > $O/tr.log 2>&1
db=$(find $O/tr -name "*.db" | head -1)
python $R/tools/prof_summary.py $db $O/kernel_stats_x3.txt > /dev/null
python $R/tools/step_listing.py $db --step 6 > $O/step_listing_x3.txt 2>&1
rm -rf $O/tr
tail -1 $O/tr.log | cut -c1-200
cd $R
python -c "
import json; d=json.load(open('gpurun_out/parity_report.json'))
for k, v in d.items():
    if 'x3' in k and 'train' in k: print(k, 'median', v['median_tensor_err'], 'worst', v['worst_tensors'][:3], 'gn', v['grad_norm'], v['ref_grad_norm'], max(v['loss_rel_err']))
    elif 'x3' in k: print(k, {a: b['max_abs'] for a, b in v.items()})
"
