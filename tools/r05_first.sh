#!/bin/bash
# round 5, first lease: the 300-seed flake hunt of the round-4 red test, then the whole GPU suite in the new file order
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05first; mkdir -p $O
timeout 900 python tools/flake_hunt.py --seeds 300 > $O/flake_hunt.txt 2>&1; echo "flake_hunt rc=$?" >> $O/flake_hunt.txt
tail -5 $O/flake_hunt.txt
timeout 1200 python -m pytest tests -x -q -m gpu > $O/t_order.txt 2>&1; tail -5 $O/t_order.txt
timeout 600 python bench.py > $O/bench.txt 2>&1; tail -c 600 $O/bench.txt
