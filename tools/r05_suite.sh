#!/bin/bash
# full GPU suite: in file order, then shuffled with the given seeds; smoke(); tails appended to gpurun_out/<dir>/suite.txt
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r05suite}; shift; mkdir -p $O
echo "#### lease $(basename $O) $(date -u +%FT%TZ) $(git rev-parse --short HEAD 2>/dev/null)" >> $O/suite.txt
if [ "$NO_ORDER" != "1" ]; then
echo "== file order (pytest tests -x -q -m gpu: the driver's command)" >> $O/suite.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_order.txt 2>&1; tail -3 $O/t_order.txt >> $O/suite.txt
fi
for seed in "$@"; do
  echo "== --shuffle $seed" >> $O/suite.txt
  timeout 1500 python -m pytest tests -q -m gpu --shuffle $seed > $O/t_$seed.txt 2>&1; tail -3 $O/t_$seed.txt >> $O/suite.txt
  grep -E "^FAILED|^ERROR" $O/t_$seed.txt >> $O/suite.txt
done
echo "== smoke" >> $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids >> $O/suite.txt
cat $O/suite.txt
