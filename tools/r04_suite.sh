#!/bin/bash
# full GPU suite: in file order, then shuffled with the given seeds; smoke(); tails appended to gpurun_out/<dir>/suite.txt
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04suite}; shift; mkdir -p $O
if [ "$NO_ORDER" != "1" ]; then
echo "== $(date -u +%FT%TZ) file order" >> $O/suite.txt
timeout 1200 python -m pytest tests -q -m gpu > $O/t_order.txt 2>&1; tail -3 $O/t_order.txt >> $O/suite.txt
fi
for seed in "$@"; do
  echo "== --shuffle $seed" >> $O/suite.txt
  timeout 1200 python -m pytest tests -q -m gpu --shuffle $seed > $O/t_$seed.txt 2>&1; tail -3 $O/t_$seed.txt >> $O/suite.txt
done
echo "== smoke" >> $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $O/suite.txt 2>&1
cat $O/suite.txt
