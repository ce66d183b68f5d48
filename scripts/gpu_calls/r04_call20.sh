#!/bin/bash
# round 4, call 20: the whole GPU suite on the final tree
O=$GRAFT_REPO_ROOT/gpurun_out/r04c20; mkdir -p $O
timeout 2400 python -m pytest tests -q -x -m gpu --durations=15 > $O/tests.log 2>&1
echo "tests rc=$?"; tail -25 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
