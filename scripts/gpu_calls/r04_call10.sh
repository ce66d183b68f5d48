#!/bin/bash
# round 4, call 10: the trained-regime legs on EVERY seed of the oracle fixture (64 / 48 per GEMM form) + the mutation leg
O=$GRAFT_REPO_ROOT/gpurun_out/r04c10; mkdir -p $O
rm -f gpurun_out/trained_parity_r04.txt
( time PLNLP_PARITY_SEEDS=full timeout 3000 python -m pytest tests/test_hip_round4.py -q -k trained --durations=6 ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -25 $O/tests.log
cat gpurun_out/trained_parity_r04.txt
