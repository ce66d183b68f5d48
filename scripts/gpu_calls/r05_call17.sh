#!/bin/bash
# round 5, call 17: the wide trained-parity legs against the regenerated g12 fixture (collab_wide 16 seeds, ddi_wide at its plateau)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c17; mkdir -p $O
rm -f gpurun_out/trained_parity_wide_r05.txt
timeout 2400 python -m pytest tests/test_hip_round5.py -q -m gpu -s -k "trained_regime_parity" > $O/wide.txt 2>&1; grep -v amdgpu.ids $O/wide.txt | grep -v "^$" | tail -30 | cut -c1-700
