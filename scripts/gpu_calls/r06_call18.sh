#!/bin/bash
# round 6, call 18: the whole -m gpu suite and smoke() on the current tree
mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r06/call18_suite.txt; cat gpurun_out/r06/call18_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
