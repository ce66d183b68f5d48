#!/bin/bash
# after the wide weight-gradient kernel: the whole GPU suite, then citation2's line + step breakdown + the kernel's microbench / counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r05q; mkdir -p $R
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpu_suite.txt 2>&1; echo "suite exit $?" >> $R/gpu_suite.txt; tail -4 $R/gpu_suite.txt
python bench.py --workload citation2 --steps 10 --warmup 5 --no-parity --no-stress --cpu-steps 1 > $R/bench_citation2.json 2>/dev/null; tail -c 400 $R/bench_citation2.json
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --workload citation2 --steps 12 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 6 45 sequence > $R/step_breakdown_citation2.txt
rm -rf $R/prof
head -12 $R/step_breakdown_citation2.txt
python scripts/bench_gemm.py --shapes cit_in_wgrad,cit_l2_wgrad,wgrad_224 --math wide --error --iters 10 > $R/gemm_wide_microbench.jsonl 2>/dev/null
OUT=$R/gemm_wide_pmc.json SH=cit_l2_wgrad bash scripts/pmc_gemm_wide.sh > /dev/null 2>&1
ls -la $R
