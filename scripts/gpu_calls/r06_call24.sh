#!/bin/bash
# round 6, call 24: what bounds the segment backward at citation2's shape -- times per form, then counters (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python scripts/probe_segment_bwd.py | tee $O/call24_times.txt
rm -rf $O/pmc24
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  d=$O/pmc24/$(echo $pass | tr ' ' '_')
  PROBE_ITERS=3 rocprofv3 --kernel-trace --pmc $pass -f csv -d $d -o s -- python3 scripts/probe_segment_bwd.py > /dev/null 2>$d.err || tail -3 $d.err
done
python3 scripts/pmc_collect.py edge_segment $O/call24_segment_pmc.json "$O/pmc24/**/*counter_collection.csv" > /dev/null
rm -rf $O/pmc24
cat $O/call24_segment_pmc.json
