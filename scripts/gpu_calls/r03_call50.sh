#!/bin/bash
# round 3, call 50: the default line and the step breakdown once more on the final code
R=gpurun_out/r03p; mkdir -p $R
python bench.py --steps 20 --warmup 5 > $R/bench_collab.json 2> $R/bench_collab.err
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o collab -- python3 bench.py --steps 20 --warmup 5 > $R/bench_collab_under_rocprof.json 2>/dev/null
f=$(find $R/prof -name "*kernel_stats.csv" | head -1); cp $f $R/bench_collab_rocprofv3_kernel_stats.csv
f=$(find $R/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_calls.py $f "csr_agg_vec_kernel<1, 32, false" 20 > $R/roofline_kernel_calls.txt
python scripts/kernel_calls.py $f "csr_agg_fused_kernel<1, 64, 1, 64, true" 20 >> $R/roofline_kernel_calls.txt
rm -rf $R/prof
rocprofv3 --kernel-trace --stats -f csv -d $R/prof -o step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > /dev/null 2>&1
f=$(find $R/prof -name "*kernel_trace.csv" | head -1); python scripts/step_profile.py $f 10 45 sequence > $R/step_breakdown_collab.txt
rm -rf $R/prof
python -c "
import json
r=json.loads(open('$R/bench_collab.json').read().strip().splitlines()[-1]); u=json.loads(open('$R/bench_collab_under_rocprof.json').read().strip().splitlines()[-1])
print(r['ms_per_step'], r['value'], r['ms_per_step_f32_mfma'], r['ms_per_step_full_forward'], r['host_busy_ms_per_step'])
print(r['train_epoch']['value'], r['train_epoch']['epoch_s'], r['eval_scoring']['value'], r['eval_scoring']['ms'])
print(r['step_capture']['other_path']['ms_per_step'], r['step_capture']['other_path']['host_busy_ms_per_step'])
for k in ('roofline','roofline_workload_agg','roofline_agg_adam','roofline_mfma'):
    v=r[k]; print(k, v['kernel_ms'], v['achieved'], v['frac'], v.get('f32_equivalent_TFLOPs'), (v.get('kernel_form') or '')[:70])
print('cpu', r['cpu_baseline']['value'], 'under rocprof', u['roofline']['kernel_ms'], u['ms_per_step'])
"
sed -n 3p $R/roofline_kernel_calls.txt; head -1 $R/step_breakdown_collab.txt
