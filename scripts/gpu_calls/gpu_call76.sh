#!/bin/bash
s=$(date +%s); python bench.py > gpurun_out/bench_threads.json 2> gpurun_out/bench_threads.err; e=$(date +%s); echo "bench seconds: $((e-s))"; tail -2 gpurun_out/bench_threads.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_threads.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['host_enqueue_ms_per_step']); print(d['cpu_baseline'])"
