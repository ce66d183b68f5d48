#!/bin/bash
# round 3, call 33: loss / norm reductions finished by the last workgroup: tests, step and epoch A/B
O=gpurun_out/r03c33; mkdir -p $O
python -m pytest tests/test_hip_round3.py -x -q -m gpu -k "fed_by_the_loss_kernel or sqnorm_with" > $O/new_tests.log 2>&1; echo "rc=$?" >> $O/new_tests.log; tail -n 6 $O/new_tests.log
python -m pytest tests -x -q -m gpu -k "loss or trajectory or captured or fused or adam or clip or hits" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
for i in 1 2; do
for f in 0 1; do
PLNLP_FUSE_LOSS_ACC=$f python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/bench_collab_f${f}_$i.json 2>/dev/null
python -c "
import json,sys; r=json.loads(open('$O/bench_collab_f${f}_$i.json').read().strip().splitlines()[-1]); print('collab loss_acc=$f', r['ms_per_step'], r['value'], 'epoch', r['train_epoch']['value'], r['train_epoch']['ms_per_step'], r['train_epoch']['loss'])"
done; done
