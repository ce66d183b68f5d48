#!/bin/bash
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
python -c "import torch; print('torch threads', torch.get_num_threads(), torch.get_num_interop_threads())"
echo "== default"; python scripts/probe_epoch.py 2>&1 | tail -12 | awk '{print $1,$2,$3,$4}' | tr '\n' ';'; echo
echo "== OMP_NUM_THREADS=8"; OMP_NUM_THREADS=8 python scripts/probe_epoch.py 2>&1 | tail -12 | awk '{print $1,$2,$3,$4}' | tr '\n' ';'; echo
echo "== OMP_WAIT_POLICY=passive"; OMP_WAIT_POLICY=passive python scripts/probe_epoch.py 2>&1 | tail -12 | awk '{print $1,$2,$3,$4}' | tr '\n' ';'; echo
