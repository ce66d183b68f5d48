#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/c46; mkdir -p $R; rm -rf $R/prof
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -f csv -d $R/prof -o p -- python3 scripts/probe_gap.py > $R/run.log 2>&1
ls $R/prof/* | head
python3 - <<'PY'
import csv, glob
R="gpurun_out/c46/prof"
def load(pat):
    f=glob.glob(R+"/**/*"+pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k=load("kernel_trace.csv"); m=load("memory_copy_trace.csv"); h=load("hip_api_trace.csv")
print(len(k), len(m), len(h))
k.sort(key=lambda r:int(r['Start_Timestamp']))
# last adam_multi kernel followed by csr_agg_vec<...false...>
idx=[i for i,r in enumerate(k) if 'adam_multi' in r['Kernel_Name']]
i=idx[-3]
t_end=int(k[i]['End_Timestamp'])
nxt=[r for r in k[i+1:] if int(r['Start_Timestamp'])>=t_end][0]
t_next=int(nxt['Start_Timestamp'])
print("gap us", (t_next-t_end)/1e3, nxt['Kernel_Name'][:60])
print("memcopies in gap:")
for r in m:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if e>=t_end-20000 and s<=t_next+20000: print("  ", (s-t_end)/1e3, (e-s)/1e3, r.get('Direction'), r.get('Bytes', r.get('Size')))
print("HIP calls overlapping the gap (start-rel us, dur us, name):")
for r in h:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if e>=t_end-30000 and s<=t_next+5000: print("  ", round((s-t_end)/1e3,1), round((e-s)/1e3,1), r['Function'])
PY
rm -rf $R/prof
