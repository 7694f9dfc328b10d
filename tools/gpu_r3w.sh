#!/bin/bash
set -u
TAG=${1:-r03w}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $O/trace -- python tools/step_drift.py ssd300 3 10 > $O/trace.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$O/trace/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for k in ('nms_kernel','nms_matrix','nms_scan','nms_topk','ssd_loss_kernel'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if k in r['Kernel_Name']]
    print(k, len(d), ' '.join(f'{x:.0f}' for x in d))
steps=[];cur=None
for r in rows:
    n=r['Kernel_Name']
    if 'preprocess_kernel' in n:
        cur=collections.Counter(); steps.append(cur)
    if cur is not None:
        k=n.replace('void ','').split('(')[0].split('<')[0].split('::')[-1]
        cur[k]+= (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
a,b=steps[5],steps[-2]
d=sorted(((a[k]-b[k],k,a[k],b[k]) for k in set(a)|set(b)),key=lambda t:-abs(t[0]))[:8]
print('step 5 vs step', len(steps)-2)
for x in d: print('%+.0f us  %s  %.0f vs %.0f'%x)
PY
rm -rf $O/trace
