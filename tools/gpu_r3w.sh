#!/bin/bash
set -u
TAG=${1:-r03w}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
BCMD="python bench.py --steps 60 --warmup 4 --no-cpu-baseline --no-conv-events --eager"
timeout 300 rocprofv3 --kernel-trace -f csv -d $O/trace -- $BCMD > $O/trace.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$O/trace/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# split into steps by preprocess kernel
steps=[];cur=None
for r in rows:
    n=r['Kernel_Name']
    if 'preprocess_kernel' in n:
        cur=collections.Counter(); steps.append(cur)
    if cur is not None:
        k=n.split('(')[0].split('::')[-1][:40]
        cur[k]+= (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
print(len(steps),'steps')
a,b=steps[6],steps[-3]
tot=lambda s:sum(s.values())
print('kernel time step 6: %.0f us, step %d: %.0f us'%(tot(a),len(steps)-3,tot(b)))
d=sorted(((b[k]-a[k],k,a[k],b[k]) for k in set(a)|set(b)),key=lambda t:-abs(t[0]))[:12]
for x in d: print('%+.0f us  %s  %.0f -> %.0f'%x)
PY
rm -rf $O/trace
