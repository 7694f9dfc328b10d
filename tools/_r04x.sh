#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r04x; mkdir -p $O; export TMPDIR=/tmp
BCMD="python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-conv-events --eager"
ODTK_LIB=$R/tools/probes/bin/libodtk_old.so timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace_old -- $BCMD > $O/t_old.log 2>&1
python tools/summarize_trace_csv.py $O/trace_old 9 > $O/ssd_old.md; rm -rf $O/trace_old
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace_new -- $BCMD > $O/t_new.log 2>&1
python tools/summarize_trace_csv.py $O/trace_new 9 > $O/ssd_new.md; rm -rf $O/trace_new
python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        m=re.match(r'\| `(.*?)` \| ([\d.]+) \| ([\d.]+) \|',l)
        if m: d[m.group(1)[:70]]=(float(m.group(2)),float(m.group(3)))
    return d
a=load('gpurun_out/r04x/ssd_old.md'); b=load('gpurun_out/r04x/ssd_new.md')
print(open('gpurun_out/r04x/ssd_old.md').readline().strip()); print(open('gpurun_out/r04x/ssd_new.md').readline().strip())
rows=[]
for k in set(a)|set(b):
    x=a.get(k,(0,0)); y=b.get(k,(0,0))
    rows.append((y[1]-x[1],k,x,y))
rows.sort(key=lambda r:-abs(r[0]))
for r in rows[:16]: print('%+.3f ms  %-72s old %s new %s'%(r[0],r[1],r[2],r[3]))
PY
