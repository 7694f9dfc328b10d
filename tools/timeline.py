"""Timeline of ONE steady-state step from a rocprofv3 `--kernel-trace -f csv` run of bench.py --eager: which queue (HIP stream) every
kernel ran on, when, and the idle gaps of the device -- the critical-path view that the per-kernel sums of summarize_trace_csv.py
cannot give (two streams overlap in the tail of the step).
usage: python tools/timeline.py <dir-or-kernel_trace.csv> [step_from_the_end=1] > profiles/xxx_timeline.md"""
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n).replace('odtk::cv::', '').replace('odtk::', '')
    return n.split('(')[0][:60]


src = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
rows = []
for f in files:
    with open(f, newline='') as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', r.get('Stream_Id', '?')),
                         int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith('preprocess')]
assert len(starts) > back, f"only {len(starts)} steps in the trace"
lo, hi = starts[-back - 1], starts[-back]
step = rows[lo:hi]
t0 = step[0][0]
t1 = max(r[1] for r in step)
queues = sorted({r[3] for r in step})
print(f"step of {len(step)} launches, first start -> last end {(t1 - t0) / 1e3:.1f} us, next step starts at {(rows[hi][0] - t0) / 1e3:.1f} us; queues {queues}\n")
for q in queues:
    b = sum(r[1] - r[0] for r in step if r[3] == q)
    print(f"queue {q}: {sum(1 for r in step if r[3] == q)} launches, busy {b / 1e3:.1f} us")
# device idle: union of the busy intervals
ev = sorted((r[0], r[1]) for r in step)
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((cur_e, s))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"\ndevice busy (union over queues) {busy / 1e3:.1f} us, idle inside the step {sum(g[1] - g[0] for g in gaps) / 1e3:.1f} us in {len(gaps)} gaps "
      f"(>= 2 us: {sum(1 for g in gaps if g[1] - g[0] >= 2000)}, sum {sum(g[1] - g[0] for g in gaps if g[1] - g[0] >= 2000) / 1e3:.1f} us)\n")
print("| start us | dur us | gap before us | queue | blocks | kernel |\n|---|---|---|---|---|---|")
last_end = {}
prev_any = t0
for s, e, k, q, g in step:
    gap = (s - prev_any) / 1e3
    print(f"| {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} | {q} | {g} | `{k}` |")
    prev_any = max(prev_any, e)
