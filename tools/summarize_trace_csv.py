"""Summarise a rocprofv3 `--kernel-trace -f csv` run into a markdown table.
usage: python tools/summarize_trace_csv.py <dir-or-kernel_trace.csv> <steps_in_trace> > profiles/xxx.md"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = n.replace('unsigned short', 'bf16')
    return n.split('(')[0][:80]


src = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
per = defaultdict(list)
grid = defaultdict(list)
meta = {}
for f in files:
    with open(f, newline='') as fh:
        for r in csv.DictReader(fh):
            k = short(r['Kernel_Name'])
            d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            per[k].append(d)
            if 'conv_' in k:
                gx = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
                gy = int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1)
                grid[(k, gx, gy)].append(d)
                meta[(k, gx, gy)] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Workgroup_Size_X'])
tot = sum(sum(v) for v in per.values())
print(f"total kernel time {tot/1e6:.3f} ms over {steps:g} steps = {tot/steps/1e6:.3f} ms/step\n")
print("| kernel | launches/step | ms/step | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"| `{k}` | {len(v)/steps:.1f} | {sum(v)/steps/1e6:.3f} | {sum(v)/len(v)/1e3:.1f} | {min(v)/1e3:.1f} | {max(v)/1e3:.1f} | {100*sum(v)/tot:.1f} |")
print("\nconv launches by grid shape\n\n| kernel | grid (blocks x, y) | wg | launches/step | avg us | vgpr | agpr | lds B |\n|---|---|---|---|---|---|---|---|")
for key, v in sorted(grid.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))[:48]:
    k, gx, gy = key
    m = meta[key]
    print(f"| `{k}` | {gx} x {gy} | {m[3]} | {len(v)/steps:.1f} | {sum(v)/len(v)/1e3:.1f} | {m[0]} | {m[1]} | {m[2]} |")
