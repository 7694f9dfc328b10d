"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a markdown table.
usage: python tools/summarize_rocpd.py <results.db> <steps_in_trace> > profiles/xxx.md"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = n.replace('unsigned short', 'bf16')
    return n.split('(')[0][:80]
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {steps:g} steps = {tot/steps/1e6:.3f} ms/step\n")
print("| kernel | launches/step | ms/step | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
for r in rows[:40]:
    print(f"| `{short(r[0])}` | {r[1]/steps:.1f} | {r[2]/steps/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | {100*r[2]/tot:.1f} |")
print("\nconv launches by grid shape\n\n| kernel | grid (blocks x, y) | launches/step | avg us | vgpr | agpr | lds B |\n|---|---|---|---|---|---|---|")
rows = list(cur.execute("select name, grid_x/workgroup_x, grid_y, count(*), avg(end-start), vgpr_count, accum_vgpr_count, lds_size from kernels where name like '%conv_%' group by name, grid_x, grid_y order by 5 desc"))
for r in rows[:40]:
    print(f"| `{short(r[0])}` | {r[1]} x {r[2]} | {r[3]/steps:.1f} | {r[4]/1e3:.1f} | {r[5]} | {r[6]} | {r[7]} |")
