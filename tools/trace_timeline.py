#!/usr/bin/env python3
"""Timeline of ONE step from a rocprofv3 kernel trace: per launch start offset, duration and the idle gap before it.
usage: trace_timeline.py t_kernel_trace.csv [launches_per_step | 0 = find the period by the last conv_post kernel]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
# a step ends with the generator's conv_post kernel
ends = [i for i, r in enumerate(rows) if "conv_post" in name(r)]
if len(ends) < 2:
    sys.exit("need at least two steps in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
busy = 0
def short(n):
    n = n.replace("void vsp::", "").replace("vsp::", "")
    if n.startswith("_ZN3vsp"):
        n = n[7:]
        k = 0
        while n[k].isdigit(): k += 1
        n = n[k:k + int(n[:k])]
    return n.split("(")[0][:44]
cls = {}
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  {short(name(r))}  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')}")
    busy += e - s
    c = cls.setdefault(short(name(r)), [0, 0, 0])
    c[0] += 1; c[1] += e - s; c[2] += max(gap, 0)
    prev_end = max(prev_end, e)
span = prev_end - t0
print(f"\nlaunches {len(step)}  span {span / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {(span - busy) / 1e3:.1f} us")
for k, (n, d, g) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:46s} n {n:3d}  dur {d / 1e3:8.1f} us  gaps-before {g / 1e3:7.1f} us")
