#!/usr/bin/env python3
"""Every launch of the last step of a rocprofv3 --kernel-trace CSV, in order, with the gap to the previous launch.
usage: trace_all.py <kernel_trace.csv> [skip-substring]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = sys.argv[2] if len(sys.argv) > 2 else None
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'conv_post' in r['Kernel_Name']]
seg = rows[idx[-2] + 1: idx[-1] + 1]
prev_end = int(rows[idx[-2]]['End_Timestamp'])
tot = gap_tot = 0.0
for r in seg:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) / 1e3
    prev_end = en
    gap_tot += max(gap, 0)
    if skip and skip in r['Kernel_Name']:
        continue
    dur = (en - st) / 1e3
    tot += dur
    name = re.sub(r'^void vsp::|^vsp::|\(.*$', '', r['Kernel_Name'])[:60]
    wg = max(1, int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1))
    grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    print(f"{name:60s} blocks {grid // wg:7d}  {dur:9.1f} us  gap {gap:6.1f} us")
print(f'listed {tot / 1e3:.3f} ms; gaps over the whole step {gap_tot / 1e3:.3f} ms')
