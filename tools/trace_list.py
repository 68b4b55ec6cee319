#!/usr/bin/env python3
"""Generator launches of the last step of a rocprofv3 --kernel-trace CSV, in order: kernel, grid, duration.
usage: trace_list.py <kernel_trace.csv> [substring]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else 'g16_'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'conv_post' in r['Kernel_Name']]
seg = rows[idx[-2] + 1: idx[-1] + 1]
tot = 0.0
for r in seg:
    if want not in r['Kernel_Name']:
        continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    tot += dur
    name = re.sub(r'^void vsp::|\(.*$', '', r['Kernel_Name'])
    print(f"{name:28s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):7d} x {r['Workgroup_Size_X']:>4s}  {dur:8.3f} ms  vgpr {r['VGPR_Count']} lds {r['LDS_Block_Size']}")
print(f'total {tot:.2f} ms')
