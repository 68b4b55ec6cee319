#!/usr/bin/env python3
"""Pivot rocprofv3 --pmc counter_collection.csv files into one row per dispatch of a kernel family.
usage: pmc_table.py <substr> <counter_collection.csv> [more.csv ...]"""
import csv, sys, collections
sub = sys.argv[1]
tables = []
for path in sys.argv[2:]:
    rows = list(csv.DictReader(open(path)))
    d = collections.OrderedDict()
    for r in rows:
        if sub not in r['Kernel_Name']:
            continue
        key = int(r['Dispatch_Id'])
        d.setdefault(key, {'name': r['Kernel_Name'], 'grid': r.get('Grid_Size', ''), 'wg': r.get('Workgroup_Size', '')})
        d[key][r['Counter_Name']] = float(r['Counter_Value'])
    tables.append(list(d.values()))
n = min(len(t) for t in tables)
for i in range(n):
    row = {}
    for t in tables:
        row.update(t[i])
    tile = row['name'].split('<')[1].split('>')[0] if '<' in row['name'] else ''
    keys = [k for k in row if k not in ('name', 'grid', 'wg')]
    wc = row.get('SQ_WAVE_CYCLES', 0) or 1
    out = [f"{i:3d} <{tile}> grid={row['grid']}"]
    for k in keys:
        v = row[k]
        if k.startswith('SQ_WAIT') or k.startswith('SQ_ACTIVE') :
            out.append(f"{k[3:]}={v/wc*100:.0f}%")
        else:
            out.append(f"{k[3:]}={v:.3g}")
    print(' '.join(out))
