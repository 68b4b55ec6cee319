#!/usr/bin/env python3
"""Per-launch figures of the split-f16 generator from a rocprofv3 --kernel-trace CSV.
usage: trace_cl.py <kernel_trace.csv> [B] [T_frames] [--brief]"""
import csv
import sys

path = sys.argv[1]
args = [a for a in sys.argv[2:] if not a.startswith('--')]
brief = '--brief' in sys.argv
B = int(args[0]) if len(args) > 0 else 64
T = int(args[1]) if len(args) > 1 else 489
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'conv_post' in r['Kernel_Name']]
seg = rows[idx[-2] + 1: idx[-1] + 1]
gen = [r for r in seg if 'cl_' in r['Kernel_Name'] and 'conv_post' not in r['Kernel_Name']]
c0 = 512
specs = []
rates = [8, 8, 4, 2]; uk = [16, 16, 4, 4]; ks = [3, 7, 11]
t = T
for i in range(4):
    cin = c0 >> i; cout = c0 >> (i + 1); s = rates[i]; kt = uk[i] // s
    specs.append((f'ups{i}', cout, cin, kt * s, t, t * s, 0))
    t *= s
    for k in ks:
        for d in (1, 3, 5):
            specs.append((f's{i}k{k}d{d}a', cout, cout, k, t, t, 0))
            specs.append((f's{i}k{k}d1b', cout, cout, k, t, t, 1))
if len(gen) != len(specs):
    print(f"# note: {len(gen)} launches vs {len(specs)} expected (fused kernels?)")
tot = 0
stage = {}
for (name, co, ci, K, N, Nout, res), r in zip(specs, gen):
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    fl = 2.0 * co * ci * K * N * B
    byt = 4.0 * B * (N * ci + Nout * co * (1 + res))
    tot += dur
    stage[name[:2] if name[0] == 's' else name] = stage.get(name[:2] if name[0] == 's' else name, 0) + dur
    if not brief:
        tile = r['Kernel_Name'].split('<')[1].split('>')[0] if '<' in r['Kernel_Name'] else ''
        print(f"{name:10s} <{tile}> Co={co:4d} Ci={ci:4d} K={K:2d} N={N:7d} {dur:7.3f} ms alg {fl/dur/1e9:6.1f} TF/s "
              f"mfma-issue {3*fl/dur/1e9/2500*100:5.1f}%  {byt/dur/1e6:7.1f} GB/s vg={r['VGPR_Count']} "
              f"av={r['Accum_VGPR_Count']} lds={r['LDS_Block_Size']}")
print('per stage ms:', {k: round(v, 2) for k, v in stage.items()})
print(f'total {tot:.2f} ms')
