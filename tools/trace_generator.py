#!/usr/bin/env python3
"""Per-launch TFLOP/s of the generator's conv launches from a rocprofv3 --kernel-trace CSV.
usage: trace_generator.py <kernel_trace.csv> [B] [T_frames]"""
import csv
import sys

path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 489
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'conv_post' in r['Kernel_Name']]
seg = rows[idx[-2] + 1: idx[-1] + 1]
gen = [r for r in seg if 'conv1d' in r['Kernel_Name'] or 'resblock' in r['Kernel_Name'] or 'ups' in r['Kernel_Name']]
gen = gen[-77:]
c0 = 512
specs = [('pre', 512, 192, 7, T)]
rates = [8, 8, 4, 2]; uk = [16, 16, 4, 4]; ks = [3, 7, 11]
t = T
for i in range(4):
    cin = c0 >> i; cout = c0 >> (i + 1); s = rates[i]; kt = uk[i] // s
    specs.append((f'ups{i}', cout * s, cin, kt, t + 1))
    t *= s
    for k in ks:
        for d in (1, 3, 5):
            specs.append((f's{i}k{k}d{d}a', cout, cout, k, t))
            specs.append((f's{i}k{k}d1b', cout, cout, k, t))
tot = 0; totf = 0
for (name, M, Cin, K, N), r in zip(specs, gen):
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    fl = 2.0 * M * Cin * K * N * B
    tot += dur; totf += fl
    tile = r['Kernel_Name'].split('<')[1].split('>')[0] if '<' in r['Kernel_Name'] else r['Kernel_Name'][:20]
    byt = (Cin + M) * N * B * 4
    print(f"{name:12s} <{tile}> M={M:5d} Cin={Cin:4d} K={K:2d} N={N:7d} {dur:8.3f} ms {fl/dur/1e9:7.1f} TF/s "
          f"{byt/dur/1e6:7.1f} GB/s vg={r['VGPR_Count']} av={r['Accum_VGPR_Count']} lds={r['LDS_Block_Size']}")
print(f"total {tot:.2f} ms, {totf/tot/1e9:.1f} TF/s")
