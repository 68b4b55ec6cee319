#!/usr/bin/env python3
"""Per-launch figures of the generator (fused ResBlock pairs on the 64/32-channel stages) from a
rocprofv3 --kernel-trace CSV.  usage: trace_fused.py <kernel_trace.csv> [B] [T_frames]"""
import csv
import sys

path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 489
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'conv_post' in r['Kernel_Name']]
seg = rows[idx[-2] + 1: idx[-1] + 1]
gen = [r for r in seg if any(k in r['Kernel_Name'] for k in ('g16_conv', 'g16_pair', 'g16_chain', 'g16_ups', 'g16_rw', 'g16_pp', 'g16_rc'))]
c0 = 512
rates = [8, 8, 4, 2]; uk = [16, 16, 4, 4]; ks = [3, 7, 11]
specs = []
t = T
for i in range(4):
    cin = c0 >> i; cout = c0 >> (i + 1); s = rates[i]
    specs.append((f'ups{i}', 'conv', cout, cin, uk[i], t, t * s, 0))
    t *= s
    for k in ks:
        for d in (1, 3, 5):
            if cout <= 64:
                specs.append((f's{i}k{k}d{d}', 'pair', cout, cout, k, t, t, 1))
            else:
                specs.append((f's{i}k{k}d{d}a', 'conv', cout, cout, k, t, t, 0))
                specs.append((f's{i}k{k}d1b', 'conv', cout, cout, k, t, t, 1))
# a g16_chain launch covers all dilation pairs of one ResBlock: merge their specs
merged = []
gi = 0
si = 0
while si < len(specs) and gi < len(gen):
    name, kind, co, ci, K, N, Nout, res = specs[si]
    if ('g16_chain' in gen[gi]['Kernel_Name'] or 'g16_rc' in gen[gi]['Kernel_Name']) and kind == 'pair':
        merged.append((name[:-2] + 'ch', 'chain', co, ci, K, N, Nout, res, 3))
        si += 3
    elif 'g16_pp' in gen[gi]['Kernel_Name'] and kind == 'conv' and name.endswith('a'):
        merged.append((name[:-1], 'pair', co, ci, K, N, Nout, 1, 1))     # one launch for the pair's two convolutions
        si += 2
    else:
        merged.append((name, kind, co, ci, K, N, Nout, res, 1))
        si += 1
    gi += 1
if len(gen) != len(merged) or si != len(specs):
    print(f"# note: {len(gen)} launches vs {len(merged)} expected")
tot = 0
stage = {}
for (name, kind, co, ci, K, N, Nout, res, npair), r in zip(merged, gen):
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if kind in ('pair', 'chain'):
        fl = npair * 2 * 2.0 * co * ci * K * N * B
        byt = npair * 4.0 * B * N * co * 5   # layer-boundary model of the convolutions the launch replaces (+ residual reads)
        real = 4.0 * B * N * co * (2 if kind == 'chain' else 3)   # x in, residual x (a chain: none), y out
    else:
        fl = 2.0 * co * ci * K * N * B
        byt = 4.0 * B * (N * ci + Nout * co * (1 + res))
        real = byt
    tot += dur
    key = name[:2] if name[0] == 's' else name
    stage[key] = stage.get(key, 0) + dur
    print(f"{name:10s} {kind:5s} C={co:4d} K={K:2d} rows={N:7d} {dur:7.3f} ms  alg {fl/dur/1e9:6.1f} TF/s  mfma-issue "
          f"{3*fl/dur/1e9/2500*100:5.1f}%  alg {byt/dur/1e6:7.1f} GB/s  moved>= {real/dur/1e6:7.1f} GB/s  vgpr={r['VGPR_Count']}")
print('per stage ms:', {k: round(v, 2) for k, v in stage.items()})
print(f'total {tot:.2f} ms')
