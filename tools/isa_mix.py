#!/usr/bin/env python3
"""Instruction mix per kernel of a device assembly listing (hipcc -S --cuda-device-only x.hip -o x.s).
usage: tools/isa_mix.py <x.s> <substring of the demangled kernel name> ..."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
names = re.findall(r'^(_Z\w+):', txt, flags=re.M)
for n in names:
    dn = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    if not any(w in dn for w in sys.argv[2:]):
        continue
    start = txt.index('\n' + n + ':')
    end = txt.index('s_endpgm', start)
    lines = [l.strip() for l in txt[start:end].split('\n')[2:] if l.strip() and not l.strip().startswith(('.', ';'))]
    cnt = {}
    for l in lines:
        op = l.split()[0]
        cnt['mfma' if 'mfma' in op else op] = cnt.get('mfma' if 'mfma' in op else op, 0) + 1
    nv = sum(v for k, v in cnt.items() if k.startswith('v_'))
    print(f"{dn[:80]}: {len(lines)} instructions, {nv} vector (non-MFMA), {cnt.get('mfma', 0)} MFMA")
    for k, v in sorted(cnt.items(), key=lambda x: -x[1])[:24]:
        print('   ', v, k)
