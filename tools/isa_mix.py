#!/usr/bin/env python3
"""Instruction mix per kernel of a device assembly listing (hipcc -S --cuda-device-only x.hip -o x.s), over the WHOLE function
(label .. .Lfunc_end: a kernel has several s_endpgm when it has early exits).
usage: tools/isa_mix.py <x.s> <substring of the demangled kernel name> ..."""
import collections
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):', txt, flags=re.M):
    n = m.group(1)
    dn = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    if not any(w in dn for w in sys.argv[2:]):
        continue
    end = txt.find('.Lfunc_end', m.end())
    lines = [l.strip() for l in txt[m.end():end].split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
    cnt = collections.Counter('mfma' if 'mfma' in l.split()[0] else l.split()[0] for l in lines)
    nv = sum(v for k, v in cnt.items() if k.startswith('v_'))
    print(f"{dn[:80]}: {len(lines)} instructions, {nv} vector (non-MFMA), {cnt.get('mfma', 0)} MFMA")
    for k, v in cnt.most_common(24):
        print('   ', v, k)
