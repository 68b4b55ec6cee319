#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of one HIP source (compile only, no GPU needed).
usage: tools/kernel_regs.py vispeech_amd/csrc/gen16.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", *extra, "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: (.*) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("vsp::", "").split("(")[0]
    g = lambda k: r.get(k, "?")
    print(f"{dn:62s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} sgpr {g('SGPRs'):>3s} spill v{g('VGPRs Spill')}/s{g('SGPRs Spill')} "
          f"occ {g('Occupancy [waves/SIMD]')} lds {g('LDS Size [bytes/block]')}")
