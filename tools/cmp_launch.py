#!/usr/bin/env python3
"""side-by-side per-launch durations of several tools/trace_fused.py tables: cmp_launch.py a/per_launch.txt b/per_launch.txt ..."""
import re, sys
tabs = []
for p in sys.argv[1:]:
    rows = []
    for ln in open(p):
        m = re.match(r"(\S+)\s+(conv|pair).*?([\d.]+) ms", ln)
        if m:
            rows.append((m.group(1), float(m.group(3))))
    tabs.append(rows)
for i in range(len(tabs[0])):
    print(f"{tabs[0][i][0]:10s} " + " ".join(f"{t[i][1]:7.3f}" for t in tabs if i < len(t)))
print("total      " + " ".join(f"{sum(r[1] for r in t):7.2f}" for t in tabs))
