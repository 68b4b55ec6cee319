#!/bin/bash
# sample board power / clocks with rocm-smi while the bench loops; usage: tools/run_power.sh [ENV=val ...]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
O=gpurun_out/power; mkdir -p $O
python bench.py --steps 400 --warmup 3 --no-cpu-baseline --profile-steps 0 > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 12
for i in $(seq 1 20); do rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | head -c 1500; echo; sleep 0.5; done > $O/smi.txt
wait $BP
cut -c1-200 $O/bench.json
python - <<EOF
import json
for ln in open("$O/smi.txt"):
    ln=ln.strip()
    if not ln.startswith("{"): continue
    try: d=json.loads(ln)
    except Exception: continue
    for k,v in d.items():
        print({kk.split(" (")[0][:28]:vv for kk,vv in v.items() if any(t in kk.lower() for t in ("power","sclk","use"))})
EOF
