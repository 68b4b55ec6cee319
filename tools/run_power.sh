#!/bin/bash
# sample board power / clocks with rocm-smi while the bench loops; usage: tools/run_power.sh [ENV=val ...]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
O=gpurun_out/power; mkdir -p $O
python bench.py --steps 150 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | head -c 1500; echo; sleep 1; done > $O/smi.txt
wait $BP
cut -c1-200 $O/bench.json
