#!/bin/bash
# kernel trace of one bench run on the GPU box -> gpurun_out/<tag>/ ; usage: tools/run_trace.sh <tag> [env=val ...]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
: "${1:?usage: run_trace.sh <tag> [ENV=val ...]}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; TAG="$1"; shift
for kv in "$@"; do export "$kv"; done
O="$R/gpurun_out/$TAG"
rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o t -- python3 "$R/bench.py" --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" || true
tail -1 "$O/bench.json" | cut -c1-300
python3 "$R/tools/trace_fused.py" "$(find "$O" -name '*kernel_trace.csv' | head -1)" > "$O/per_launch.txt"
tail -2 "$O/per_launch.txt"
