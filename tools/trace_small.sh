#!/bin/bash
# launch-by-launch timeline of one step at a small operating point (one utterance, or rank 0's slice at N ranks);
# usage (through gpurun): tools/trace_small.sh <tag> <one|N> [ENV=val ...]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; TAG="$1"; S="$2"; shift; shift
for kv in "$@"; do export "$kv"; done
O="$R/gpurun_out/$TAG"; rm -rf "$O"; mkdir -p "$O"
if [ "$S" = "one" ]; then W="--workload C2 --batch 1"; else W="--shard-of $S --shard-rank 0"; fi
# shellcheck disable=SC2086
rocprofv3 --kernel-trace --output-format csv -d "$O/tr" -o t -- python3 "$R/bench.py" --in-flight 1 $W --steps 3 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$O/bench.json" 2> "$O/bench.err" || true
python3 "$R/tools/trace_timeline.py" "$(find "$O/tr" -name '*kernel_trace.csv' | head -1)" > "$O/timeline.txt" 2>&1 || true
rm -rf "$O/tr"
tail -25 "$O/timeline.txt"
