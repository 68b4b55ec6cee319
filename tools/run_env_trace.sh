#!/bin/bash
# per-launch generator table (tools/trace_fused.py) of one bench run under each environment setting, same box.
# usage: tools/run_env_trace2.sh "<VAR=V ..>" ...   ("-" = no extra variables)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; n=0
for E in "$@"; do
  n=$((n + 1)); O="$R/gpurun_out/envtrace_$n"; rm -rf "$O"; mkdir -p "$O"
  if [ "$E" = "-" ]; then E=""; fi
  ( for kv in $E; do export "$kv"; done
    rocprofv3 --kernel-trace --output-format csv -d "$O" -o t -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" )
  python3 "$R/tools/trace_fused.py" "$(find "$O" -name '*kernel_trace.csv' | head -1)" > "$O/per_launch.txt"
  echo "== [$E]"; cat "$O/per_launch.txt"
  rm -f "$O"/*kernel_trace.csv "$O"/*agent_info.csv
done
