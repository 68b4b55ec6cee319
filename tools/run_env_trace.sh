#!/bin/bash
# per-launch generator times (tools/trace_list.py) for several environment settings on ONE box.
# usage: tools/run_env_trace.sh <substring> "<VAR=V ...>" ...   ("-" = no extra variables)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; W="$1"; shift
i=0
for E in "$@"; do
  if [ "$E" = "-" ]; then E=""; fi
  O="$R/gpurun_out/et$i"; rm -rf "$O"; mkdir -p "$O"
  # (variables are exported into a subshell: rocprofv3 must be followed by the program itself)
  ( for kv in $E; do export "$kv"; done
    rocprofv3 --kernel-trace --output-format csv -d "$O" -o t -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" ) || true
  echo "== [$E]"
  python3 "$R/tools/trace_list.py" "$(find "$O" -name '*kernel_trace.csv' | head -1)" "$W" | tee "$O/list.txt"
  rm -f "$O"/*kernel_trace.csv "$O"/*agent_info.csv
  i=$((i+1))
done
