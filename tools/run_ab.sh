#!/bin/bash
# interleaved A/B of library builds on ONE box: kernel trace per build, per-stage generator times.
# usage: tools/run_ab.sh <rounds> <dir-under-build> ...   ("product" = the in-tree library)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; N="$1"; shift
for i in $(seq 1 "$N"); do
  for D in "$@"; do
    if [ "$D" != "product" ]; then export VSP_LIB_PATH="$R/build/$D/libvispeech_hip.so"; else unset VSP_LIB_PATH; fi
    O="$R/gpurun_out/ab_${D}_$i"; rm -rf "$O"; mkdir -p "$O"
    rocprofv3 --kernel-trace --output-format csv -d "$O" -o t -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" || true
    python3 "$R/tools/trace_fused.py" "$(find "$O" -name '*kernel_trace.csv' | head -1)" > "$O/per_launch.txt"
    echo "== $D #$i: $(tail -2 "$O/per_launch.txt" | tr '\n' ' ')"
    rm -f "$O"/*kernel_trace.csv "$O"/*agent_info.csv
  done
done
