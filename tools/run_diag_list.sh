#!/bin/bash
# per-launch times of the ablation builds (tools/ablate.sh) for kernels matching a substring, on ONE box.
# usage: tools/run_diag_list.sh <substring> <diag> ...   (0 = product)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; W="$1"; shift
for D in "$@"; do
  if [ "$D" != "0" ]; then export VSP_LIB_PATH="$R/build/diag$D/libvispeech_hip.so"; else unset VSP_LIB_PATH; fi
  O="$R/gpurun_out/dl$D"; rm -rf "$O"; mkdir -p "$O"
  rocprofv3 --kernel-trace --output-format csv -d "$O" -o t -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" || true
  echo "== diag $D"
  python3 "$R/tools/trace_list.py" "$(find "$O" -name '*kernel_trace.csv' | head -1)" "$W" | tee "$O/list.txt"
  rm -f "$O"/*kernel_trace.csv "$O"/*agent_info.csv
done
