#!/bin/bash
# time the generator kernels of each gen16 ablation build (tools/ablate.sh) on the GPU box; usage: tools/run_g16_ablate.sh 0 1 2 ...
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for D in "$@"; do
  if [ "$D" != "0" ]; then export VSP_LIB_PATH="$R/build/diag$D/libvispeech_hip.so"; else unset VSP_LIB_PATH; fi
  O="$R/gpurun_out/g16abl$D"; rm -rf "$O"; mkdir -p "$O"
  rocprofv3 --kernel-trace --output-format csv -d "$O" -o t -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" || true
  python3 "$R/tools/trace_fused.py" "$(find "$O" -name '*kernel_trace.csv' | head -1)" > "$O/per_launch.txt"
  echo "== diag $D: $(tail -2 "$O/per_launch.txt" | tr '\n' ' ')"
done
