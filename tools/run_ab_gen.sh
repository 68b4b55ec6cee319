#!/bin/bash
# interleaved same-box A/B of library builds: C3 step + generator ms, one-utterance latency, C5.
# usage (through gpurun): tools/run_ab_gen.sh <rounds> <build dir | product> ...
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
N="$1"; shift
for r in $(seq 1 "$N"); do
  for D in "$@"; do
    if [ "$D" != "product" ]; then export VSP_LIB_PATH="$GRAFT_REPO_ROOT/build/$D/libvispeech_hip.so"; else unset VSP_LIB_PATH; fi
    python - "$D" <<'PY'
import json, subprocess, sys
def run(*a):
    o = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", *a], capture_output=True, text=True).stdout.strip().splitlines()
    return json.loads(o[-1])
c3 = run("--steps", "12", "--warmup", "3")
one = run("--workload", "C2", "--batch", "1", "--steps", "30", "--warmup", "5")
c5 = run("--workload", "C5", "--steps", "8", "--warmup", "2")
print(f"{sys.argv[1]:10s} C3 {c3['ms_per_step']:7.2f} ms (generator {c3['roofline']['kernel_ms_per_step']:6.2f})   one utterance {one['ms_per_step']:6.3f} ms   C5 {c5['ms_per_step']:6.2f} ms")
PY
  done
done
