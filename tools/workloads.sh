#!/bin/bash
# round 4: the other workloads' step times (one utterance, C5, C2) for the in-tree library and optionally build/<dir>
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; cd "$R"
for D in "$@"; do
  if [ "$D" != "product" ]; then export VSP_LIB_PATH="$R/build/$D/libvispeech_hip.so"; else unset VSP_LIB_PATH; fi
  python - "$D" <<'PY'
import json, subprocess, sys
def run(*a):
    o = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", *a], capture_output=True, text=True).stdout.strip().splitlines()
    return json.loads(o[-1])
one = run("--workload", "C2", "--batch", "1", "--steps", "40", "--warmup", "8")
c5 = run("--workload", "C5", "--steps", "10", "--warmup", "3")
c2 = run("--workload", "C2", "--steps", "10", "--warmup", "3")
f = lambda d: f"{d['ms_per_step']:7.3f} ms (gen {d['roofline']['kernel_ms_per_step']:6.3f}, frame {d['roofline']['frame_rate_convs']['ms_per_step']:5.3f}, att {d['roofline']['attention']['ms_per_step']:5.3f} @ {100*d['roofline']['attention']['mfma_utilisation']:4.1f}%)"
print(f"{sys.argv[1]:8s} one utterance {f(one)} | C5 {f(c5)} | C2 {f(c2)}")
PY
done
