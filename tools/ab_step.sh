#!/bin/bash
# interleaved A/B of library builds on ONE box by STEP time (bench.py, 20 timed steps, no profiler): the generator is
# power-limited, so a saving in one kernel can be partly given back by the kernels after it -- per-kernel traces (run_ab.sh)
# attribute, this decides.  usage: tools/ab_step.sh <rounds> <dir-under-build|product> ...
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; N="$1"; shift
for i in $(seq 1 "$N"); do
  for D in "$@"; do
    if [ "$D" != "product" ]; then export VSP_LIB_PATH="$R/build/$D/libvispeech_hip.so"; else unset VSP_LIB_PATH; fi
    python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('== %s #%s: %.3f ms/step  generator %.3f ms' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['kernel_ms_per_step']))" "$D" "$i"
  done
done
