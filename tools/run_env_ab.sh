#!/bin/bash
# interleaved A/B of environment settings on ONE box with the in-tree library: step and generator time per setting.
# usage: tools/run_env_ab.sh <rounds> "<VAR=V VAR2=V2>" "<...>" ...   ("-" = no extra variables)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; N="$1"; shift
for i in $(seq 1 "$N"); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then E=""; fi
    # shellcheck disable=SC2086
    env $E python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('== [%s] #%s: %.2f ms/step  generator %.2f ms in %d launches  frame-rate convs %.2f ms  attention %.2f ms' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['kernel_ms_per_step'], r['launches'], r['frame_rate_convs']['ms_per_step'], r['attention']['ms_per_step']))" "$E" "$i"
  done
done
