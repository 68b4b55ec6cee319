#!/bin/bash
# same-box A/B of the generator's ResBlock chains on side streams (VSP_RB_STREAMS=<stage mask>) against one stream (=0), by
# STEP time, at the C3 batch and at rank 0's slice of it at N = 2 / 4 / 8 (bench.py --shard-of).
# usage: tools/ab_streams.sh <rounds> "<masks>" "<shard-ofs>"
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; N="${1:-2}"
for i in $(seq 1 "$N"); do
  for S in ${3:-1 2 4 8}; do
    for V in ${2:-0 15}; do
      VSP_RB_STREAMS=$V python3 "$R/bench.py" --shard-of $S --shard-rank 0 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== shard-of %s streams=%s #%s: %.3f ms/step  %.1f M samples/s' % (sys.argv[1], sys.argv[2], sys.argv[3], d['ms_per_step'], d['value'] / 1e6))" "$S" "$V" "$i"
    done
  done
done
