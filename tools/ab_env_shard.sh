#!/bin/bash
# same-box A/B of environment settings by STEP time at rank 0's slice of the C3 batch at N ranks (bench.py --shard-of) and
# for one utterance.  usage: tools/ab_env_shard.sh <rounds> "<shard-ofs>" "<VAR=V ...>" "<...>"   ("-" = no extra variables)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; N="$1"; SH="$2"; shift; shift
for i in $(seq 1 "$N"); do
  for S in $SH; do
    for E in "$@"; do
      EE="$E"; if [ "$E" = "-" ]; then EE=""; fi
      if [ "$S" = "one" ]; then W="--workload C2 --batch 1 --steps 100 --warmup 10"; else W="--shard-of $S --shard-rank 0 --steps 20 --warmup 5"; fi
      # shellcheck disable=SC2086
      env $EE python3 "$R/bench.py" $W --no-cpu-baseline --profile-steps 0 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== %s [%s] #%s: %.3f ms/step  %.1f M samples/s' % (sys.argv[1], sys.argv[2], sys.argv[3], d['ms_per_step'], d['value'] / 1e6))" "$S" "$E" "$i"
    done
  done
done
