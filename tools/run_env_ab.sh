#!/bin/bash
# bench A/B over values of one environment variable: tools/run_env_ab.sh VAR v1 v2 ... (each run twice, interleaved)
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for rep in 1 2; do for V in "$@"; do
  env $VAR=$V python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$V', round(d['ms_per_step'],2),'ms', round(d['value']/1e6,1),'M samples/s')"
done; done
