#!/bin/bash
# one rocprofv3 --pmc pass per counter group over a short bench run; usage: tools/run_pmc.sh <tag> "<counters>" ["<counters>" ...]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; TAG="${1:?tag}"; shift
i=0
for G in "$@"; do
  O="$R/gpurun_out/$TAG/g$i"; rm -rf "$O"; mkdir -p "$O"
  rocprofv3 --pmc $G --output-format csv -d $O -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/err.txt
  i=$((i+1))
done
python3 $R/tools/pmc_table.py "${PMC_KERNEL:-g16_conv<4, 4}" $(ls $R/gpurun_out/$TAG/g*/p_counter_collection.csv) | tail -${PMC_TAIL:-12}
