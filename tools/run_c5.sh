#!/bin/bash
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/c5"; rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
head -14 $O/t_kernel_stats.csv | cut -c1-160
