#!/bin/bash
# kernel trace of one bench run on the GPU box -> gpurun_out/<tag>/ ; usage: tools/run_trace.sh <tag> [env=val ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf $R/gpurun_out/$TAG; mkdir -p $R/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
tail -1 $R/gpurun_out/$TAG/bench.json | cut -c1-300
python3 $R/tools/trace_fused.py $(find $R/gpurun_out/$TAG -name '*kernel_trace.csv' | head -1)
