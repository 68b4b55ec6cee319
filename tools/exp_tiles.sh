#!/bin/bash
# A/B of the VSP_TILE_EXP tile variants on the GPU box: parity of the generator stage, then per-stage
# kernel time of one bench step (rocprofv3 kernel trace).  usage: tools/exp_tiles.sh 0 1 2 4 8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/exp
for E in "$@"; do
  export VSP_TILE_EXP=$E
  echo "=== VSP_TILE_EXP=$E"
  true
  rm -rf $R/gpurun_out/exp/t$E
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/exp/t$E -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/exp/bench$E.json 2>$R/gpurun_out/exp/bench$E.err
  python3 -c "import json;d=json.loads(open('$R/gpurun_out/exp/bench$E.json').read().strip().splitlines()[-1]);print('ms_per_step',round(d['ms_per_step'],2))"
  python3 $R/tools/trace_cl.py $(find $R/gpurun_out/exp/t$E -name '*kernel_trace.csv' | head -1) --brief
done
