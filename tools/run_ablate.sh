#!/bin/bash
# time the generator kernels of each ablation build (tools/ablate.sh) on the GPU box; usage: tools/run_ablate.sh 0 1 3 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for D in "$@"; do
  if [ "$D" != "0" ]; then export VSP_LIB_PATH=$R/build/diag$D/libvispeech_hip.so; else unset VSP_LIB_PATH; fi
  rm -rf $R/gpurun_out/abl$D; mkdir -p $R/gpurun_out/abl$D
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl$D -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/abl$D/bench.json 2> $R/gpurun_out/abl$D/bench.err
  echo "== diag $D: $(python3 $R/tools/trace_fused.py $(find $R/gpurun_out/abl$D -name '*kernel_trace.csv' | head -1) | tail -2 | tr '\n' ' ')"
done
