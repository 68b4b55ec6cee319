#!/bin/bash
# Frame-rate kernel selection at the per-rank operating points (round 5): the experiments build's
# VSP_FR_SPLITK (64 x 32 tiles up to which the channel-split latency kernel runs; product 512) and VSP_FR_BLOCKS
# (64 x 128 tiles up to which conv_frame_f16s runs; product 256) swept at N = 2 / 4 / 8 (rank 0's slice, global padding).
# usage: tools/sweep_frame_thresholds.sh   (needs build/exp/libvispeech_hip.so: -DVSP_EXPERIMENTS)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; export VSP_LIB_PATH="$R/build/exp/libvispeech_hip.so"
O="$R/gpurun_out/r05_frame_sweep"; mkdir -p "$O"
for N in 8 4 2; do
  for CFG in "512 256" "1024 256" "2048 256" "4096 256" "512 512" "512 1024" "1024 512" "2048 1024" "4096 2048" "8192 4096"; do
    set -- $CFG
    VSP_FR_SPLITK=$1 VSP_FR_BLOCKS=$2 python3 "$R/bench.py" --shard-of $N --steps 20 --warmup 4 --no-cpu-baseline 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('N=$N splitk<=$1 frame<=$2: %.3f ms/step | generator %.2f  attention %.3f  frame %.3f ms' % (d['ms_per_step'], r['kernel_ms_per_step'], r['attention']['ms_per_step'], r['frame_rate_convs']['ms_per_step']))"
  done
done 2>&1 | tee "$O/sweep.txt"
