#!/bin/bash
# Per-rank operating points of the strong-scaling metric on ONE GPU: rank 0's shard_range slice of the C3 batch at
# N = 1 / 2 / 4 / 8 under the global frame padding (bench.py --shard-of N), bench line + per-launch generator table each.
# usage: tools/operating_points.sh <tag> [steps]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; TAG="${1:-r05_op}"; STEPS="${2:-20}"
O="$R/gpurun_out/$TAG"; rm -rf "$O"; mkdir -p "$O"
for N in 1 2 4 8; do
  python3 "$R/bench.py" --shard-of $N --shard-rank 0 --steps "$STEPS" --warmup 4 --no-cpu-baseline > "$O/bench_n$N.json" 2> "$O/bench_n$N.err" || true
  tail -1 "$O/bench_n$N.json" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('N=$N B=%d  %.2f ms/step  %.1f M samples/s | generator %.2f ms (%d launches)  attention %.3f ms  frame %.2f ms (%d launches)' % (
  d['config']['utterances_per_gpu'], d['ms_per_step'], d['value']/1e6, r['kernel_ms_per_step'], r['launches'],
  r['attention']['ms_per_step'], r['frame_rate_convs']['ms_per_step'], r['frame_rate_convs']['launches']))"
done 2>&1 | tee "$O/summary.txt"
for N in 4 8; do
  B=$((64 / N))
  rocprofv3 --kernel-trace --output-format csv -d "$O/trace_n$N" -o t -- python3 "$R/bench.py" --in-flight 1 --shard-of $N --shard-rank 0 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$O/trace_n$N.err" || true
  python3 "$R/tools/trace_fused.py" "$(find "$O/trace_n$N" -name '*kernel_trace.csv' | head -1)" $B 489 > "$O/per_launch_n$N.txt" 2>&1 || true
  python3 "$R/tools/trace_timeline.py" "$(find "$O/trace_n$N" -name '*kernel_trace.csv' | head -1)" > "$O/timeline_n$N.txt" 2>&1 || true
  rm -rf "$O/trace_n$N"
  tail -2 "$O/per_launch_n$N.txt"
done
