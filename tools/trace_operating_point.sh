#!/bin/bash
# Kernel trace of one rank's slice of the C3 batch at N ranks (bench.py --shard-of N): generator per-launch table and the
# launch-by-launch timeline of everything else -> gpurun_out/op_trace/.   usage: tools/trace_operating_point.sh "8 4"
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/op_trace"; rm -rf "$O"; mkdir -p "$O"
for N in ${1:-8}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t$N" -o t -- python3 "$R/bench.py" --in-flight 1 --shard-of $N --shard-rank 0 --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline > "$O/bench_$N.json" 2> "$O/err_$N.txt" || true
  python3 "$R/tools/trace_fused.py" "$O/t$N/t_kernel_trace.csv" > "$O/generator_per_launch_$N.txt" || true
  python3 "$R/tools/trace_timeline.py" "$O/t$N/t_kernel_trace.csv" > "$O/timeline_$N.txt" || true
  cp "$O/t$N/t_kernel_stats.csv" "$O/kernel_stats_$N.csv" || true
  rm -rf "$O/t$N"
done
tail -3 "$O"/generator_per_launch_*.txt
