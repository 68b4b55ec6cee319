set -u
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for i in 1 2; do
  for V in 0 1; do
    O="$R/gpurun_out/ab_rw64_${V}_$i"; rm -rf "$O"; mkdir -p "$O"
    export VSP_RW64=$V
    rocprofv3 --kernel-trace --output-format csv -d "$O" -o t -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err" || true
    python3 "$R/tools/trace_fused.py" "$(find "$O" -name '*kernel_trace.csv' | head -1)" > "$O/per_launch.txt"
    echo "== VSP_RW64=$V #$i: $(grep s2k3 "$O/per_launch.txt" | awk '{printf "%s ", $9}') | $(tail -2 "$O/per_launch.txt" | tr '\n' ' ')"
    rm -f "$O"/*kernel_trace.csv
  done
done
