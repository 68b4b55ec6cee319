#!/bin/bash
# Reproduce the intermittent two-rank timeout: N runs of `python bench.py --gpus 2` on ONE GPU over
# gloo, each under its own timeout; a hung rank dumps its Python stacks (VSP_BENCH_DUMP_AFTER) and exits non-zero.
# A run is "ok" when it exits 0 AND printed its JSON line: the verdict is on each run's line and the total counts THOSE
# (counters updated inside a `| tee` pipeline are lost in its subshell).
# usage: tools/loop_two_rank.sh [runs] [tag] [extra bench args...]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; N="${1:-30}"; TAG="${2:-two_rank_loop}"; shift || true; shift || true
O="$R/gpurun_out/$TAG"; rm -rf "$O"; mkdir -p "$O"
export VSP_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 VSP_BENCH_PG_TIMEOUT=60 VSP_BENCH_DUMP_AFTER=75
cd "$R"
: > "$O/summary.txt"
for i in $(seq 1 "$N"); do
  t0=$(date +%s%N)
  timeout 120 python3 bench.py --gpus 2 --batch 6 --steps 2 --warmup 1 "$@" > "$O/run_$i.out" 2> "$O/run_$i.err"
  rc=$?
  t1=$(date +%s%N)
  if [ $rc -eq 0 ] && grep -q '^{' "$O/run_$i.out"; then v=ok; rm -f "$O/run_$i.err" "$O/run_$i.out"; else v=BAD; fi
  printf "run %2d %s rc=%d %d.%ds\n" "$i" "$v" "$rc" "$(( (t1 - t0) / 1000000000 ))" "$(( (t1 - t0) / 100000000 % 10 ))" | tee -a "$O/summary.txt"
done
echo "ok=$(grep -c ' ok rc=0' "$O/summary.txt") bad=$(grep -c ' BAD ' "$O/summary.txt") of $N" | tee -a "$O/summary.txt"
