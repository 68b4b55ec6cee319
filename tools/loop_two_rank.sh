#!/bin/bash
# Reproduce the intermittent two-rank timeout: N runs of `python bench.py --gpus 2` on ONE GPU over
# gloo, each under its own timeout; a hung rank dumps its Python stacks (VSP_BENCH_DUMP_AFTER) and exits non-zero.
# usage: tools/loop_two_rank.sh [runs] [extra bench args...]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; N="${1:-30}"; shift || true
O="$R/gpurun_out/r05_two_rank_loop"; rm -rf "$O"; mkdir -p "$O"
export VSP_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 VSP_BENCH_PG_TIMEOUT=60 VSP_BENCH_DUMP_AFTER=75
cd "$R"
ok=0; bad=0
for i in $(seq 1 "$N"); do
  t0=$(date +%s)
  timeout 120 python3 bench.py --gpus 2 --batch 6 --steps 2 --warmup 1 "$@" > "$O/run_$i.out" 2> "$O/run_$i.err"
  rc=$?
  t1=$(date +%s)
  if [ $rc -eq 0 ] && grep -q '^{' "$O/run_$i.out"; then ok=$((ok+1)); rm -f "$O/run_$i.err" "$O/run_$i.out"; else bad=$((bad+1)); fi
  printf "run %2d rc=%d %ds\n" "$i" "$rc" "$((t1 - t0))"
done | tee "$O/summary.txt"
echo "ok=$(grep -c 'rc=0' "$O/summary.txt") of $N" | tee -a "$O/summary.txt"
