#!/bin/bash
# full GPU test suite + one bench line (checkpoint of a round-4 milestone); usage: r04_checkpoint.sh <tag>
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04_ckpt_${1:-x}"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee "$O/pytest_gpu.txt"
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2> "$O/bench.err" | tee "$O/bench.json" | cut -c1-400
