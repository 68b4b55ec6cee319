#!/bin/bash
# round 4, first GPU run of the register-weights pair kernel: unit parity, same-box A/B against the ring kernel, trace.
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04_rw_first"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_cl_ops.py -x -q -m gpu -k "resblock" 2>&1 | tail -15 > "$O/pytest_resblock.txt"
cat "$O/pytest_resblock.txt"
timeout 900 bash tools/run_env_ab.sh 2 "VSP_PAIR=ring" "-" 2>&1 | tee "$O/ab.txt"
timeout 600 bash tools/run_trace.sh r04_rw_first/trace 2>&1 | tail -5
grep "s3k" "$O/trace/per_launch.txt"
