#!/bin/bash
# round 4: quick parity subset + same-box A/B of the in-tree library against build/head (tools/build_rev.sh HEAD head)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; cd "$R"
timeout 900 python -m pytest tests/test_cl_ops.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 1200 bash tools/run_ab.sh ${1:-3} head product
