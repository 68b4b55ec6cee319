#!/bin/bash
# round 4: operand-image hand-over of ResBlock intermediates (VSP_TIMG): parity subset, then same-box A/B against fp32 hand-over
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; cd "$R"
timeout 900 python -m pytest tests/test_cl_ops.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4
timeout 900 bash tools/run_env_ab.sh 2 "VSP_TIMG=0" "-"
