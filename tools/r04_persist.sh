#!/bin/bash
# round 4: persistent g16_conv blocks (VSP_G16_PERSIST): conv parity, then same-box A/B
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"; cd "$R"
timeout 900 python -m pytest tests/test_cl_ops.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 bash tools/run_env_ab.sh 3 "VSP_G16_PERSIST=0" "-"
