#!/bin/bash
# round 4 iteration loop for g16_rw: unit parity (bit-identity), per-launch times at the C3 size under VSP_RW_DIAG settings,
# in-kernel stamps.  usage: r04_rw_iter.sh "<diag ..>" [<K> <dil> ...]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R="$GRAFT_REPO_ROOT"
cd "$R"
timeout 600 python -m pytest tests/test_cl_ops.py -x -q -m gpu -k "resblock" 2>&1 | tail -4
rm -f "$R/gpurun_out/r04_rw_diag/table.txt"
# shellcheck disable=SC2086
timeout 900 bash tools/r04_rw_diag.sh $1 > /dev/null 2>&1
cat "$R/gpurun_out/r04_rw_diag/table.txt"
shift
[ $# -ge 2 ] && timeout 300 bash tools/run_stamps_rw.sh "$@"
