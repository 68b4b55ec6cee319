#!/bin/bash
# usage: tools/run_stamps_rw.sh <K> <dil> ...   (pairs of arguments; see tools/stamps_rw.py; the library is build/rwstamps/)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export VSP_LIB_PATH="$GRAFT_REPO_ROOT/build/rwstamps/libvispeech_hip.so"
while [ $# -ge 2 ]; do python tools/stamps_rw.py "$1" "$2" 2>&1 | grep -v amdgpu.ids; shift 2; done
