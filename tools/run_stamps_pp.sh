#!/bin/bash
# usage: tools/run_stamps_pp.sh <K> <dil> ...   (pairs of arguments; see tools/stamps_pp.py; the library is build/ppstamps/)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export VSP_LIB_PATH="$GRAFT_REPO_ROOT/build/ppstamps/libvispeech_hip.so"
while [ $# -ge 2 ]; do python tools/stamps_pp.py "$1" "$2" 2>&1 | grep -v amdgpu.ids; shift 2; done
