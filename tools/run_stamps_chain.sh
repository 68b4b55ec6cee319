#!/bin/bash
# usage: tools/run_stamps_chain.sh <launch index> ...   (see tools/stamps_chain.py)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export VSP_LIB_PATH="$GRAFT_REPO_ROOT/build/g16stamps/libvispeech_hip.so"
for L in "$@"; do VSP_STAMP_CHAIN="$L" python tools/stamps_chain.py 2>&1 | grep -v amdgpu.ids; done
