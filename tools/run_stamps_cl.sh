#!/bin/bash
cd $GRAFT_REPO_ROOT
export VSP_LIB_PATH=$GRAFT_REPO_ROOT/build/stamps/libvispeech_hip.so
for L in "$@"; do VSP_STAMP_CL=$L python tools/stamps.py 2>&1 | grep -v amdgpu.ids; done
