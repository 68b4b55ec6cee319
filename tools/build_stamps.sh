#!/bin/bash
# Build the in-kernel-stamp variant of libvispeech_hip.so (-DG16_STAMPS, see gen16.hip) into build/g16stamps/.
set -e
cd "$(dirname "$0")/../vispeech_amd/csrc"
mkdir -p ../../build/g16stamps
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DG16_STAMPS -shared conv_mfma.hip cl_misc.hip gen16.hip attention.hip attention_f16s.hip misc.hip api.hip -x hip weights.cpp -o ../../build/g16stamps/libvispeech_hip.so 2>&1 | grep -E "error" || true
# ... and of the frame-rate latency kernels (-DFR_STAMPS, see conv_mfma.hip; tools/stamps_frame.py) into build/frstamps/.
mkdir -p ../../build/frstamps
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DFR_STAMPS -shared conv_mfma.hip cl_misc.hip gen16.hip attention.hip attention_f16s.hip misc.hip api.hip -x hip weights.cpp -o ../../build/frstamps/libvispeech_hip.so 2>&1 | grep -E "error" || true
