#!/bin/bash
# Build timing-only ablation variants of libvispeech_hip.so (G16_DIAG bits, see gen16.hip) into build/diag<N>/.
# Results of these builds are WRONG by construction; they only price the kernel's components.
set -e
cd "$(dirname "$0")/../vispeech_amd/csrc"
for D in "$@"; do
  mkdir -p ../../build/diag$D
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DG16_DIAG=$D -shared conv_mfma.hip cl_misc.hip gen16.hip attention.hip attention_f16s.hip misc.hip api.hip -x hip weights.cpp -o ../../build/diag$D/libvispeech_hip.so 2>&1 | grep -E "error" || true
done
