#!/bin/bash
# Timing-only ablation builds of the frame-rate convolution kernel (CONV_DIAG bits, conv_mfma.hip) into build/cdiag<N>/.
set -e
cd "$(dirname "$0")/../vispeech_amd/csrc"
for D in "$@"; do
  mkdir -p ../../build/cdiag$D
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCONV_DIAG=$D -shared conv_mfma.hip cl_misc.hip gen16.hip attention.hip attention_f16s.hip misc.hip api.hip -x hip weights.cpp -o ../../build/cdiag$D/libvispeech_hip.so 2>&1 | grep -E "error" || true
done
