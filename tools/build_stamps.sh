#!/bin/bash
# Diagnostic builds with in-kernel wall-clock stamps (s_memrealtime), next to the product library:
#   build/g16stamps/  -DG16_STAMPS  (gen16.hip: g16_pair phase stamps, tools/stamps_pair.py)
#   build/rwstamps/   -DRW_STAMPS   (gen16_rw.hip: per-wave phase timeline of the persistent pair kernel, tools/stamps_rw.py)
#   build/ppstamps/   -DPP_STAMPS   (gen16_pp.hip: phase timeline of the 128-channel pair kernel, tools/stamps_pp.py)
set -e
cd "$(dirname "$0")/../vispeech_amd/csrc"
SRCS="conv_mfma.hip cl_misc.hip gen16.hip gen16_rw.hip gen16_pipe.hip gen16_pp.hip gen16_rc.hip attention.hip attention_f16s.hip misc.hip api.hip -x hip weights.cpp"
for V in "g16stamps G16_STAMPS" "rwstamps RW_STAMPS" "ppstamps PP_STAMPS"; do
  set -- $V
  if [ -n "${ONLY:-}" ] && [ "$ONLY" != "$1" ]; then continue; fi
  mkdir -p ../../build/$1
  # shellcheck disable=SC2086
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -D$2 -shared $SRCS -o ../../build/$1/libvispeech_hip.so 2>&1 | grep -E "error" || true
  ls -la ../../build/$1/libvispeech_hip.so
done
