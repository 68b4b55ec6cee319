#!/bin/bash
# Build libvispeech_hip.so of a git revision into build/<dir>/ (for same-box A/B runs with tools/run_ab.sh).
# usage: tools/build_rev.sh <git-rev> <dir-under-build>
set -eu
REV="${1:?git revision}"; D="${2:?directory under build/}"
R="$(cd "$(dirname "$0")/.." && pwd)"
T="$(mktemp -d)"
trap 'rm -rf "$T"' EXIT
git -C "$R" archive "$REV" vispeech_amd/csrc include | tar -x -C "$T"
mkdir -p "$R/build/$D"
cd "$T/vispeech_amd/csrc"
SRCS="$(sed -n 's/^SRCS *[:+]*= *//p' Makefile)"
# shellcheck disable=SC2086
EXTRA="$(grep -q -- '-packed-fp32-ops' Makefile && echo '-Xclang -target-feature -Xclang -packed-fp32-ops' || true)"   # (the revision's own code-generation flags)
# shellcheck disable=SC2086
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $EXTRA -shared $(for f in $SRCS; do case "$f" in *.cpp) echo "-x hip $f";; *) echo "$f";; esac; done) -o "$R/build/$D/libvispeech_hip.so" 2>&1 | grep -E "error" || true
ls -la "$R/build/$D/libvispeech_hip.so"
