#!/bin/bash
# Builds the library of the WORKING TREE with extra flags (experiment macros) for A/B runs against the shipped one:
#   scripts/dev/build_tree_lib.sh <tag> [extra hipcc flags ...]   ->   exp_libs/lib_tree_<tag>.so
# e.g.   scripts/dev/build_tree_lib.sh o2 -O2      (the experiment macros of rounds 3-5 are gone: round 6 removed the switches that lost)
set -eu
TAG=$1; shift
ROOT=$(git rev-parse --show-toplevel)
mkdir -p "$ROOT/exp_libs"
OUT="$ROOT/exp_libs/lib_tree_${TAG}.so"
(cd "$ROOT/tum-control_amd/csrc" && ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC \
    -mllvm -amdgpu-mfma-vgpr-form=1 -Rpass-analysis=kernel-resource-usage "$@" -o "$OUT" tum_nmpc.hip 2> "$OUT.remarks" || { tail -20 "$OUT.remarks"; exit 1; })
grep -A12 "Function Name: _ZN3tum10ipm_kernelILb0ELi5" "$OUT.remarks" | grep -E "VGPRs:|AGPRs|Spill|Scratch" | sed 's/.*remark: //' | tr '\n' ' '; echo
echo "$OUT"
