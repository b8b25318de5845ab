#!/bin/bash
# Builds the library of an EARLIER COMMIT for A/B runs against the current tree (scripts/dev/ab2.py, scripts/dev/ab.sh):
#   scripts/dev/build_exp_lib.sh <git-rev> [extra hipcc flags ...]   ->   exp_libs/lib_<short rev>[_<tag>].so
# e.g.   scripts/dev/build_exp_lib.sh 3bc4d34                 (the kernels before round 3's instruction-level work on ipm_kernel)
#        TAG=wps2 scripts/dev/build_exp_lib.sh HEAD -DIPM_WPS=2  (two wavefronts per SIMD in the interior point kernel)
# exp_libs/ is git-ignored (built artefacts) but travels to the GPU box with gpurun; every A/B file under profiles/ names the
# revision (and flags) its libraries were built from, so this script reproduces them.
set -eu
REV=$1; shift
SHORT=$(git rev-parse --short "$REV")
ROOT=$(git rev-parse --show-toplevel)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive "$REV" tum-control_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/exp_libs"
OUT="$ROOT/exp_libs/lib_${SHORT}${TAG:+_$TAG}.so"
(cd "$TMP/tum-control_amd/csrc" && ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC \
    -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -o "$OUT" tum_nmpc.hip)
echo "$OUT"
