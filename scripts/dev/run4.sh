set -u
mkdir -p gpurun_out/r3c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 1 1024 4096; do timeout 300 python scripts/dev/ipm4_prof.py $B 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3c/ipm4_prof.txt
