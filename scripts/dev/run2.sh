set -u
mkdir -p gpurun_out/r3a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python scripts/dev/overlap_probe.py 2>&1 | grep streams > gpurun_out/r3a/overlap2.txt
cat gpurun_out/r3a/overlap2.txt
