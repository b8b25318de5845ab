set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
SN_BATCH=4096 timeout 600 python scripts/snmpc_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e/snmpc_bench.txt
