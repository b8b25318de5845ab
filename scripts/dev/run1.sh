set -u
mkdir -p gpurun_out/r3a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3a/pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | grep metric > gpurun_out/r3a/bench.json
timeout 300 python scripts/dev/overlap_probe.py 2>&1 | grep streams > gpurun_out/r3a/overlap.txt
cat gpurun_out/r3a/pytest.log gpurun_out/r3a/overlap.txt; cut -c1-1500 gpurun_out/r3a/bench.json
