set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do echo "TUM_NMPC_COND=$v"; TUM_NMPC_COND=$v timeout 300 python scripts/pipe_check.py pipeline 2>&1 | grep -E "^pipeline|warm"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3stats -o s -- python bench.py --streams 1 --no-cpu-baseline --no-schedule-legs --steps 10 --warmup 2 > /dev/null 2>&1 < /dev/null; head -6 gpurun_out/c3stats/s_kernel_stats.csv | cut -c1-140
for v in 1 0; do TUM_NMPC_COND=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-schedule-legs 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cond variant $v: 3 streams value %.3f M'%(d['value']/1e6))"; done
