set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final2/stats_rep -o s -- python bench.py --streams 1 --same-batch --no-cpu-baseline --no-schedule-legs --steps 20 --warmup 3 > gpurun_out/final2/rep.log 2>&1 < /dev/null
grep metric gpurun_out/final2/rep.log | cut -c1-400; head -6 gpurun_out/final2/stats_rep/s_kernel_stats.csv | cut -c1-150
TUM_NMPC_SCHEDULE=natural timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-schedule-legs 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('natural order, 3 streams: value %.3f M'%(d['value']/1e6))"
