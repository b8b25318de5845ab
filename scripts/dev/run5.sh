set -u
mkdir -p gpurun_out/r3d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
./scripts/probes/probe_mfma_4x4x4 > gpurun_out/r3d/probe_4x4x4.txt 2>&1
head -20 gpurun_out/r3d/probe_4x4x4.txt; tail -2 gpurun_out/r3d/probe_4x4x4.txt
timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
for S in 1 2 3 4; do timeout 300 python bench.py --steps 20 --warmup 3 --streams $S --no-cpu-baseline --no-schedule-legs 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams',d['config']['streams'],'value %.3f M'%(d['value']/1e6),'ms/step %.3f'%d['ms_per_step'])"; done | tee gpurun_out/r3d/streams.txt
for C in 3 4 5; do for S in 1 3; do timeout 300 python bench.py --config $C --steps 10 --warmup 2 --streams $S --no-cpu-baseline --no-schedule-legs 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config',$C,'streams',d['config']['streams'],'value %.3f M'%(d['value']/1e6),'ms/step %.3f'%d['ms_per_step'],'ok',d['status_ok_frac'])"; done; done | tee -a gpurun_out/r3d/streams.txt
