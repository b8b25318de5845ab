# A/B of builds of the library through the whole bench line (fresh batches, --streams S): usage ab_bench.sh S lib1 lib2 ... ("shipped")
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "shipped" ]; then unset TUM_NMPC_LIB; else export TUM_NMPC_LIB=$PWD/$lib; fi
  timeout 300 python bench.py --steps 30 --warmup 3 --streams $S --no-cpu-baseline --no-schedule-legs 2>/dev/null < /dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib streams',d['config']['streams'],'value %.3f M solves/s'%(d['value']/1e6),'ms/step %.3f'%d['ms_per_step'])"
done
done
