# A/B of experimental builds of the library on config 2: TUM_NMPC_LIB=<lib> scripts/pipe_check.py pipeline (device time per solve, ipm_kernel time)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for lib in "$@"; do
  echo "== $lib"
  if [ "$lib" = "shipped" ]; then timeout 300 python scripts/pipe_check.py pipeline 2>&1 | grep -E "^pipeline|warm"; else TUM_NMPC_LIB=$PWD/$lib timeout 300 python scripts/pipe_check.py pipeline 2>&1 | grep -E "^pipeline|warm"; fi
done | tee gpurun_out/ab/ab.txt
