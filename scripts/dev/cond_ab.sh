cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  T=$(basename $L .so)
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/condab/$T -o s -- python scripts/dev/one_lib_solves.py $L > /dev/null 2>&1 < /dev/null
  echo "$T: $(grep -h 'cond_kernel' gpurun_out/condab/$T/*/s_kernel_stats.csv gpurun_out/condab/$T/s_kernel_stats.csv 2>/dev/null | head -1 | cut -d, -f1-5)"
done
