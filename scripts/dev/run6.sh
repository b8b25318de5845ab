set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/dev/ab.sh exp_libs/lib_r3base.so shipped exp_libs/lib_r3base.so shipped
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
