set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/dev/ab.sh exp_libs/lib_r3b_mfma4scale.so shipped exp_libs/lib_r3b_mfma4scale.so shipped
timeout 300 python scripts/phase_profile.py 4096 pipeline 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
