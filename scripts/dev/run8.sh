cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_snmpc.py -m gpu -q -k "pipeline_vs_oracle" 2>&1 | grep -E "^E |passed|failed|Error|assert" | head -30
