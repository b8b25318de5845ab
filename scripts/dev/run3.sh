set -u
mkdir -p gpurun_out/r3b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -20 > gpurun_out/r3b/pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 2>gpurun_out/r3b/bench.err | grep metric > gpurun_out/r3b/bench.json
cat gpurun_out/r3b/pytest.log; tail -3 gpurun_out/r3b/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench.json'))
print({k:d[k] for k in ('value','value_single_stream','value_repeated_batch','ms_per_step','solve_ms_per_step')})
print(d['roofline']['kernel_ms'],d['roofline']['frac'],d['roofline']['sustained'],d['roofline']['dominant'])
print(d.get('n38'), d.get('parity_vs_oracle_max_abs_u0'))
PY
