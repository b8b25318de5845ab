# GPU clock and power while the bench loop runs (one stream, same batch): rocm-smi sampled every second next to a 20 000-step run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 200 python bench.py --steps 20000 --warmup 3 --streams ${1:-1} --same-batch --no-cpu-baseline --no-schedule-legs > /tmp/b.log 2>&1) &
BP=$!
for i in $(seq 1 45); do
  echo "t=$i $(rocm-smi --showclocks 2>/dev/null | grep -i 'sclk' | head -1 | sed 's/.*(\(.*\)).*/\1/') $(rocm-smi --showpower 2>/dev/null | grep -i 'power (W)' | head -1 | sed 's/.*: //') W"
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
grep metric /tmp/b.log | cut -c1-160
