#!/bin/bash
# round 6: how many capsules (= HIP streams) the fresh-batch loop of bench.py runs with reproducibly, and whether the number of hardware queues the
# runtime multiplexes streams onto (GPU_MAX_HW_QUEUES, default 4) changes that. Six fresh processes per cell.   hw_queues.sh [queue settings ...]
run() { python bench.py --steps 40 --warmup 5 --streams $1 --no-cpu-baseline --no-schedule-legs --no-other-configs --no-host-legs 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f'%(d['value']/1e6), end=' ')"; }
QS="${@:-default 8 16}"
for Q in $QS; do for S in 3 4 5 6 8; do echo -n "queues $Q streams $S: "; for r in 1 2 3 4 5 6; do if [ $Q = default ]; then run $S; else GPU_MAX_HW_QUEUES=$Q run $S; fi; done; echo; done; done
