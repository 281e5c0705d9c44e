#!/bin/bash
# tools/exp_schedule.sh: bench.py --many x --streams at 20 and 200 steps on whatever box this lands on (the pool has fast and slow ones)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for c in "16 3" "16 2" "16 4" "8 3" "8 4" "12 3"; do set -- $c
  for k in 20 200; do w=5; [ $k = 200 ] && w=20
    python3 bench.py --many $1 --streams $2 --steps $k --warmup $w --no-cpu-baseline --no-secondary --sustained-s 0 --no-hot 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('many $1 streams $2 steps $k: %.3f us'%(d['ms_per_step']*1e3))"
  done; done; done
