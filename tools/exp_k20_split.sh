#!/bin/bash
# tools/exp_k20_split.sh: how a K = 20-step region should be cut into launches -- 16 + 4 (the loader's full groups) against balanced
# cuts (10 + 10, 7 + 7 + 6, 5 x 4) on as many streams, same box, two rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for ms in "16 3" "16 2" "10 2" "10 3" "7 3" "5 4" "16 1"; do
    set -- $ms
    python3 bench.py --steps 20 --warmup 5 --many $1 --streams $2 --no-cpu-baseline --no-secondary --sustained-s 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('many %2d streams %d: us/batch %.3f  (kernel alone %.1f us per launch, frac %.3f)' % ($1, $2, d['ms_per_step'] * 1e3, d['roofline']['kernel_ms'] * 1e3, d['roofline']['frac']))"
  done
done
