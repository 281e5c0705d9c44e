#!/bin/bash
# tools/exp_cfg4_inflight.sh: the cfg4 step for in_flight x group of the native ring, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$R/gpurun_out/exp_cfg4_inflight; mkdir -p $T; cd $R
for combo in "3 1" "2 1" "4 1" "6 1" "3 2" "2 2" "3 1"; do
  set -- $combo
  GVL_CFG4_INFLIGHT=$1 GVL_CFG4_GROUP=$2 timeout 300 python3 bench.py --workload cfg4 --steps 100 --warmup 10 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('in_flight $1 group $2: step %.2f us' % (d['ms_per_step'] * 1e3))"
done | tee $T/out.txt
