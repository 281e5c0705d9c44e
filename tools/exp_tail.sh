#!/bin/bash
# tools/exp_tail.sh: fractional rows per wave (bench.py --tune pipe_rows_x100=..) at the driver's --steps 20 and at 200 steps, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$R/gpurun_out/exp_tail; mkdir -p $T; cd $R
for x in 200 175 150 125 200 175 150 125; do
  for k in 20 20 200; do
    w=5; [ $k = 200 ] && w=20
    timeout 300 python3 bench.py --tune pipe_rows_x100=$x --steps $k --warmup $w --no-cpu-baseline --no-secondary --sustained-s 0 --no-hot 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('rpw_x100 $x steps $k: us/batch %.3f  alone %.3f' % (d['ms_per_step'] * 1e3, r['kernel_ms_per_batch'] * 1e3))"
  done
done | tee $T/out.txt
