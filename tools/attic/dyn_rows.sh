#!/bin/bash
# tools/dyn_rows.sh: the dynamic-row-hand-out experiment (gvl_set_tuning(GVL_TUNE_PIPE_DYNAMIC, waves)) against the static schedule, same
# box, alternating: bench.py at the driver's --steps 20 and at 200 steps; parity first (the fuzz's many-batch leg under the knob).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for w in 0 8192 16384 24576; do
  for k in 20 200; do
    wu=5; [ $k = 200 ] && wu=20
    timeout 300 python3 bench.py --tune pipe_dynamic=$w --steps $k --warmup $wu --no-cpu-baseline --no-secondary --sustained-s 2 --no-hot 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('dynamic waves %6d steps %3d: us/batch %.3f  launch alone %.3f  sustained %.3f' % ($w, $k, d['ms_per_step'] * 1e3, r['kernel_ms_per_batch'] * 1e3, d['sustained']['ms_per_step'] * 1e3))"
  done
done; done
