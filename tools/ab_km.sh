#!/bin/bash
# tools/ab_km.sh: the pipelined kernel with keep masks / ref_only compiled out (default for launches without them) against the general
# form (GVL_DBG=1073741824 forces it) on the headline, cold and hot, same box, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for dbg in 0 1073741824; do
  for mode in "cold" "hot --scale small --rotate 1"; do
    set -- $mode; m=$1; shift
    GVL_DBG=$dbg python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sustained-s 2 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GVL_DBG %-10s %-4s: step us %.3f  kernel/batch us %.3f  sustained us %.3f' % ('$dbg', '$m', d['ms_per_step']*1e3, d['roofline']['kernel_ms_per_batch']*1e3, d['sustained']['ms_per_step']*1e3))"
  done
done; done
