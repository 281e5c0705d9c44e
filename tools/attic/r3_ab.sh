#!/bin/bash
# tools/r3_ab.sh "<dbg values>" [workloads]: same-box A/B of GVL_DBG settings on the default (cold) bench; 16384 = no lean kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
DBGS=${1:-"0 16384"}; WL=${2:-"cfg3 cfg2"}
for rep in 1 2; do for w in $WL; do for dbg in $DBGS; do echo -n "$w GVL_DBG=$dbg: "; GVL_DBG=$dbg timeout 300 python bench.py --no-cpu-baseline --steps 200 --workload $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('in flight us/step %.2f   alone %.2f   hot alone %.2f' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, (r['kernel_ms_hot'] or 0)*1e3))"; done; done; done
