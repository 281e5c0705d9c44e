#!/bin/bash
# tools/ab_dbg.sh "<dbg values>" [bench args]: same-box A/B of GVL_DBG settings on the default bench, cold and hot
cd ${GRAFT_REPO_ROOT:-/root/repo}
DBGS=${1:-0}; shift
for rep in 1 2; do for sc in "--scale hg38" "--scale small --rotate 1"; do for dbg in $DBGS; do echo -n "$sc GVL_DBG=$dbg $*: "; GVL_DBG=$dbg timeout 300 python bench.py --no-cpu-baseline --no-hot --steps 200 $sc "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight ms/step %.4f   alone %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done; done; done
