#!/bin/bash
# tools/r3_lib_ab.sh: same-box A/B of two builds (tools/lib_prev.so, tools/lib_new.so) on the cfg4 step
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for c in "prev 2 1" "new 2 1" "new 2 2" "new 2 4" "new 1 4"; do
  set -- $c
  GVL_HIP_LIB=$PWD/tools/lib_$1.so GVL_LEAN_SUB=$2 GVL_TRACK_SUB=$3 python bench.py --workload cfg4 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib $1 lean_sub $2 track_sub $3: cfg4 step us', round(d['ms_per_step']*1e3, 2), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 2), 'tracks_batch', round([v['ms'] for k, v in d['kernels'].items() if k.startswith('gvl_tracks')][0]*1e3, 1))"
done; done
