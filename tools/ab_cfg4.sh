#!/bin/bash
# tools/ab_cfg4.sh name1 name2 ...: same-box A/B of builds tools/lib_<name>.so on the cfg4 step (no profiler): step, haplotype kernel, tracks batch
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for n in "$@"; do
  GVL_HIP_LIB=$PWD/tools/lib_$n.so python bench.py --workload cfg4 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib $n: cfg4 step us', round(d['ms_per_step']*1e3, 2), 'hap kernel us', round(d['roofline']['kernel_ms']*1e3, 2), 'tracks_batch', round([v['ms'] for k, v in d['kernels'].items() if k.startswith('gvl_tracks')][0]*1e3, 1))"
done; done
