#!/bin/bash
# tools/cmp_libs.sh [workloads]: same-box A/B of whole builds: every tools/lib_*.so through bench.py, two rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for l in tools/lib_*.so; do for w in ${1:-cfg3 cfg2}; do echo -n "$l $w: "; GVL_HIP_LIB=$PWD/$l timeout 200 python bench.py --steps 200 --no-cpu-baseline --sustained-s 0 --min-region-ms 300 --workload $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('in flight %.2f  alone %.2f  hot alone %.2f' % (d['ms_per_step']*1e3, r['kernel_ms']*1e3, (r['kernel_ms_hot'] or 0)*1e3))"; done; done; done
