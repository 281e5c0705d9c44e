#!/bin/bash
# tools/ab_cfg4_flags.sh "<flags...>": the cfg4 step (bench.py --workload cfg4) under each GVL_DBG value, alternating, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$R/gpurun_out/ab_cfg4_flags; mkdir -p $T; cd $R
FLAGS=${1:-"0 268435456"}
for rep in 1 2; do
  for f in $FLAGS; do
    GVL_DBG=$f timeout 300 python3 bench.py --workload cfg4 --steps 100 --warmup 10 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; k = r.get('kernels', {})
print('GVL_DBG $f: step %.2f us  hap kernel %.2f  ' % (d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3) + '  '.join('%s %.2f' % (n[:28], v['ms'] * 1e3) for n, v in k.items() if isinstance(v, dict)))"
  done
done | tee $T/out.txt
