#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
python bench.py --workload cfg4 --steps 20 --warmup 3 2>$O/cfg4_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernels']
print('step %.1f us' % (d['ms_per_step'] * 1e3))
for n, v in k.items():
    if isinstance(v, dict): print('  ', n[:70], {a: (round(b * 1e3, 1) if 'ms' in a else round(b, 3)) for a, b in v.items() if a != 'algorithmic_bytes'})"
tail -n 3 $O/cfg4_err.txt
