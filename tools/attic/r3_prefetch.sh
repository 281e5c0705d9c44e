#!/bin/bash
# tools/r3_prefetch.sh: epochs prepared ahead -- loader tests, cfg5 epoch rates, cfg4 step
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r3_prefetch.txt
: > $O
timeout 1200 python -m pytest tests/test_loader.py -q -x 2>&1 | tail -5 | tee -a $O
timeout 600 python tools/epoch_bench.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $O
for i in 1 2; do
timeout 600 python bench.py --workload cfg4 --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('cfg4 step us', round(d['ms_per_step']*1e3, 2), d['config']['dataset'], 'kernel us', round(r['kernel_ms']*1e3, 2))
    else: print(l.rstrip()[:200])
" | tee -a $O
done
