#!/bin/bash
# tools/r3_long.sh: the lean kernel's LONG form (rows of several chunks) -- parity, fuzz, cfg4 A/B against the all-purpose kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r3_long.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "long_rows or lengths_snp" 2>&1 | tail -5 | tee -a $O
FUZZ_LONG=1 timeout 900 python tools/fuzz_lean.py ${1:-400} 1 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $O
timeout 600 python tools/fuzz_lean.py 600 7 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O
for dbg in 0 1048576 0 1048576; do
  echo "== cfg4 GVL_DBG=$dbg" | tee -a $O
  GVL_DBG=$dbg timeout 600 python bench.py --workload cfg4 --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('step us', round(d['ms_per_step']*1e3, 2), 'kernel', r['kernel'], 'us', round(r['kernel_ms']*1e3, 2), 'frac', round(r['frac'], 3), {k[:40]: round(v["ms"]*1e3, 2) for k, v in d["kernels"].items() if isinstance(v, dict)})
    else: print(l.rstrip()[:300])
" | tee -a $O
done
