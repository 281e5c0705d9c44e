#!/bin/bash
# tools/r3_sub.sh: chunks per wave of the lean kernel's LONG form (GVL_LEAN_SUB): parity, fuzz, timings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "long_rows or lengths_snp" 2>&1 | tail -2
for sub in 2 1 4 3; do
  echo "== GVL_LEAN_SUB=$sub"
  GVL_LEAN_SUB=$sub FUZZ_LONG=1 timeout 900 python tools/fuzz_lean.py ${1:-1500} $((20 + sub)) 2>&1 | grep -v amdgpu.ids | tail -4
  GVL_LEAN_SUB=$sub GVL_DBG=32768 FUZZ_LONG=1 timeout 900 python tools/fuzz_lean.py 300 $((30 + sub)) 2>&1 | grep -v amdgpu.ids | tail -2
  GVL_LEAN_SUB=$sub timeout 300 python tools/kern_cfg4.py 0 2 2>&1 | grep -v amdgpu.ids
  GVL_LEAN_SUB=$sub timeout 300 python bench.py --workload cfg4 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 step us', round(d['ms_per_step']*1e3, 2), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 2))"
done
timeout 300 python bench.py --no-cpu-baseline --sustained-s 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 us', round(d['ms_per_step']*1e3, 3), 'kernel us', round(d['roofline']['kernel_ms']*1e3, 2))"
