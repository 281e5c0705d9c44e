#!/bin/bash
# tools/quick.sh : GPU tests (planned path, then scalar path forced) + kernel time for cfg3/cfg2
for d in 0 8 32 512; do echo "GVL_DBG=$d:"; GVL_DBG=$d timeout 1000 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -2; done
for w in cfg3 cfg2; do for d in ${DBGS:-0}; do echo -n "$w dbg=$d: "; GVL_DBG=$d timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --workload $w --streams ${STREAMS:-1} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_us', round(d['roofline']['kernel_ms']*1000,2), 'GB/s', round(d['roofline']['achieved'],1), 'step_us', round(d['ms_per_step']*1000,2))"; done; done
