#!/bin/bash
# A/B by GVL_DBG values: kernel time + pipelined step for cfg3 / cfg2
for w in cfg3 cfg2; do for d in ${DBGS:-0 16}; do for s in 1 3; do echo -n "$w dbg=$d streams=$s: "; GVL_DBG=$d timeout 200 python bench.py --steps 400 --warmup 30 --no-cpu-baseline --workload $w --streams $s 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_us', round(d['roofline']['kernel_ms']*1000,2), 'step_us', round(d['ms_per_step']*1000,2))"; done; done; done
