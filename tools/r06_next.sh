#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "few_long or short_rows_many" > $O/t_new.log 2>&1; tail -n 3 $O/t_new.log
: > $O/spliced_crews_ab.txt
for cfg in "4096 9000 1" "1024 9000 1"; do for d in 0 256 0 256; do echo "== $cfg GVL_DBG=$d" >> $O/spliced_crews_ab.txt; GVL_DBG=$d python tools/spliced_bench.py $cfg 2>&1 | grep -E "workload|\"kernel_ms|routing_kernel_ms" >> $O/spliced_crews_ab.txt; done; done
cat $O/spliced_crews_ab.txt
python -m pytest tests -m gpu -q -x > $O/gpu_suite3.log 2>&1; tail -n 5 $O/gpu_suite3.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
