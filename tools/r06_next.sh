#!/bin/bash
# rows per wave for batches of short ragged rows (spliced exons): GVL_TUNE_PIPE_ROWS_X100 sweep
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
: > $O/spliced_rows.txt
for x in 100 150 200 300 400 600 800 1200 1600; do
  for cfg in "4096 9000 1 1" "4096 2500 1 1" "4096 2500 0 1" "1024 9000 1 1"; do
    echo "== $cfg x100=$x" >> $O/spliced_rows.txt; python tools/spliced_bench.py $cfg $x 2>&1 | grep -E "\"kernel_ms|routing_kernel_ms" >> $O/spliced_rows.txt
  done
done
cat $O/spliced_rows.txt
