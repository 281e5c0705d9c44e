#!/bin/bash
# one GPU call: the new tests, gvl_tracks_batch A/B, spliced batches with and without the front workgroups
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_loader.py -q -x -k "sized_inside or row_plans_rows" > $O/t_tracks.log 2>&1; tail -3 $O/t_tracks.log
python -m pytest tests/test_gpu_parity.py -q -x -k "few_long_ones" > $O/t_mixed.log 2>&1; tail -3 $O/t_mixed.log
python -m pytest tests/test_gpu_svar2.py -q -x -k "consensus" > $O/t_cons.log 2>&1; tail -3 $O/t_cons.log
python tools/tracks_batch_ab.py > $O/tracks_ab.txt 2>&1; tail -4 $O/tracks_ab.txt
: > $O/spliced_ab.txt
for cfg in "4096 9000 1" "4096 2500 1" "4096 9000 0" "1024 9000 1 1" "256 9000 1 1" "256 9000 1"; do
  for d in 0 256; do echo "== $cfg GVL_DBG=$d" >> $O/spliced_ab.txt; GVL_DBG=$d python tools/spliced_bench.py $cfg 2>&1 | grep -E "workload|kernel_ms|routing_kernel_ms|ms_per_step" >> $O/spliced_ab.txt; done
done
cat $O/spliced_ab.txt
