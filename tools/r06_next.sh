#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gpu_suite_final2.log 2>&1; tail -n 2 $O/gpu_suite_final2.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
python bench.py > $O/bench_default4.json 2> $O/bench_default4.err; tail -c 200 $O/bench_default4.json
