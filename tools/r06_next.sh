#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gpu_suite_final.log 2>&1; tail -n 4 $O/gpu_suite_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
GVL_DBG=262144 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok under GVL_DBG=262144')" 2>&1 | tail -n 1
