#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
for i in 1 2; do python -m pytest tests -m gpu -q -x > $O/gpu_suite_rep$i.log 2>&1; tail -n 1 $O/gpu_suite_rep$i.log; done
timeout 600 python tools/loader_stress.py > $O/loader_stress.txt 2>&1; tail -n 3 $O/loader_stress.txt
