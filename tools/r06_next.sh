#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "many or few_long or ragged" > $O/t_new.log 2>&1; tail -n 15 $O/t_new.log
