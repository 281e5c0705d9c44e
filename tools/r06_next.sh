#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_svar2.py -q -x -k "del_only" 2>&1 | tail -n 8
