#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_splice.py -q -x 2>&1 | tail -n 3
python tools/spliced_bench.py 256 2>&1 | grep -E "ms_per_step"; python tools/spliced_bench.py 4096 2>&1 | grep -E "ms_per_step"
