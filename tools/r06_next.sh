#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_splice.py tests/test_loader.py -q -x -k "splice or Splice or spliced" 2>&1 | tail -n 3
python tools/spliced_bench.py 256 2>&1 | grep -E "ms_per_step|kernel_ms"
