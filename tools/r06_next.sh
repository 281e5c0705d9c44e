#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_bench.py tests/test_splice.py tests/test_loader.py tests/test_gpu_svar2.py -q -x 2>&1 | tail -n 3
python tools/spliced_bench.py 256 2>&1 | grep -E "ms_per_step"
python tools/ffi_bench.py 2>&1 | grep -v amdgpu.ids | tail -n 4
