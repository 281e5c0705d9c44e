#!/bin/bash
# GPU: track tests on both walks, fuzz, cfg4 timing
timeout 900 python -m pytest tests/test_gpu_tracks.py -m gpu -q --timeout 300 -x 2>&1 | tail -4
GVL_DBG=8 timeout 900 python -m pytest tests/test_gpu_tracks.py -m gpu -q --timeout 300 -x 2>&1 | tail -2
timeout 600 python tools/fuzz_tracks.py ${FUZZ_N:-1500} 2>&1 | tail -3
timeout 300 python tools/track_bench.py 0 2>&1 | tail -2
GVL_DBG=8 timeout 300 python tools/track_bench.py 0 2>&1 | tail -2
timeout 300 python tools/track_bench.py 4 2>&1 | tail -2
