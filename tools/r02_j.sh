#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02j}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 900 python tools/epoch_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/epoch.txt
timeout 600 python tools/cfg4_dataset_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/cfg4_dataset.txt
timeout 600 python bench.py --workload cfg4 --steps 20 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc=$?"; cat $O/bench_cfg4.json | cut -c1-1500; tail -3 $O/bench_cfg4.err
