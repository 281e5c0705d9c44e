#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02i}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 900 python tools/epoch_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/epoch.txt
export TMPDIR=/tmp; cd /tmp
QUICK=1 REPS=6 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/epoch_prof -- python3 $R/tools/epoch_bench.py > $O/epoch_prof.log 2>&1
cat $(find $O/epoch_prof -name "*kernel_stats.csv" | head -1) | head -8
