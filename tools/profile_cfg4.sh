#!/bin/bash
# rocprofv3 kernel stats for cfg4 (haplotypes + one realigned track) -> gpurun_out/<tag>_cfg4/
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_cfg4
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/track_bench.py 0 > $OUT/run.log 2>&1
tail -3 $OUT/run.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
