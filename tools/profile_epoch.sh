#!/bin/bash
# rocprofv3 kernel stats of the cfg5 epoch (native loader) -> gpurun_out/<tag>_epoch/
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_epoch
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/epoch_bench.py > $OUT/run.log 2>&1
python3 - $(find $OUT -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) > 100: print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1000, 2), "us")
PY
