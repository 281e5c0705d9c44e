#!/bin/bash
# rocprofv3 kernel stats of the cfg4 step (bench.py --workload cfg4) -> gpurun_out/<tag>/cfg4_prof
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02_cfg4}
mkdir -p $O
export TMPDIR=/tmp; cd /tmp
# one batch in flight, so that the kernel durations are not inflated by kernels of other batches running next to them
export GVL_CFG4_INFLIGHT=1 GVL_CFG4_GROUP=1
for dbg in ${CFG4_DBGS:-0 4194304}; do
GVL_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4_prof_d$dbg -- python3 $R/bench.py --workload cfg4 --steps 20 --warmup 3 > $O/cfg4_prof_d$dbg.log 2>&1
echo "== GVL_DBG=$dbg"; python3 - $(find $O/cfg4_prof_d$dbg -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 20: print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), ("%.2f" % (float(r["AverageNs"]) / 1000)).rjust(9), "us")
PY
done
