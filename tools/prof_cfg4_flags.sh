#!/bin/bash
# tools/prof_cfg4_flags.sh "<flags...>" [in_flight]: rocprofv3 kernel stats of the cfg4 step under each GVL_DBG value
# (in_flight 1: every kernel alone on the chip)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$R/gpurun_out/prof_cfg4_flags; mkdir -p $T
FLAGS=${1:-"0 268435456"}
export GVL_CFG4_INFLIGHT=${2:-3}
cd /tmp
for f in $FLAGS; do
  GVL_DBG=$f rocprofv3 --kernel-trace --stats --output-format csv -d $T/f$f -- python3 $R/bench.py --workload cfg4 --steps 100 --warmup 10 > $T/f$f.log 2>&1
  echo "== GVL_DBG $f in_flight $GVL_CFG4_INFLIGHT"
  python3 - $(ls -t $(find $T/f$f -name "*kernel_stats.csv") | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    n = r["Name"]
    n = n[n.find("::") + 2:] if "::" in n else n
    print("  %-60s calls %6s avg %9.2f us  min %8.2f max %8.2f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done | tee $T/out_${GVL_CFG4_INFLIGHT}.txt
