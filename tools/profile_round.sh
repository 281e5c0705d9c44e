#!/bin/bash
# tools/profile_round.sh <tag>: rocprofv3 evidence for the round, written under gpurun_out/<tag>/
#  1. --kernel-trace --stats of the DEFAULT bench command (what the driver runs)
#  2. the same with --streams 1 (the kernel alone on one stream)
#  3. PMC passes (kernel-trace + pmc only, one counter family per pass): FETCH_SIZE, WRITE_SIZE
#     for the bench kernel and for tools/kbench.bin (known byte counts -> calibration)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/${1:-prof}
mkdir -p $T
cd /tmp
[ -x $R/tools/kbench.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $R/tools/kbench.hip -o $R/tools/kbench.bin
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats_default -- python3 $R/bench.py --no-cpu-baseline > $T/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats_1stream -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $T/stats_1stream.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $T/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 30 --warmup 5 > $T/pmc_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $T/pmc_kbench_$c -- $R/tools/kbench.bin > $T/pmc_kbench_$c.log 2>&1
done
for d in stats_default stats_1stream; do echo "== $d"; cat $(find $T/$d -name "*kernel_stats.csv" | head -1); tail -1 $T/$d.log | cut -c1-400; done
python3 - $T <<'PY'
import csv, sys, glob, collections
T = sys.argv[1]
for tag in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_kbench_FETCH_SIZE", "pmc_kbench_WRITE_SIZE"):
    f = glob.glob(f"{T}/{tag}/**/*counter_collection.csv", recursive=True)
    if not f: print(tag, "no file"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        print(f"{tag:24s} {k:70s} {c:12s} n={len(v):4d} mean={sum(v)/len(v):12.1f}")
PY
