#!/bin/bash
# PMC passes for the cfg4 step's kernels (one counter per pass, kernel-trace only): FETCH_SIZE / WRITE_SIZE per launch
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/${1:-pmc_cfg4}
mkdir -p $T
export GVL_CFG4_INFLIGHT=1 GVL_CFG4_GROUP=1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $T/pmc_cfg4_$c -- python3 $R/bench.py --workload cfg4 --steps 6 --warmup 2 > $T/pmc_cfg4_$c.log 2>&1
done
python3 - $T <<'PY'
import csv, sys, glob, collections
T = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{T}/pmc_cfg4_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print(c, "no file"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if any(x in k for x in ("reconstruct_kernel", "recon_lean", "realign", "intervals_to_tracks", "track_lengths")):
            v = sorted(v)
            print(f"{c:11s} {k:70s} n={len(v):4d} median={v[len(v)//2]:12.1f} KB")
PY
