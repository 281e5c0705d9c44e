#!/bin/bash
# tools/pmc_realign_parts.sh: where realign_tracks_kernel<PAINT>'s instructions go (GVL_DBG 8388608 / 16777216 ablations), SQ counters per wave
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/pmc_realign_parts
rm -rf $T; mkdir -p $T
[ -f $R/tools/libgvl_hip_diag.so ] || bash $R/tools/build_diag.sh      # (the ablation bits exist in the diagnostic build only)
export GVL_HIP_LIB=$R/tools/libgvl_hip_diag.so
export GVL_CFG4_INFLIGHT=1 GVL_CFG4_GROUP=1
cd /tmp
for dbg in 0 8388608 16777216; do
GVL_DBG=$dbg rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $T/a$dbg -- python3 $R/bench.py --workload cfg4 --steps 6 --warmup 2 > $T/a$dbg.log 2>&1
GVL_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $T/s$dbg -- python3 $R/bench.py --workload cfg4 --steps 6 --warmup 2 > $T/s$dbg.log 2>&1
python3 - $T $dbg <<'PY'
import csv, sys, glob, collections
T, dbg = sys.argv[1], sys.argv[2]
f = glob.glob(f"{T}/a{dbg}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "realign_tracks_kernel<true>" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {c: sorted(v)[len(v) // 2] for c, v in acc.items()}
w = med.get("SQ_WAVES", 1)
t = None
for r in csv.DictReader(open(glob.glob(f"{T}/s{dbg}/**/*kernel_stats.csv", recursive=True)[0])):
    if "realign_tracks_kernel<true>" in r["Name"]: t = float(r["AverageNs"]) / 1e3
print(f"GVL_DBG={dbg}: per wave VALU {med.get('SQ_INSTS_VALU', 0) / w:.0f} SALU {med.get('SQ_INSTS_SALU', 0) / w:.0f} LDS {med.get('SQ_INSTS_LDS', 0) / w:.0f}; kernel {t} us")
PY
done
rm -rf $T
