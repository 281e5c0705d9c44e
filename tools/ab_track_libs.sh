#!/bin/bash
# tools/ab_track_libs.sh name1 name2 ...: same-box A/B of builds tools/lib_<name>.so: the track kernels of the cfg4 step under the
# profiler (one batch in flight; GVL_DBG=0: tracks straight from the intervals, 4194304: painter + realignment) and the cfg4 step
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/ab_track_libs
rm -rf $T; mkdir -p $T
cd /tmp
for rep in 1 2; do for n in "$@"; do
export GVL_HIP_LIB=$R/tools/lib_$n.so
for dbg in 0 4194304; do
GVL_DBG=$dbg GVL_CFG4_INFLIGHT=1 GVL_CFG4_GROUP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $T/s -- python3 $R/bench.py --workload cfg4 --steps 6 --warmup 2 > $T/s.log 2>&1
python3 - $T $n $dbg <<'PY'
import csv, sys, glob
T, n, dbg = sys.argv[1:4]
for r in csv.DictReader(open(glob.glob(f"{T}/s/**/*kernel_stats.csv", recursive=True)[0])):
    if any(k in r["Name"] for k in ("realign_tracks_kernel", "realign_paint_kernel", "recon_lean_kernel", "intervals_to_tracks_tiled")) and int(r["Calls"]) > 100:
        print(f"lib {n} GVL_DBG={dbg}: {r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:44]:44s} {float(r['AverageNs'])/1e3:.2f} us")
PY
rm -rf $T/s
GVL_DBG=$dbg python3 $R/bench.py --workload cfg4 --steps 100 --warmup 10 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib $n GVL_DBG=$dbg: cfg4 step us', round(d['ms_per_step']*1e3, 2), {k[:20]: round(v['ms']*1e3,1) for k,v in d['kernels'].items() if isinstance(v,dict)})"
done; done; done
rm -rf $T
