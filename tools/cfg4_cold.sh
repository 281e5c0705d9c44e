#!/bin/bash
# tools/cfg4_cold.sh: the cfg4 step and its kernels (one batch in flight, rocprofv3 kernel stats) for datasets of 64 / 256 / 512 samples x 16
# regions: 64 samples = 70 MB of intervals + records + plans, resident in the 256 MB Infinity Cache; 256 and up are not
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=/tmp/cfg4_cold_$$; mkdir -p $T; cd /tmp
for S in ${SAMPLES:-64 256 512}; do
  GVL_CFG4_S=$S python3 $R/bench.py --workload cfg4 --steps 100 --warmup 10 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=$S: step us', round(d['ms_per_step']*1e3, 2), '|', d['config']['dataset'])"
  GVL_CFG4_S=$S GVL_CFG4_INFLIGHT=1 GVL_CFG4_GROUP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $T/s -- python3 $R/bench.py --workload cfg4 --steps 40 --warmup 8 > $T/s.log 2>&1
  python3 - $T $S <<'PY'
import csv, sys, glob
T, S = sys.argv[1:3]
for r in csv.DictReader(open(glob.glob(f"{T}/s/**/*kernel_stats.csv", recursive=True)[0])):
    if any(k in r["Name"] for k in ("realign_paint_kernel", "recon_lean_kernel", "hap_plan", "track_plan", "track_lengths", "prepare_request")):
        print(f"   S={S} one batch in flight: {r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:46]:46s} {float(r['AverageNs'])/1e3:8.2f} us x {r['Calls']}")
PY
  rm -rf $T/s
done
rm -rf $T
