#!/bin/bash
# tools/profile_round5_cfg4.sh <tag>: config 4's evidence of round 5 under gpurun_out/<tag>/: bench lines (--steps 100 / 20; the dataset beyond the
# Infinity Cache), kernel stats with one batch in flight, SQ instruction mix of the step's kernels, the kernels' lines of tools/cfg4_sub.py
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/${1:-r05c4}; mkdir -p $T; cd $R
b() { local name=$1; shift; timeout 600 python3 bench.py "$@" > $T/bench_$name.json 2> $T/bench_$name.err || echo "bench $name FAILED" | tee -a $T/status.txt; }
b cfg4 --workload cfg4 --steps 100 --warmup 10
b cfg4_k20 --workload cfg4 --steps 20 --warmup 3
GVL_CFG4_S=512 b cfg4_cold --workload cfg4 --steps 100 --warmup 10
GVL_DBG=536870912 b cfg4_noplans --workload cfg4 --steps 100 --warmup 10
GVL_DBG=1073741824 b cfg4_r04routing --workload cfg4 --steps 100 --warmup 10
CFG4_DBGS="0" bash tools/profile_cfg4.sh ${1:-r05c4}/cfg4 > $T/cfg4_kernels.txt 2>&1
cp $(find $T/cfg4/cfg4_prof_d0 -name "*kernel_stats.csv" | head -1) $T/cfg4_kernel_stats.csv
bash tools/pmc_cfg4_sq.sh > $T/cfg4_sq.txt 2>&1
python3 tools/cfg4_sub.py > $T/cfg4_sub.txt 2>&1
python3 tools/ref_long.py 2>&1 | grep -v amdgpu.ids > $T/ref_long.txt
find $T -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $T -name "*.db" -delete
cat $T/cfg4_kernels.txt | tail -12; grep -v amdgpu.ids $T/cfg4_sq.txt | head -30
python3 - $T <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1].ljust(28), "step us %.2f (%.3f)  hap kernel %.2f us frac %.3f  %s" % (
            d["ms_per_step"] * 1e3, r["step_frac"], r["kernel_ms"] * 1e3, r["frac"], {k[:24]: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items() if isinstance(v, dict)}))
    except Exception as e:
        print(f, "unreadable", e)
PY
