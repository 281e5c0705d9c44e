#!/bin/bash
# same-box A/B: round-1 tree vs the current one (hot), stamps of the current one, cold/hot bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02e}
mkdir -p $O
cd $R
for rep in 1 2; do
 (cd tools/ab_r01 && timeout 300 python bench.py --steps 300 --streams 4 --no-cpu-baseline > $O/r01_bench_$rep.json 2> $O/r01_bench_$rep.err)
 python - $O/r01_bench_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("r01 tree: ms/step %.4f kern %.4f" % (d["ms_per_step"], r["kernel_ms"]))
PY
done
for sc in "small 1" "hg38 64"; do for dbg in 0 192; do timeout 300 python tools/stamps.py cfg3 $sc $dbg; done; done 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
run() { local name=$1; local dbg=$2; shift 2
  GVL_DBG=$dbg timeout 600 python bench.py --no-cpu-baseline --no-hot "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed" | tee -a $O/status.txt; }
for dbg in 0 128 192; do
run cold_d${dbg} $dbg --steps 200
run hot_d${dbg} $dbg --steps 200 --scale small --rotate 1
done
run cold_many4_s1 0 --steps 200 --many 4 --streams 1
run cold_many4_s2 0 --steps 200 --many 4 --streams 2
run cold_many8_s1 0 --steps 200 --many 8 --streams 1
run cold_many8_s2 0 --steps 200 --many 8 --streams 2
run cold_many2_s2 0 --steps 200 --many 2 --streams 2
run cold_many2_s4 0 --steps 200 --many 2 --streams 4
run hot_many4_s1 0 --steps 200 --many 4 --streams 1 --scale small --rotate 1
run hot_many4_s2 0 --steps 200 --many 4 --streams 2 --scale small --rotate 1
run hot_many8_s1 0 --steps 200 --many 8 --streams 1 --scale small --rotate 1
for f in $O/bench_*.json; do echo $(basename $f); python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print("   value %.3e ms/step %.4f | kern %.4f frac %.3f pip_frac %.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["pipelined_frac"]))
except Exception as e:
    print("   failed", e)
PY
done
