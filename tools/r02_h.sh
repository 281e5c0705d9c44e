#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02h}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for sc in "small 1" "hg38 64"; do for dbg in 0 256; do timeout 300 python tools/stamps.py cfg3 $sc $dbg 2>&1 | grep -v amdgpu.ids; done; done
run() { local name=$1; local dbg=$2; shift 2
  GVL_DBG=$dbg timeout 600 python bench.py --no-cpu-baseline --no-hot "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed"; }
for rep in 1 2; do
for dbg in 0 256; do
run hot_d${dbg}_$rep $dbg --steps 200 --scale small --rotate 1
run cold_d${dbg}_$rep $dbg --steps 200
run hot_cfg2_d${dbg}_$rep $dbg --steps 200 --scale small --rotate 1 --workload cfg2
run cold_cfg2_d${dbg}_$rep $dbg --steps 200 --workload cfg2
done; done
for f in $O/bench_*.json; do echo $(basename $f); python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print("   ms/step %.4f | kern %.4f" % (d["ms_per_step"], r["kernel_ms"]))
except Exception as e:
    print("   failed", e)
PY
done
