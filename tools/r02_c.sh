#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02c}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt; tail -3 $O/pytest.log
for sc in "small 1" "hg38 64"; do for dbg in 0 128 192; do timeout 300 python tools/stamps.py cfg3 $sc $dbg; done; done 2>&1 | tee $O/stamps.txt
run() { local name=$1; local dbg=$2; shift 2
  GVL_DBG=$dbg timeout 600 python bench.py --no-cpu-baseline --no-hot "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed" | tee -a $O/status.txt; }
for rep in 1 2; do for dbg in 0 128 192; do
run cold_d${dbg}_$rep $dbg --steps 200
run hot_d${dbg}_$rep $dbg --steps 200 --scale small --rotate 1
done; done
for f in $O/bench_*.json; do echo $(basename $f); python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print("   value %.3e ms/step %.4f | kern %.4f frac %.3f pip_frac %.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["pipelined_frac"]))
except Exception as e:
    print("   failed", e)
PY
done
