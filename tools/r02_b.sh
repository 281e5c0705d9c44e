#!/bin/bash
# GPU pass: full suite + cold/hot bench under the new record layout, with A/B switches
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02b}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
tail -25 $O/pytest.log
run() { # name, env, args...
  local name=$1; local dbg=$2; shift 2
  GVL_DBG=$dbg timeout 600 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed" | tee -a $O/status.txt
}
run cold_default 0 --steps 200
run cold_nosrec 64 --steps 200
run cold_nospec 128 --steps 200
run cold_neither 192 --steps 200
run cold_k20 0 --steps 20 --warmup 5
run cold_k2000 0 --steps 2000
run hot_default 0 --steps 200 --scale small --rotate 1
run hot_nosrec 64 --steps 200 --scale small --rotate 1
run hot_nospec 128 --steps 200 --scale small --rotate 1
run hot_neither 192 --steps 200 --scale small --rotate 1
run cold_cfg2 0 --steps 200 --workload cfg2
run cold_haps 0 --steps 200 --haps
for f in $O/bench_*.json; do echo $(basename $f); python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print("   value %.3e ms/step %.4f wall %.4f | kern %.4f hot %.4f frac %.3f pip_frac %.3f regions %d" % (d["value"], d["ms_per_step"], d["timing"]["wall_ms_per_step"], r["kernel_ms"], r["kernel_ms_hot"] or 0, r["frac"], r["pipelined_frac"], d["timing"]["regions"]))
except Exception as e:
    print("   failed", e)
PY
done
