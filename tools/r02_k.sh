#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02k}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for dbg in 0 1024; do GVL_DBG=$dbg timeout 600 python bench.py --workload cfg4 --steps 20 --warmup 3 > $O/bench_cfg4_d$dbg.json 2> $O/bench_cfg4_d$dbg.err; echo "cfg4 dbg=$dbg rc=$?"; done
python - $O <<'PY'
import json,sys
for dbg in (0,1024):
    try:
        d=json.loads(open(f"{sys.argv[1]}/bench_cfg4_d{dbg}.json").read().strip().splitlines()[-1])
        print(dbg, "ms/step %.4f" % d["ms_per_step"], {k:(round(v["ms"]*1e3,1) if isinstance(v,dict) else round(v*1e3,1)) for k,v in d["kernels"].items()}, "recon us", round(d["roofline"]["kernel_ms"]*1e3,1))
    except Exception as e: print(dbg,"failed",e)
PY
run() { local name=$1; shift
  timeout 600 python bench.py --no-cpu-baseline --no-hot "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed"; }
for rep in 1 2; do
run cold_pf0_$rep --steps 200
run cold_pf2_$rep --steps 200 --prefetch 2
run cold_pf4_$rep --steps 200 --prefetch 4
run cold_pf8_$rep --steps 200 --prefetch 8
done
run hot_pf0 --steps 200 --scale small --rotate 1
run small_rot_pf0 --steps 200 --scale small --rotate 64
run small_rot_pf4 --steps 200 --scale small --rotate 64 --prefetch 4
for f in $O/bench_cold*.json $O/bench_hot*.json $O/bench_small*.json; do echo $(basename $f); python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print("   ms/step %.4f | kern %.4f" % (d["ms_per_step"], r["kernel_ms"]))
except Exception as e:
    print("   failed", e)
PY
done
