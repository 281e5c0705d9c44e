#!/bin/bash
# tools/ab_kernel.sh <tag> [dbg sets, default "0 64 128 192"]: same-box A/B of the reconstruct kernel --
#  the round-1 tree (exported once with `git archive <round-1 commit> | tar -x -C tools/ab_r01`, built in place)
#  next to the current one: GPU suite, per-phase stamps (tools/stamps.py, needs tools/libgvl_hip_diag.so
#  built with -DGVL_DIAG), and bench.py hot / cold under each GVL_DBG set, twice.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-ab}
SETS=${2:-"0 64 128 192"}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
echo "== r01 stamps (hot)"; (cd tools/ab_r01 && timeout 300 python tools/stamps.py cfg3 2>&1 | grep -v amdgpu.ids | head -10)
for sc in "small 1" "hg38 64"; do for dbg in 0 192; do timeout 300 python tools/stamps.py cfg3 $sc $dbg 2>&1 | grep -v amdgpu.ids; done; done
run() { local name=$1; local dbg=$2; shift 2
  GVL_DBG=$dbg timeout 600 python bench.py --no-cpu-baseline --no-hot "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed"; }
for rep in 1 2; do
(cd tools/ab_r01 && timeout 300 python bench.py --steps 300 --streams 4 --no-cpu-baseline > $O/r01_bench_$rep.json 2> $O/r01_bench_$rep.err)
for dbg in $SETS; do
run hot_d${dbg}_$rep $dbg --steps 200 --scale small --rotate 1
run cold_d${dbg}_$rep $dbg --steps 200
done; done
for f in $O/r01_bench_*.json $O/bench_*.json; do echo $(basename $f); python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print("   ms/step %.4f | kern %.4f" % (d["ms_per_step"], r["kernel_ms"]))
except Exception as e:
    print("   failed", e)
PY
done
