#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02f}
mkdir -p $O
cd $R
echo "== r01 stamps (hot)"; (cd tools/ab_r01 && timeout 300 python tools/stamps.py cfg3 2>&1 | grep -v amdgpu.ids | head -12)
echo "== new (many) stamps hot dbg192"; timeout 300 python tools/stamps.py cfg3 small 1 192 2>&1 | grep -v amdgpu.ids
echo "== new (single args) stamps hot dbg192"; GVL_DIAG_LIB=$R/tools/lib_single_diag.so timeout 300 python tools/stamps.py cfg3 small 1 192 2>&1 | grep -v amdgpu.ids
echo "== new (single args) stamps hot dbg0"; GVL_DIAG_LIB=$R/tools/lib_single_diag.so timeout 300 python tools/stamps.py cfg3 small 1 0 2>&1 | grep -v amdgpu.ids
run() { local name=$1; local dbg=$2; local lib=$3; shift 3
  GVL_HIP_LIB=$lib GVL_DBG=$dbg timeout 600 python bench.py --no-cpu-baseline --no-hot "$@" > $O/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed"; }
for rep in 1 2; do
(cd tools/ab_r01 && timeout 300 python bench.py --steps 300 --streams 4 --no-cpu-baseline > $O/r01_bench_$rep.json 2> $O/r01_bench_$rep.err)
for dbg in 0 192; do
run hot_many_d${dbg}_$rep $dbg $R/genvarloader_amd/libgvl_hip.so --steps 200 --scale small --rotate 1
run hot_single_d${dbg}_$rep $dbg $R/tools/lib_single.so --steps 200 --scale small --rotate 1
done; done
run cold_single_d0 0 $R/tools/lib_single.so --steps 200
run cold_many_d0 0 $R/genvarloader_amd/libgvl_hip.so --steps 200
for f in $O/r01_bench_*.json $O/bench_*.json; do echo $(basename $f); python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print("   ms/step %.4f | kern %.4f" % (d["ms_per_step"], r["kernel_ms"]))
except Exception as e:
    print("   failed", e)
PY
done
