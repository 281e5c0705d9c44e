#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r02o}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
bash tools/profile_cfg4.sh $1 2>&1 | tail -22 | cut -c1-140
timeout 600 python bench.py --workload cfg4 --steps 20 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; python - $O/bench_cfg4.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg4 ms/step %.4f value %.3e" % (d["ms_per_step"], d["value"]), {k:(round(v["ms"]*1e3,1) if isinstance(v,dict) else round(v*1e3,1)) for k,v in d["kernels"].items()}, "recon us", round(d["roofline"]["kernel_ms"]*1e3,1))
PY
timeout 600 python tools/cfg4_dataset_bench.py 2>&1 | grep -v amdgpu.ids
