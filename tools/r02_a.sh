#!/bin/bash
# first GPU pass of round 2: suite, the new bench (cold / hot, K = 20 vs 200 vs 2000), streaming floors
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
for k in 20 200 2000; do
  timeout 600 python bench.py --steps $k --warmup 5 --no-cpu-baseline > $O/bench_hg38_k$k.json 2> $O/bench_hg38_k$k.err; echo "bench k=$k rc=$?" | tee -a $O/status.txt
done
timeout 600 python bench.py --steps 200 --scale small --rotate 1 --no-cpu-baseline > $O/bench_small_hot.json 2> $O/bench_small_hot.err
timeout 600 python bench.py --steps 200 --scale small --rotate 64 --no-cpu-baseline > $O/bench_small_rot.json 2> $O/bench_small_rot.err
timeout 600 python bench.py --steps 200 --scale hg38 --rotate 1 --no-cpu-baseline > $O/bench_hg38_hot.json 2> $O/bench_hg38_hot.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/status.txt
[ -x $R/tools/kbench.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $R/tools/kbench.hip -o $R/tools/kbench.bin
timeout 120 $R/tools/kbench.bin > $O/kbench.txt 2>&1
tail -3 $O/pytest.log; for f in $O/bench_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print(" value %.3e ms/step %.4f wall %.4f kern %.4f hot %s frac %.3f pip_frac %.3f regions %d" % (d["value"], d["ms_per_step"], d["timing"]["wall_ms_per_step"], r["kernel_ms"], r["kernel_ms_hot"], r["frac"], r["pipelined_frac"], d["timing"]["regions"]))
except Exception as e:
    print(" failed", e)
PY
done; cat $O/kbench.txt
