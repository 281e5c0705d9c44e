#!/bin/bash
# tools/ab_libs.sh <tag> lib1.so lib2.so ...: same-box A/B of whole builds through bench.py (cold + hot), twice
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-ablibs}; shift
mkdir -p $O; cd $R
for rep in 1 2; do for lib in "$@"; do for sc in "--scale hg38" "--scale small --rotate 1"; do
  echo -n "$lib $sc: "; GVL_HIP_LIB=$R/$lib timeout 300 python bench.py --no-cpu-baseline --no-hot --steps 200 $sc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight ms/step %.4f   alone %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done; done
