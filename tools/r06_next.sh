#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for lib in genvarloader_amd/libgvl_hip.so tools/lib_paint_wpb2.so tools/lib_paint_wpb1.so; do echo -n "$lib: "; GVL_HIP_LIB=$PWD/$lib python bench.py --workload cfg4 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernels']
print('step %.1f us' % (d['ms_per_step'] * 1e3), {n[:28]: round(v['ms'] * 1e3, 1) for n, v in k.items() if isinstance(v, dict) and ('tracks_batch' in n or 'realign' in n)})"; done; done > $O/cfg4_wpb_ab.txt; cat $O/cfg4_wpb_ab.txt
