#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_tracks.py tests/test_loader.py -q -x -k "track or Track or cfg4" > $O/t_tracks.log 2>&1; tail -n 3 $O/t_tracks.log
for sub in 1 2 4; do echo "== GVL_PAINT_SUB=$sub"; GVL_PAINT_SUB=$sub python tools/stamps_paint.py 2>&1 | grep -v amdgpu.ids; done > $O/stamps_paint_sub.txt; cat $O/stamps_paint_sub.txt
for d in 0 4096 0 4096; do echo -n "GVL_DBG=$d: "; GVL_DBG=$d python bench.py --workload cfg4 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); k = d['kernels']
print('step %.1f us' % (d['ms_per_step'] * 1e3), {n[:28]: round(v['ms'] * 1e3, 1) for n, v in k.items() if isinstance(v, dict) and ('tracks_batch' in n or 'realign' in n)})"; done > $O/cfg4_sub_ab.txt; cat $O/cfg4_sub_ab.txt
