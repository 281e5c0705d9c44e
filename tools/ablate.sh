#!/bin/bash
# tools/ablate.sh: where a batch's time goes -- the default bench with stores and / or reference loads switched off
# (GVL_DBG 2 = no stores, 4 = no reference loads, 6 = neither), cold (hg38 scale, rotating) and hot (64 Mbp, one batch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for sc in "--scale hg38" "--scale small --rotate 1"; do for dbg in 0 2 4 6; do echo -n "$sc GVL_DBG=$dbg: "; GVL_DBG=$dbg timeout 300 python bench.py --no-cpu-baseline --no-hot --steps 200 $sc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight ms/step %.4f   alone %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; done; done
