#!/bin/bash
# tools/ab_legs.sh name1 name2 ...: same-box A/B of builds tools/lib_<name>.so ("tree" = the tree's own library): the headline at --steps 20, the
# launch alone, sustained, and the ragged / annotated / channel-major legs -- three alternating passes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for n in "$@"; do
  if [ $n = tree ]; then unset GVL_HIP_LIB; else export GVL_HIP_LIB=$PWD/tools/lib_$n.so; fi
  GVL_BENCH_SKIP=cfg4,cfg4_cold,keep_mask,reference,random_shifts python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-s 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['secondary']
print('lib %-6s: step us %.3f  kernel/batch %.3f  sustained %.3f | ragged %.3f  annotated %.3f  onehot_cl %.3f' % ('$n', d['ms_per_step']*1e3, d['roofline']['kernel_ms_per_batch']*1e3, d['sustained']['ms_per_step']*1e3,
      s['ragged']['ms_per_step']*1e3, s['annotated']['ms_per_step']*1e3, s['onehot_cl']['ms_per_step']*1e3))"
done; done
