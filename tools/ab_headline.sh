#!/bin/bash
# tools/ab_headline.sh name1 name2 ...: same-box A/B of builds tools/lib_<name>.so ("tree" = the tree's own library) on the headline:
# bench.py --steps 20 (region), the kernel alone, sustained -- three alternating passes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for n in "$@"; do
  if [ $n = tree ]; then unset GVL_HIP_LIB; else export GVL_HIP_LIB=$PWD/tools/lib_$n.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sustained-s 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib $n: step us %.3f  kernel/batch us %.3f  sustained us %.3f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms_per_batch']*1e3, d['sustained']['ms_per_step']*1e3))"
done; done
