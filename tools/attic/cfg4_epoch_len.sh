for S in 64 256; do
  echo "== S=$S"
  SECONDS=0; GVL_CFG4_S=$S python bench.py --workload cfg4 --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('step us', round(d['ms_per_step']*1e3, 2), d['config']['dataset'], 'kernel us', round(r['kernel_ms']*1e3, 2))
    else: print(l.rstrip()[:200])
"
echo "wall $SECONDS s"; done
