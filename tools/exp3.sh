cd $GRAFT_REPO_ROOT
for many in 1 2 4 5 10; do for st in 2 3 4; do for K in 20 200; do
echo -n "many=$many streams=$st K=$K: "
python bench.py --steps $K --warmup 5 --many $many --streams $st --no-cpu-baseline --sustained-s 0 --no-hot --min-region-ms 300 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_us', round(d['ms_per_step']*1000,3))"
done; done; done
