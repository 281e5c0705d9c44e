#!/bin/bash
# tools/profile_round_cfg4.sh <tag>: the round's evidence for BASELINE config 4 and the loader, under gpurun_out/<tag>/
#  1. cfg4 bench lines: default; GVL_DBG=1048576 (all-purpose haplotype kernel), 4194304 (tracks painted first), both
#  2. kernel stats of the step (one batch in flight), default and with both switches (= round 2's kernels)
#  3. PMC: FETCH_SIZE / WRITE_SIZE per launch, SQ instruction mix of the step's kernels; where realign_tracks_kernel<PAINT>'s go
#  4. cfg5 epochs through the native loader (isolated and chained, with and without the prepared-ahead epoch), genome-scale epoch
#  5. fuzz: lean kernel (one chunk, LONG), tracks from intervals, tracks painted first
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-prof_cfg4}
T=$R/gpurun_out/$tag
mkdir -p $T
cd $R
b() { local name=$1; shift; timeout 600 python3 bench.py "$@" > $T/bench_$name.json 2> $T/bench_$name.err || echo "bench $name FAILED" | tee -a $T/status.txt; }
b cfg4 --workload cfg4 --steps 100 --warmup 10
b cfg4_k20 --workload cfg4 --steps 20 --warmup 3
GVL_DBG=1048576 b cfg4_nolong --workload cfg4 --steps 100 --warmup 10
GVL_DBG=4194304 b cfg4_paintfirst --workload cfg4 --steps 100 --warmup 10
GVL_DBG=5242880 b cfg4_r02path --workload cfg4 --steps 100 --warmup 10
CFG4_DBGS="0 5242880" bash tools/profile_cfg4.sh $tag/cfg4 > $T/cfg4_kernels.txt 2>&1
bash tools/pmc_cfg4.sh $tag/pmc > $T/cfg4_pmc.txt 2>&1
bash tools/pmc_cfg4_sq.sh > $T/cfg4_sq.txt 2>&1
bash tools/pmc_realign_parts.sh > $T/realign_parts.txt 2>&1
python3 tools/kern_cfg4.py 0 2 6 1048576 2>&1 | grep -v amdgpu.ids > $T/kern_cfg4.txt
COMBOS=3x16,3x16t,3x8 REPS=4 python3 tools/epoch_bench.py 2>&1 | grep -v amdgpu.ids > $T/cfg5_epoch.txt
NO_PREFETCH=1 COMBOS=3x16,3x16t,3x8 REPS=4 python3 tools/epoch_bench.py 2>&1 | grep -v amdgpu.ids > $T/cfg5_epoch_noprefetch.txt
SCALE=hg38 COMBOS=3x16 python3 tools/epoch_bench.py 2>&1 | grep -v amdgpu.ids > $T/cfg5_epoch_hg38.txt
{
  python3 tools/fuzz_lean.py 30000 11
  FUZZ_LONG=1 python3 tools/fuzz_lean.py 20000 12
  GVL_DBG=32768 FUZZ_LONG=1 python3 tools/fuzz_lean.py 3000 13
  GVL_DBG=65536 FUZZ_LONG=1 python3 tools/fuzz_lean.py 6000 14
  python3 tools/fuzz_fused_tracks.py 6000 15
  GVL_DBG=2097152 python3 tools/fuzz_fused_tracks.py 1500 16
  python3 tools/fuzz_tracks.py 6000 17
  python3 tools/fuzz.py 3000 18
} 2>&1 | grep -v amdgpu.ids > $T/fuzz.txt
find $T -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $T -name "*.db" -delete
tail -3 $T/cfg4_kernels.txt; cat $T/fuzz.txt | tail -12
python3 - $T <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1].ljust(28), "step us %.2f  kernel %.2f us frac %.3f  step_GBps %.0f  %s" % (
            d["ms_per_step"] * 1e3, r["kernel_ms"] * 1e3, r["frac"], r["step_GBps"], {k[:24]: round(v["ms"] * 1e3, 1) for k, v in d["kernels"].items() if isinstance(v, dict)}))
    except Exception as e:
        print(f, "unreadable", e)
PY
