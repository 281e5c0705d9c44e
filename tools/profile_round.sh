#!/bin/bash
# tools/profile_round.sh <tag>: the round's evidence, written under gpurun_out/<tag>/ (copy what is judged into profiles/)
#  1. bench lines: default (cold hg38, with cpu_baseline), --steps 20 / 2000 three times each, hot (--scale small --rotate 1),
#     cfg2, --haps, cfg4, cfg1 --cpu-only
#  2. rocprofv3 --kernel-trace --stats of the default command and of --streams 1 (the kernel alone); steps and warmup are multiples of
#     the 16 batches per launch there, so that every launch of the profiled kernel is a full one and the average is per launch
#  3. PMC passes (kernel-trace + pmc only, one counter per pass): FETCH_SIZE, WRITE_SIZE for the bench kernel cold and hot,
#     and for tools/kbench.bin (known byte counts -> calibration); SQ instruction mix
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/${1:-prof}
mkdir -p $T
cd $R
[ -x $R/tools/kbench.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $R/tools/kbench.hip -o $R/tools/kbench.bin
b() { local name=$1; shift; timeout 900 python3 bench.py "$@" > $T/bench_$name.json 2> $T/bench_$name.err || echo "bench $name FAILED" | tee -a $T/status.txt; }
b default
for i in 1 2 3; do b k20_$i --steps 20 --warmup 5 --no-cpu-baseline --no-secondary; b k2000_$i --steps 2000 --warmup 50 --no-cpu-baseline --sustained-s 0 --no-secondary; done
# round 3's path on the same box: a launch per batch, 3 in flight (recon_lean_kernel), at the driver's arguments and at 200 steps
b many1_k20 --many 1 --streams 3 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
b many1 --many 1 --streams 3 --no-cpu-baseline --sustained-s 0 --no-secondary
GVL_DBG=67108864 b nopipe_many5 --no-cpu-baseline --sustained-s 0 --no-secondary
GVL_DBG=16384 b nolean --no-cpu-baseline --sustained-s 0 --no-secondary
b hot_small --scale small --rotate 1 --no-cpu-baseline --sustained-s 0 --no-secondary
b warm64 --rotate 64 --no-cpu-baseline --sustained-s 0 --no-secondary
b rotate1024 --rotate 1024 --no-cpu-baseline --sustained-s 0 --no-secondary
b cfg2 --workload cfg2 --no-cpu-baseline --sustained-s 0
b cfg3_haps --haps --no-cpu-baseline --sustained-s 0
b cfg4 --workload cfg4 --steps 20 --warmup 3
b cfg1_cpu --workload cfg1 --cpu-only
timeout 300 $R/tools/kbench.bin > $T/kbench.txt 2>&1
# the loader: cfg5 epoch, genome-scale epoch, genome-scale RAGGED epoch (and each without the pipelined kernel / with per-batch sizing)
COMBOS=3x16,3x8 python3 tools/epoch_bench.py > $T/cfg5_epoch.txt 2>&1
SCALE=hg38 COMBOS=3x16,3x8,2x16 CHAIN=3 python3 tools/epoch_bench.py > $T/cfg5_epoch_hg38.txt 2>&1
GVL_DBG=67108864 SCALE=hg38 COMBOS=3x16 CHAIN=3 python3 tools/epoch_bench.py > $T/cfg5_epoch_hg38_nopipe.txt 2>&1
RAGGED=1 SCALE=hg38 COMBOS=3x16,3x8 CHAIN=3 python3 tools/epoch_bench.py > $T/ragged_epoch_hg38.txt 2>&1
GVL_DBG=134217728 RAGGED=1 SCALE=hg38 COMBOS=3x16 CHAIN=3 python3 tools/epoch_bench.py > $T/ragged_epoch_hg38_sizing_per_batch.txt 2>&1
GVL_DBG=67108864 RAGGED=1 SCALE=hg38 COMBOS=3x16 CHAIN=3 python3 tools/epoch_bench.py > $T/ragged_epoch_hg38_nopipe.txt 2>&1
python3 tools/pipe_deferred.py hg38 16 > $T/deferred_rows.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats_default -- python3 $R/bench.py --steps 192 --warmup 16 --no-cpu-baseline --sustained-s 0 --no-secondary --min-region-ms 100 > $T/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats_1stream -- python3 $R/bench.py --steps 192 --warmup 16 --no-cpu-baseline --streams 1 --no-hot --sustained-s 0 --no-secondary --min-region-ms 100 > $T/stats_1stream.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats_1stream_hot -- python3 $R/bench.py --steps 192 --warmup 16 --no-cpu-baseline --streams 1 --no-hot --sustained-s 0 --no-secondary --min-region-ms 100 --scale small --rotate 1 > $T/stats_1stream_hot.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $T/pmc_cold_$c -- python3 $R/bench.py --no-cpu-baseline --no-hot --streams 1 --steps 32 --warmup 16 --min-region-ms 1 --sustained-s 0 --no-secondary > $T/pmc_cold_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $T/pmc_hot_$c -- python3 $R/bench.py --no-cpu-baseline --no-hot --streams 1 --steps 32 --warmup 16 --min-region-ms 1 --sustained-s 0 --no-secondary --scale small --rotate 1 > $T/pmc_hot_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $T/pmc_kbench_$c -- $R/tools/kbench.bin > $T/pmc_kbench_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $T/pmc_cold_SQ -- python3 $R/bench.py --no-cpu-baseline --no-hot --streams 1 --steps 32 --warmup 16 --min-region-ms 1 --sustained-s 0 --no-secondary > $T/pmc_cold_SQ.log 2>&1
for d in stats_default stats_1stream stats_1stream_hot; do echo "== $d"; head -3 $(find $T/$d -name "*kernel_stats.csv" | head -1) | cut -c1-220; done
python3 - $T <<'PY'
import csv, sys, glob, collections, json
T = sys.argv[1]
for tag in ("pmc_cold_FETCH_SIZE", "pmc_cold_WRITE_SIZE", "pmc_hot_FETCH_SIZE", "pmc_hot_WRITE_SIZE", "pmc_kbench_FETCH_SIZE", "pmc_kbench_WRITE_SIZE", "pmc_cold_SQ"):
    f = glob.glob(f"{T}/{tag}/**/*counter_collection.csv", recursive=True)
    if not f: print(tag, "no file"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if "reconstruct" in k or "recon_lean" in k or "k_" in k:
            print(f"{tag:24s} {k:60s} {c:18s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
for f in sorted(glob.glob(f"{T}/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline", {})
        print(f.split("/")[-1].ljust(26), "value %.4e  ms/step %.5f  kernel_ms %s  frac %s  pipelined_frac %s" % (
            d["value"], d["ms_per_step"] or 0, r.get("kernel_ms"), r.get("frac"), r.get("pipelined_frac")))
    except Exception as e:
        print(f, "unreadable", e)
PY
