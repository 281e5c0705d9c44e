export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/pmc_cfg4_sq
mkdir -p $T
export GVL_CFG4_INFLIGHT=1 GVL_CFG4_GROUP=1
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $T/a -- python3 $R/bench.py --workload cfg4 --steps 6 --warmup 2 > $T/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $T/b -- python3 $R/bench.py --workload cfg4 --steps 6 --warmup 2 > $T/b.log 2>&1
python3 - $T <<'PY'
import csv, sys, glob, collections
T = sys.argv[1]
for d in ("a", "b"):
    f = glob.glob(f"{T}/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no file"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[(r["Kernel_Name"][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        if any(x in k for x in ("reconstruct_kernel", "recon_lean", "realign", "intervals_to_tracks_tiled")):
            v = sorted(v); print(f"{k:64s} {c:24s} n={len(v):3d} median={v[len(v)//2]:14.1f}")
PY
rm -rf $T
