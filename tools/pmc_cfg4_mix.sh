export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$R/gpurun_out/pmc_cfg4_quick; rm -rf $T; mkdir -p $T
cd /tmp
export GVL_CFG4_INFLIGHT=1
for f in 0 8388608 16777216; do
GVL_DBG=$f rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $T/g$f -- python3 $R/bench.py --workload cfg4 --steps 20 --warmup 3 --max-regions 3 > $T/g$f.log 2>&1
python3 - $T/g$f $f <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "realign" if "realign" in k else ("recon_long" if "recon_lean_kernel" in k else None)
        if k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
med = {kc: sorted(v)[len(v)//2] for kc, v in acc.items()}
for k in ("realign", "recon_long"):
    w = med.get((k, "SQ_WAVES"), 1)
    print("GVL_DBG", sys.argv[2], k, "waves %d" % w, "  ".join("%s %.0f" % (c.replace("SQ_INSTS_", ""), med[(k, c)] / w) for (kk, c) in sorted(med) if kk == k and c != "SQ_WAVES"))
PY
done
