#!/bin/bash
# tools/pmc_sq_cfg3.sh [GVL_DBG]: SQ instruction mix per row of the headline kernel (cold, one stream, full launches of 16 batches)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; T=/tmp/pmc_sq_$$; mkdir -p $T; cd /tmp
GVL_DBG=${1:-0} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $T -- python3 $R/bench.py --no-cpu-baseline --no-hot --streams 1 --steps 32 --warmup 16 --min-region-ms 1 --sustained-s 0 --no-secondary > $T/log 2>&1
python3 - $T ${1:-0} <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "recon_lean_rows_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = 65536.0
print("GVL_DBG", sys.argv[2], " per row:", "  ".join(f"{k.replace('SQ_INSTS_', '')} {sum(v) / len(v) / rows:.1f}" for k, v in sorted(acc.items()) if k != "SQ_WAVES"), " waves", sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"])))
PY
rm -rf $T
