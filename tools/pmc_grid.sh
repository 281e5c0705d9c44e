#!/bin/bash
# tools/pmc_grid.sh <tag>: SQ instruction mix of the kernels tools/exp_grid.py launches (one G, one stream, a short leg)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmcgrid}
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  GS=${GS:-16} STREAMS=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/exp_grid.py ${SCALE:-hg38} 0.02 > $R/gpurun_out/$tag.log 2>&1
  f=$(find $R/gpurun_out/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "recon" not in k: continue
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
PY
  rm -rf $R/gpurun_out/$tag
done
