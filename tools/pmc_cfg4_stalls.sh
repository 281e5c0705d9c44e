#!/bin/bash
# tools/pmc_cfg4_stalls.sh: where the cfg4 kernels' cycles go (SQ / SQC counters, kernels alone: in_flight 1)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$R/gpurun_out/pmc_cfg4_stalls; rm -rf $T; mkdir -p $T
cd /tmp
rocprofv3 -L > $T/avail.txt 2>&1
export GVL_CFG4_INFLIGHT=1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $T/g$i -- python3 $R/bench.py --workload cfg4 --steps 20 --warmup 3 --max-regions 3 > $T/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $T/g$i.log)"
done
python3 - $T <<'PY'
import csv, sys, glob, collections
T = sys.argv[1]
for f in sorted(glob.glob(f"{T}/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "realign" if "realign" in k else ("recon_long" if "recon_lean_kernel" in k else None)
        if k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        v = sorted(v)
        print(f"{k:12s} {c:24s} n={len(v):4d} median={v[len(v)//2]:16.1f}")
PY
rm -rf $T/g*/
