#!/bin/bash
# tools/pmc_kern_cfg4.sh: SQ instruction mix of the cfg4 haplotype kernel (tools/kern_cfg4.py) per wave
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/pmc_kern_cfg4
rm -rf $T; mkdir -p $T
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $T/a -- python3 $R/tools/kern_cfg4.py ${1:-0} > $T/a.log 2>&1
python3 - $T <<'PY'
import csv, sys, glob, collections
T = sys.argv[1]
f = glob.glob(f"{T}/a/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    acc[(r["Kernel_Name"][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
waves = {}
for (k, c), v in sorted(acc.items()):
    if "recon" in k:
        v = sorted(v); m = v[len(v) // 2]
        if c == "SQ_WAVES": waves[k] = m
for (k, c), v in sorted(acc.items()):
    if "recon" in k:
        v = sorted(v); m = v[len(v) // 2]
        print(f"{k:90s} {c:20s} n={len(v):3d} median={m:14.1f} per wave {m / max(waves.get(k, 1), 1):8.1f}")
PY
rm -rf $T/a
